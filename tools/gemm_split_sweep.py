"""For every distinct contraction of ONE config-2 train step that runs below 140 TFLOP/s-equivalent and may be cut along K (no fused activation):
the call replayed with split_k = 1 .. 64 (HIP events, 10 repeats; the outputs are scratch here), best split next to the engine's choice.
usage: python tools/gemm_split_sweep.py [--config3]"""
import ctypes as C
import os
import sys
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multi_speaker_tts_amd import lib
from multi_speaker_tts_amd.engine import TrainEngine
from multi_speaker_tts_amd.params import Dims

dev = torch.device("cuda:0")
c3 = "--config3" in sys.argv
dims = Dims()
eng = TrainEngine(dims, device=dev, seed=1234, gemm_dtype="bf16" if c3 else "f32")
batch = bench.synthetic_batch(dims, 32, 128, 800, 1234, 0, dev)
w = eng.plan(32, 128, 800)
for _ in range(2):
    eng.forward(batch, w); eng.loss_and_backward(w); eng.adam_step()
torch.cuda.synchronize()
calls = []
real_call = lib.call


def spy(name, *a):
    if name in ("mstts_gemm_f32", "mstts_gemm_bf16"):
        cp = lib.GemmDesc()
        C.memmove(C.byref(cp), C.byref(a[0]._obj), C.sizeof(lib.GemmDesc))
        calls.append((name, cp))
    return real_call(name, *a)


lib.call = spy
import multi_speaker_tts_amd.engine as E
E.call = spy
eng.forward(batch, w); eng.loss_and_backward(w)
torch.cuda.synchronize()
lib.call = real_call
E.call = real_call


def timed(name, d):
    for _ in range(2):
        real_call(name, C.byref(d))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        real_call(name, C.byref(d))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 100.0        # us per call


groups = OrderedDict()
for name, d in calls:
    key = (name, d.M, d.N, d.K, d.trans_a, d.trans_b, d.win_T, d.win_C, d.split_k, d.batch, d.accumulate, d.act)
    groups.setdefault(key, []).append(d)
print("%-60s %8s %8s | best split" % ("kernel M N K ta tb winT winC split batch acc act", "count", "us"))
gain = 0.0
for key, ds in groups.items():
    d = ds[0]
    base = timed(key[0], d)
    tf = 2.0 * d.M * d.N * d.K * max(1, d.batch) / base / 1e6
    if d.act != 0 or tf > 140 or base < 15 or d.batch > 1:
        continue
    res = {}
    keep = d.split_k
    for sk in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64):
        if sk > 1 and d.K // sk < 64:
            continue
        d.split_k = sk
        try:
            res[sk] = timed(key[0], d)
        except Exception as e:          # (a split the kernel refuses)
            res[sk] = float("nan")
    d.split_k = keep
    best = min((v, k) for k, v in res.items() if v == v)
    gain += (base - best[0]) * len(ds)
    print("%-60s %8d %8.1f | sk %2d: %6.1f us   %s" % (" ".join(str(x) for x in key), len(ds), base, best[1], best[0],
                                                      " ".join("%d:%.0f" % (k, v) for k, v in res.items())))
print("sum of (engine's choice - best) over the step: %.1f us" % gain)
