"""Every mstts_gemm_f32 call of ONE config-2 train step, recorded at the ctypes boundary (descriptor contents) and then replayed one by one
(HIP events, 10 repeats each on the live buffers): count x microseconds x TFLOP/s per distinct call, sorted by total time.
usage: python tools/gemm_step_profile.py [--config3]"""
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
        d = a[0]._obj
        cp = lib.GemmDesc()
        C.memmove(C.byref(cp), C.byref(d), C.sizeof(lib.GemmDesc))
        calls.append((name, cp))
    return real_call(name, *a)


lib.call = spy
import multi_speaker_tts_amd.engine as E
E.call = spy
eng.forward(batch, w); eng.loss_and_backward(w)
torch.cuda.synchronize()
lib.call = real_call
E.call = real_call

groups = OrderedDict()
for name, d in calls:
    key = (name, d.M, d.N, d.K, d.trans_a, d.trans_b, d.win_T, d.win_C, d.split_k, d.batch, d.accumulate, d.act)
    groups.setdefault(key, []).append(d)
rows = []
for key, ds in groups.items():
    d = ds[0]
    for _ in range(2):
        real_call(key[0], C.byref(d))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        real_call(key[0], C.byref(d))
    e1.record()
    torch.cuda.synchronize()
    us = 100.0 * e0.elapsed_time(e1)
    fl = 2.0 * d.M * d.N * d.K * max(1, d.batch)
    rows.append((us * len(ds), len(ds), us, fl / us / 1e6, key))
rows.sort(reverse=True)
print("%9s %3s %9s %8s  %s" % ("total us", "n", "us each", "TFLOP/s", "kernel M N K ta tb winT winC split batch acc act"))
for tot, n, us, tf, key in rows:
    print("%9.1f %3d %9.1f %8.1f  %s" % (tot, n, us, tf, " ".join(str(k) for k in key)))
print("sum: %.2f ms in %d calls" % (sum(r[0] for r in rows) / 1e3, len(calls)))
