"""Every C-ABI call of ONE config-2 train step that is not a contraction or a persistent launch, recorded at the ctypes boundary and replayed one
by one (HIP events, 10 repeats each on the live buffers; outputs are scratch here): count x microseconds per distinct (function, scalar arguments),
sorted by total time - where the step's ~3 ms of streaming work go.   usage: python tools/step_call_profile.py [--config3]"""
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
eng = TrainEngine(dims, device=dev, seed=1234, recurrent_dtype="bf16" if c3 else "f32", gemm_dtype="bf16" if c3 else "f32")
batch = bench.synthetic_batch(dims, 32, 128, 800, 1234, 0, dev)
w = eng.plan(32, 128, 800)
for _ in range(2):
    eng.forward(batch, w); eng.loss_and_backward(w); eng.adam_step()
torch.cuda.synchronize()
SKIP = ("mstts_gemm_f32", "mstts_gemm_bf16", "persistent", "mstts_last_error", "mstts_lstm_seq")
calls = []
real_call = lib.call


def spy(name, *a):
    if not any(s in name for s in SKIP):
        calls.append((name, a))
    return real_call(name, *a)


import multi_speaker_tts_amd.engine as E
import multi_speaker_tts_amd.params as P
mods = [m for m in (lib, E, P) if hasattr(m, "call")]
for m in mods:
    m.call = spy
eng.forward(batch, w); eng.loss_and_backward(w); eng.adam_step()
torch.cuda.synchronize()
for m in mods:
    m.call = real_call


def scalars(a):
    out = []
    for x in a:
        if isinstance(x, (int, float)):
            out.append(x if isinstance(x, int) and abs(x) < (1 << 40) and not (x > (1 << 32)) else "p")
        else:
            out.append("p")
    return tuple(out)


groups = OrderedDict()
for name, a in calls:
    groups.setdefault((name,) + tuple(s for s in scalars(a) if s != "p"), []).append(a)
rows = []
for key, al in groups.items():
    a = al[0]
    try:
        for _ in range(2):
            real_call(key[0], *a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            real_call(key[0], *a)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100.0
    except Exception as ex:
        us = float("nan")
    rows.append((us * len(al), len(al), us, key))
rows.sort(key=lambda r: -(r[0] if r[0] == r[0] else 0))
tot = 0.0
print("%9s %4s %9s  call (scalar arguments)" % ("total us", "n", "us each"))
for t, n, us, key in rows:
    if t == t:
        tot += t
    print("%9.1f %4d %9.1f  %s" % (t, n, us, " ".join(str(k) for k in key)))
print("sum: %.2f ms in %d calls" % (tot / 1e3, sum(r[1] for r in rows)))
