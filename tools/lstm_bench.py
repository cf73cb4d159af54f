"""Encoder BiLSTM recurrence (B = 32, T = 128, H = 256, both directions): the persistent launches (csrc/persist_lstm.hip) against the
launch-per-step pair drivers, forward and BPTT, microseconds per sequence (HIP events, 20 repeats)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from multi_speaker_tts_amd import lib
from tests import test_gpu_persist_lstm as TL

dev = torch.device("cuda:0")
B, T = 32, 128
st = TL._setup(dev, B, T, seed=1)
st["lens"][:] = T


def timed(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


L = lib.load()
o = TL._run_fwd(dev, st, True)
ref = TL._run_fwd(dev, st, False)
extra = {"gates_" + dr: torch.empty(L.mstts_lstm_seq_ws_floats(B, TL.H, 0), device=dev) for dr in ("fw", "bw")}
hp = {dr: torch.zeros(2 * L.mstts_cell_act_floats(B, TL.H), device=dev) for dr in ("fw", "bw")}
whp = {}
for dr in ("fw", "bw"):
    whp[dr] = torch.empty(TL.H * 4 * TL.H, device=dev)
    lib.call("mstts_pack_cell_fwd", lib.ptr(st["wh_" + dr]), 4 * TL.H, lib.ptr(whp[dr]), TL.H, TL.H)
qs_p = TL._fwd_descs(st, o, None)
qs_s = TL._fwd_descs(st, ref, extra)
for q, dr in zip(qs_s, ("fw", "bw")):
    q.wh_p, q.h_p = lib.ptr(whp[dr]), lib.ptr(hp[dr])
pk, xch, ctrl, hist = o["pk"], o["xch"], o["ctrl"], o["hist"]
f_p = timed(lambda: lib.call("mstts_lstm_seq_fwd_pair_persistent", C.byref(qs_p[0]), C.byref(qs_p[1]), lib.ptr(pk["fw"]), lib.ptr(pk["bw"]), lib.ptr(xch), lib.ptr(ctrl), lib.ptr(hist)))
f_s = timed(lambda: lib.call("mstts_lstm_seq_fwd_pair", C.byref(qs_s[0]), C.byref(qs_s[1])))
if os.environ.get("LSTM_PROF"):
    torch.cuda.synchronize()
    stv = xch[2 * 4 * 8192: 2 * 4 * 8192 + 12].view(torch.int64).cpu().numpy()
    print("forward stamps (us / step, workgroup 0): top %.2f  gather %.2f  mfma+lds %.2f  barrier %.2f  update+publish %.2f  history %.2f" % tuple(stv[i] * 0.01 / T for i in range(6)))
print("forward : persistent %.1f us incl. pack / unpack kernels (%.2f us / step), launch per step %.1f us (%.2f us / step)" % (f_p, f_p / T, f_s, f_s / T))
import types
rb = {}


def bwd(persistent):
    return TL._run_bwd(dev, st, o if persistent else ref, persistent)


import time
for name, pers in (("persistent", True), ("launch per step", False)):
    bwd(pers)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        bwd(pers)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10 * 1e6
    print("BPTT    : %s %.1f us per sequence incl. host set-up (%.2f us / step)" % (name, dt, dt / T))
