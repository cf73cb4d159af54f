"""Timing of the single-launch attention step: graph replay (frozen epoch = no waiting) and eager with fresh epochs."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_speaker_tts_amd import lib
from tools.microbench import timeit
dev = torch.device("cuda:0")
B, T, M, A, CH, KS = 32, 128, 768, 128, 32, 31
rn = lambda *sh: torch.randn(*sh, device=dev)
keys, values = rn(B, T, A), rn(B, T, M)
conv_k, conv_b, dense_k, sw, sb = rn(KS, 1, CH) * .3, rn(CH) * .1, rn(CH, A) * .3, rn(A) * .5, rn(A) * .1
loc_k, loc_b = torch.zeros(KS, A, device=dev), torch.zeros(A, device=dev)
c = lib.LsaConst()
c.B, c.T, c.A, c.M, c.KS, c.CH = B, T, A, M, KS, CH
c.keys, c.values, c.lengths = lib.ptr(keys), lib.ptr(values), None
c.conv_k, c.conv_b, c.dense_k, c.score_w, c.score_b = lib.ptr(conv_k), lib.ptr(conv_b), lib.ptr(dense_k), lib.ptr(sw), lib.ptr(sb)
lib.call("mstts_lsa_fold_location", c.conv_k, c.conv_b, c.dense_k, lib.ptr(loc_k), lib.ptr(loc_b), KS, CH, A)
c.loc_k, c.loc_b = lib.ptr(loc_k), lib.ptr(loc_b)
q = rn(8, B, A); cum = torch.rand(B, T, device=dev); al = torch.zeros(B, T, device=dev); cn = torch.zeros(B, T, device=dev); cx = torch.zeros(B, M, device=dev)
en = torch.zeros(B, T, device=dev)
gran = torch.zeros(B * T + 1, dtype=torch.int64, device=dev)
ep = [0]
def step():
    ep[0] += 1
    lib.call("mstts_lsa_step_fwd", C.byref(c), lib.ptr(q), 8, B * A, None, lib.ptr(cum), lib.ptr(al), lib.ptr(cn), lib.ptr(cx), M, None, 0, None, lib.ptr(gran), ep[0])
def two():
    lib.call("mstts_lsa_energy_fwd", C.byref(c), lib.ptr(q), 8, B * A, None, lib.ptr(cum), lib.ptr(en))
    lib.call("mstts_lsa_context_fwd", C.byref(c), lib.ptr(en), lib.ptr(cum), lib.ptr(al), lib.ptr(cn), lib.ptr(cx), M, None, 0)
print("dbg=%s" % os.environ.get("MSTTS_LSA_STEP_DEBUG", "0"))
print("  fused, graph replay (frozen epoch): %.2f us" % timeit(step, 500, graph=True))
print("  fused, eager fresh epochs          : %.2f us" % timeit(step, 2000, graph=False))
print("  two launches, graph replay         : %.2f us" % timeit(two, 500, graph=True))
print("  two launches, eager                : %.2f us" % timeit(two, 2000, graph=False))
print("  time-outs:", int(gran[-1]))
