"""Attention step (mstts_lsa_step_fwd, single launch with in-launch energy exchange) over batch sizes, T = 128, M = 768:
algorithmic bytes / launch time against the 8 TB/s HBM peak, (a) as a link of a dependent chain - each launch consumes the previous
launch's cumulative alignment, as the decoder loop does - and (b) back to back on independent buffers.  Shows what part of the headline
fraction (batch 32) is the fixed cost of a dependent launch and what the kernel streams once that cost is amortised."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from multi_speaker_tts_amd import lib

dev = torch.device("cuda:0")
T, M, A, CH, KS = 128, 768, 128, 32, 31
g = torch.Generator(device="cpu").manual_seed(3)
rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(dev).contiguous()
out = []
for B in (32, 64, 128, 256):
    keys, values = rn(B, T, A), rn(B, T, M)
    conv_k, conv_b, dense_k, sw, sb_ = rn(KS, 1, CH, sc=0.3), rn(CH, sc=0.1), rn(CH, A, sc=0.3), rn(A, sc=0.5), rn(A, sc=0.1)
    loc_k, loc_b, loc_kt = torch.zeros(KS, A, device=dev), torch.zeros(A, device=dev), torch.zeros(A, 36, device=dev)
    c = lib.LsaConst()
    c.B, c.T, c.A, c.M, c.KS, c.CH = B, T, A, M, KS, CH
    c.keys, c.values, c.lengths = lib.ptr(keys), lib.ptr(values), None
    c.conv_k, c.conv_b, c.dense_k, c.score_w, c.score_b = lib.ptr(conv_k), lib.ptr(conv_b), lib.ptr(dense_k), lib.ptr(sw), lib.ptr(sb_)
    lib.call("mstts_lsa_fold_location", c.conv_k, c.conv_b, c.dense_k, lib.ptr(loc_k), lib.ptr(loc_b), KS, CH, A)
    lib.call("mstts_lsa_filter_by_unit", lib.ptr(loc_k), lib.ptr(loc_kt), KS, A)
    c.loc_k, c.loc_b, c.loc_kt = lib.ptr(loc_k), lib.ptr(loc_b), lib.ptr(loc_kt)
    N = 200
    q = rn(B, A)
    cum = torch.zeros(2, B, T, device=dev)
    al, cx = torch.zeros(B, T, device=dev), torch.zeros(B, M, device=dev)
    gran = torch.zeros(B * T + 1, dtype=torch.int64, device=dev)
    # algorithmic bytes of one launch: keys + values + cumulative alignment in/out + alignment + context + query + filter
    bytes_ = 4 * (B * T * A + B * T * M + 3 * B * T + B * M + B * A + KS * A)

    def run(n, e0):
        for i in range(n):
            lib.call("mstts_lsa_step_fwd", C.byref(c), lib.ptr(q), 1, 0, None, lib.ptr(cum[i & 1]), lib.ptr(al), lib.ptr(cum[(i + 1) & 1]),
                     lib.ptr(cx), M, None, 0, None, lib.ptr(gran), e0 + i + 1)
    run(20, 0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(N, 20); b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / N
    assert int(gran[B * T]) == 0
    row = {"B": B, "bytes_per_launch": bytes_, "dependent_chain_us": round(us, 2), "achieved_GBps": round(bytes_ / us / 1e3, 1),
           "frac_of_8TBps": round(bytes_ / us / 1e3 / 8000.0, 3)}
    out.append(row)
    print(json.dumps(row))
