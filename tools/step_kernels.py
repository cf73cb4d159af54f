"""Isolated timing of the per-step decoder kernels (graph-replayed back-to-back)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multi_speaker_tts_amd import lib
from multi_speaker_tts_amd.engine import TrainEngine
from multi_speaker_tts_amd.params import Dims
from tools.microbench import timeit

dev = torch.device("cuda:0")
dims = Dims()
eng = TrainEngine(dims, device=dev)
L = 40
batch = bench.synthetic_batch(dims, 32, 128, L, 1234, 0, dev)
w = eng.train_step(batch)
torch.cuda.synchronize()
d = w.dec
B, H, M, A, T = 32, 1024, 768, 128, 128
st = 5
BH = B * H
W0, W1, WP = M + H, 2 * H, H + M
f4 = lambda t, off=0: lib.ptr(t, off)
p = lib.LstmPointFwd()
p.B, p.H, p.gates_h, p.gates_parts, p.gates_pstride = B, H, lib.ptr(w.gates_ws), 4, 4 * BH
p.xw, p.xw_sb = lib.ptr(w.xw0, st * 4 * BH), 4 * H
p.c_prev, p.h_prev, p.h_prev_ld = lib.ptr(w.c0, st * BH), lib.ptr(w.in0, st * B * W0 + M), W0
p.zc, p.zh, p.zoneout = lib.ptr(w.masks["dec_zc_0"], st * BH), lib.ptr(w.masks["dec_zh_0"], st * BH), 0.1
p.out, p.out_sb = lib.ptr(w.in1, st * B * W1), W1
p.c_next, p.h_next, p.h_next_ld = lib.ptr(w.c0, (st + 1) * BH), lib.ptr(w.in0, (st + 1) * B * W0 + M), W0
p.acts_out, p.c_raw = lib.ptr(w.acts0, st * 4 * BH), lib.ptr(w.craw0, st * BH)
print("lstm_point_fwd (4 slabs)      : %.2f us" % timeit(lambda: lib.call("mstts_lstm_point_fwd", C.byref(p)), 1000, graph=True))
p.gates_parts = 1
print("lstm_point_fwd (1 slab)       : %.2f us" % timeit(lambda: lib.call("mstts_lstm_point_fwd", C.byref(p)), 1000, graph=True))
en = lambda parts: lib.call("mstts_lsa_energy_fwd", C.byref(d.lsa), lib.ptr(w.q_ws), parts, B * A, lib.ptr(w.q_hist, st * B * A),
                            lib.ptr(w.cum_hist, st * B * T), lib.ptr(w.energy_ws))
print("lsa_energy (16 q slabs)       : %.2f us" % timeit(lambda: en(16), 1000, graph=True))
print("lsa_energy (1 q slab)         : %.2f us" % timeit(lambda: en(1), 1000, graph=True))
cx = lambda: lib.call("mstts_lsa_context_fwd", C.byref(d.lsa), lib.ptr(w.energy_ws), lib.ptr(w.cum_hist, st * B * T), lib.ptr(w.align_hist, st * B * T),
                      lib.ptr(w.cum_hist, (st + 1) * B * T), lib.ptr(w.in0, (st + 1) * B * W0), W0, lib.ptr(w.pj, st * B * WP + H), WP)
print("lsa_context                   : %.2f us" % timeit(cx, 1000, graph=True))
G = torch.zeros(2, B, T, device=dev); da = torch.zeros(B, T, device=dev); df = torch.zeros(2, B, T, 32, device=dev)
dal = lambda: lib.call("mstts_lsa_dalign_bwd", C.byref(d.lsa), lib.ptr(w.d_pj, st * B * WP + H), WP, lib.ptr(w.d_in0, (st + 1) * B * W0), W0, w.d_in0_parts,
                       (L + 1) * B * W0, lib.ptr(G), lib.ptr(df), lib.ptr(G, B * T), lib.ptr(da))
print("lsa_dalign                    : %.2f us" % timeit(dal, 1000, graph=True))
de = torch.zeros(B, T, device=dev); dq = torch.zeros(B, A, device=dev)
den = lambda: lib.call("mstts_lsa_denergy_bwd", C.byref(d.lsa), lib.ptr(w.align_hist, st * B * T), lib.ptr(da), lib.ptr(w.q_hist, st * B * A),
                       lib.ptr(w.cum_hist, st * B * T), lib.ptr(de), lib.ptr(dq), lib.ptr(df, B * T * 32))
print("lsa_denergy                   : %.2f us" % timeit(den, 1000, graph=True))
X = torch.randn(B, H, device=dev); Wq = torch.randn(H, A, device=dev)
ks = lib.load().mstts_skinny_fwd_splits(A, H)
Pq = torch.zeros(ks, B, A, device=dev)
print("skinny_fwd q (ks=%d)           : %.2f us" % (ks, timeit(lambda: lib.call("mstts_skinny_fwd", lib.ptr(X), H, lib.ptr(Wq), A, lib.ptr(Pq), 0, B, A, H, ks), 1000, graph=True)))
