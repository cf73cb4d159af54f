"""Back-to-back timing of the fp32 and bf16 skinny products at the decoder shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_speaker_tts_amd import lib
from tools.microbench import timeit
dev = torch.device("cuda:0")
L = lib.load()
M = 32
for (N, K) in [(4096, 1792), (4096, 2048), (128, 1024)]:
    X = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev) * 0.02
    ks = L.mstts_skinny_fwd_splits(N, K); P = torch.zeros(16, M, N, device=dev)
    f32 = lambda: lib.call("mstts_skinny_fwd", lib.ptr(X), K, lib.ptr(W), N, lib.ptr(P), 0, M, N, K, ks)
    kb = L.mstts_skinny_bf16_fwd_splits(N, K); Wp = torch.zeros(K * N, dtype=torch.int16, device=dev)
    lib.call("mstts_pack_bf16_fwd", lib.ptr(W), N, lib.ptr(Wp), K, N, kb)
    bf = lambda: lib.call("mstts_skinny_fwd_bf16", lib.ptr(X), K, lib.ptr(Wp), lib.ptr(P), 0, M, N, K, kb)
    print("fwd %dx%dx%d: fp32 (ks=%d) %.2f us   bf16 (ks=%d) %.2f us" % (M, K, N, ks, timeit(f32, 500, graph=True), kb, timeit(bf, 500, graph=True)))
for (R, N) in [(1792, 4096), (2048, 4096), (1024, 128)]:
    dG = torch.randn(M, N, device=dev); W = torch.randn(R, N, device=dev) * 0.02
    ns = L.mstts_skinny_bwd_splits(R, N); P = torch.zeros(16, M, R, device=dev)
    f32 = lambda: lib.call("mstts_skinny_bwd", lib.ptr(dG), N, lib.ptr(W), N, lib.ptr(P), 0, M, R, N, ns)
    nb = L.mstts_skinny_bf16_bwd_splits(R, N); Wq = torch.zeros(R * N, dtype=torch.int16, device=dev)
    lib.call("mstts_pack_bf16_bwd", lib.ptr(W), N, lib.ptr(Wq), R, N, nb)
    bf = lambda: lib.call("mstts_skinny_bwd_bf16", lib.ptr(dG), N, lib.ptr(Wq), lib.ptr(P), 0, M, R, N, nb)
    print("bwd %dx%dx%d: fp32 (ns=%d) %.2f us   bf16 (ns=%d) %.2f us" % (M, N, R, ns, timeit(f32, 500, graph=True), nb, timeit(bf, 500, graph=True)))
