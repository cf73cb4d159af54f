"""Times a few of the train step's contractions with the six-product bf16 split switched on and off (HIP events, 20 repeats).
usage: python tools/gemm_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_speaker_tts_amd import lib

dev = torch.device("cuda:0")
SHAPES = [(8192, 8192, 8192, 0, 0, None, 1), (2048, 4096, 25632, 1, 0, None, 2)] if "--quick" in sys.argv else [  # M, N, K, ta, tb, win, split_k
    (2048, 4096, 25632, 1, 0, None, 2), (25632, 512, 2560, 0, 0, (801, 512, 2), 1), (2560, 512, 25632, 1, 0, (801, 512, 2), 9),
    (25632, 256, 4096, 0, 1, None, 1), (4096, 512, 2560, 0, 0, (128, 512, 2), 1), (25632, 80, 2560, 0, 0, (801, 512, 2), 1), (8192, 8192, 8192, 0, 0, None, 1)]
for M, N, K, ta, tb, win, sk in SHAPES:
    if win:
        rows = K if ta else M
        A = torch.randn(rows, win[1], device=dev)
    else:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    Cm = torch.zeros(M, N, device=dev)
    out = []
    for mode in (2, 1, 0):                       # 2: split, 256 x 256 x 16 tiles where they apply; 1: split, 128 x 128 x 32 only; 0: f32-input MFMA
        lib.call("mstts_gemm_split3", 1 if mode else 0)
        lib.load().mstts_gemm_split_big(1 if mode == 2 else 0)
        f = lambda: lib.gemm(A, B, Cm, M, N, K, A.shape[1], B.shape[1], N, trans_a=bool(ta), trans_b=bool(tb), win=win, split_k=sk, accumulate=(sk > 1))
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        out.append("%8.1f us %6.1f TF" % (us, 2.0 * M * N * K / us * 1e-6))
    lib.call("mstts_gemm_split3", 1); lib.load().mstts_gemm_split_big(1)
    print("%6d %5d %6d ta%d tb%d win %-15s sk%-2d | split big %s | split %s | f32 mfma %s" % (M, N, K, ta, tb, win, sk, out[0], out[1], out[2]), flush=True)
