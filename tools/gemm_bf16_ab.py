"""Times config 3's large contractions on mstts_gemm_bf16 with the 256 x 256-tile kernel on and off (HIP events, 20 repeats).
usage: python tools/gemm_bf16_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_speaker_tts_amd import lib

dev = torch.device("cuda:0")
SHAPES = [(2048, 4096, 25632, 1, 0, None, 2), (1792, 4096, 25632, 1, 0, None, 4), (25632, 512, 2560, 0, 0, (801, 512, 2), 1), (2560, 512, 25632, 1, 0, (801, 512, 2), 9),
          (25632, 256, 4096, 0, 1, None, 1), (256, 4096, 25632, 1, 0, None, 12), (4096, 512, 2560, 0, 0, (128, 512, 2), 1), (25632, 4096, 256, 0, 0, None, 1), (8192, 8192, 8192, 0, 0, None, 1)]
for M, N, K, ta, tb, win, sk in SHAPES:
    if win:
        rows = K if ta else M
        A = torch.randn(rows, win[1], device=dev)
    else:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    Cm = torch.zeros(M, N, device=dev)
    out = []
    for big in (1, 0):
        lib.load().mstts_gemm_bf16_big(big)
        f = lambda: lib.gemm(A, B, Cm, M, N, K, A.shape[1], B.shape[1], N, trans_a=bool(ta), trans_b=bool(tb), win=win, split_k=sk, accumulate=(sk > 1), bf16=True)
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        out.append("%8.1f us %6.1f TF" % (us, 2.0 * M * N * K / us * 1e-6))
    lib.load().mstts_gemm_bf16_big(1)
    print("%6d %5d %6d ta%d tb%d win %-15s sk%-2d | 256x256 tile %s | 128x128 tile %s" % (M, N, K, ta, tb, win, sk, out[0], out[1]), flush=True)
