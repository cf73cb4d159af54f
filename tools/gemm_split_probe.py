import os, sys
sys.path.insert(0, "/root/repo")
import torch
from multi_speaker_tts_amd import lib
from tools.microbench import timeit
dev = torch.device("cuda:0")
for (m, n, k) in [(1792, 4096, 25632), (2048, 4096, 25632), (1024, 128, 25632), (256, 4096, 25632), (2560, 512, 25632), (768, 128, 4096)]:
    A = torch.randn(k, m, device=dev); Bm = torch.randn(k, n, device=dev); Cm = torch.zeros(m, n, device=dev)
    for sp in (1, 2, 3, 4, 6, 8, 16):
        f = lambda: lib.gemm(A, Bm, Cm, m, n, k, m, n, n, trans_a=True, split_k=sp)
        us = timeit(f, 10)
        print("TN %dx%dx%d split %2d: %8.1f us  %6.1f TFLOP/s" % (m, n, k, sp, us, 2.0 * m * n * k / us / 1e6))
