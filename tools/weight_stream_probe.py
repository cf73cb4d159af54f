"""Where do the recurrent weights stream from?  skinny_fwd (32 x 1792 x 4096, 29.4 MB of fp32 weights per launch) back to back
while rotating over n distinct weight sets: 1 set fits the 8 x 4 MB L2s, 4-8 sets (117-235 MB) fit the 256 MB Infinity Cache,
16-32 sets (470-940 MB) only HBM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_speaker_tts_amd import lib
from tools.microbench import timeit
dev = torch.device("cuda:0")
L = lib.load()
M, N, K = 32, 4096, 1792
X = torch.randn(M, K, device=dev)
ks = L.mstts_skinny_fwd_splits(N, K)
P = torch.zeros(ks, M, N, device=dev)
for nset in (1, 2, 4, 8, 16, 32):
    Ws = [torch.randn(K, N, device=dev) * 0.02 for _ in range(nset)]
    i = [0]
    def f():
        i[0] = (i[0] + 1) % nset
        lib.call("mstts_skinny_fwd", lib.ptr(X), K, lib.ptr(Ws[i[0]]), N, lib.ptr(P), 0, M, N, K, ks)
    us = timeit(f, 640, graph=True)
    print("%2d weight sets (%4.0f MB): %.2f us/launch  -> %.2f TB/s incl. fixed cost" % (nset, nset * K * N * 4 / 1e6, us, K * N * 4 / us / 1e6))
    del Ws
