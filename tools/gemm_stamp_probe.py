import os, sys
sys.path.insert(0, "/root/repo")
import torch
from multi_speaker_tts_amd import lib
dev = torch.device("cuda:0")
for (M, N, K, ta) in ((8192, 8192, 8192, 0), (2048, 4096, 12816, 1)):
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn(K, N, device=dev); Cm = torch.zeros(M, N, device=dev)
    st = torch.zeros(8, dtype=torch.int64, device=dev)
    for _ in range(3):
        lib.gemm(A, B, Cm, M, N, K, A.shape[1], N, N, trans_a=bool(ta), bias=st.view(torch.float32))
    torch.cuda.synchronize()
    v = st.cpu().tolist()
    print(M, N, K, "consumer: wait %d of %d cycles over %d tiles (%.0f + %.0f per tile) | producer: wait %d of %d (%.0f + %.0f per tile), of which waiting for the older register set on every second tile %.0f" % (
        v[0], v[1], v[2], v[0] / v[2], (v[1] - v[0]) / v[2], v[4], v[5], v[4] / v[6], (v[5] - v[4]) / v[6], v[7] / (v[6] / 2)))
