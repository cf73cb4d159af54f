"""Barrier-time stamps of the split GEMM's consumer waves (build with MSTTS_EXTRA_HIPCC_FLAGS=-DGS_STAMP): per K-tile, cycles a consumer wave of
workgroup 0 spends inside the barrier and outside it (1 536 = the MFMA issue time of a tile).  usage: python tools/gemm_stamp_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    print(M, N, K, "consumer: waits %.0f + busy %.0f cycles per K-tile over %d tiles; shader clock while it ran: %.0f MHz (cycle counter / 100 MHz wall clock)"
          % (v[0] / max(v[2], 1), (v[1] - v[0]) / max(v[2], 1), v[2], v[1] / max(v[3], 1) * 100.0))
