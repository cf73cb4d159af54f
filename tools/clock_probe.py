"""Sustained clock under the fp32 MFMA GEMM: runs the dW1 contraction back to back and samples rocm-smi's sclk while it runs."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_speaker_tts_amd import lib
dev = torch.device("cuda:0")
M, N, K = 2048, 4096, 25632
A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); C = torch.zeros(M, N, device=dev)
samples = []
def poll():
    for _ in range(12):
        try:
            out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            samples.append([l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l])
        except Exception as e:
            samples.append([repr(e)])
        time.sleep(0.4)
t = threading.Thread(target=poll); t.start()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
n = 1500
for _ in range(n):
    lib.gemm(A, B, C, M, N, K, M, N, N, trans_a=True)
e1.record(); torch.cuda.synchronize()
t.join()
ms = e0.elapsed_time(e1) / n
print("dW1 %d x %d x %d: %.1f us, %.1f TFLOP/s" % (M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
for s in samples[:12]:
    print(s)
