"""Audio.melspectrogram timing and roofline: 10 s of 16 kHz audio -> [80, 801] mel, HIP events around N calls.
Algorithmic bytes (what a fused kernel would have to move): wav 4 B/sample in, mel 80 x 4 B/frame out, plus the constant tables once
(windowed DFT basis 800 x 2056 fp32, mel filterbank 1028 x 80 fp32; L2-resident across calls).  Arithmetic as built: a dense windowed
DFT on the fp32 matrix cores, 2 * 800 * 2056 + 2 * 1028 * 80 FLOP per frame."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from multi_speaker_tts_amd import Audio

dev = torch.device("cuda:0")
sr, secs, n_calls = 16000, 10.0, 50
g = np.random.default_rng(0)
y = torch.tensor((0.3 * g.normal(size=int(sr * secs))).astype(np.float32), device=dev)
for _ in range(3):
    m = Audio.melspectrogram(y, 1025, 12.5, 50, 80, sr, max_abs_value=4, device=dev, return_tensor=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n_calls):
    m = Audio.melspectrogram(y, 1025, 12.5, 50, 80, sr, max_abs_value=4, device=dev, return_tensor=True)
e1.record()
torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / n_calls
frames = 1 + y.numel() // 200
alg = y.numel() * 4 + frames * 80 * 4
tables = 800 * 2056 * 4 + 1028 * 80 * 4
flop = frames * (2 * 800 * 2056 + 2 * 1028 * 80)
print(json.dumps({"kernel": "mstts_stft_mel (preemph/pad + DFT GEMM + magnitude + mel GEMM + dB/normalise: 5 launches)", "audio_seconds": secs, "frames": frames,
                  "us_per_call": us, "x_realtime": secs / (us * 1e-6), "algorithmic_bytes": alg, "constant_table_bytes": tables,
                  "hbm_roofline_us_at_8TBs": (alg + tables) / 8e12 * 1e6, "achieved_GBs_on_algorithmic_plus_tables": (alg + tables) / (us * 1e-6) / 1e9,
                  "frac_of_8TBs": (alg + tables) / (us * 1e-6) / 8e12, "dft_gemm_tflops": flop / (us * 1e-6) / 1e12}))
