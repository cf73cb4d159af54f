"""Audio.melspectrogram timing and roofline: 10 s of 16 kHz audio -> [801, 80] mel, HIP events around N calls.
Algorithmic bytes: wav 4 B/sample in, mel 80 x 4 B/frame out (the tables - window, twiddles, 80 x 1025 filterbank, 0.34 MB - stay in L2).
Built as ONE launch (workgroup per frame, 1024-point complex FFT in LDS); the older DFT-as-GEMM form is timed beside it."""
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
frames = 1 + y.numel() // 200


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n_calls):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n_calls


# (1) one-launch FFT-in-LDS path, kernel only (tables and offsets prepared once, like a feeder would)
n_fft, hop, win, hann, tw, fb, rng = Audio._fft_constants(1025, 12.5, 50, 80, sr, str(dev))
woff = torch.tensor([0, y.numel()], dtype=torch.int64, device=dev)
foff = torch.tensor([0, frames], dtype=torch.int64, device=dev)
mel = torch.empty(frames, 80, device=dev)
from multi_speaker_tts_amd import lib
fft_call = lambda: lib.call("mstts_stft_fft", lib.ptr(y), lib.ptr(woff), lib.ptr(foff), 1, 0.97, lib.ptr(hann), lib.ptr(tw), lib.ptr(fb), lib.ptr(rng),
                            n_fft, hop, win, 80, 4.0, 20.0, lib.ptr(mel), None, frames, None, None, 0.0, 0)
us_fft = timed(fft_call)
# 32 utterances per launch (a feeder batch)
yy = y.repeat(32)
woff32 = torch.arange(33, dtype=torch.int64, device=dev) * y.numel()
foff32 = torch.arange(33, dtype=torch.int64, device=dev) * frames
mel32 = torch.empty(32 * frames, 80, device=dev)
us_fft32 = timed(lambda: lib.call("mstts_stft_fft", lib.ptr(yy), lib.ptr(woff32), lib.ptr(foff32), 32, 0.97, lib.ptr(hann), lib.ptr(tw), lib.ptr(fb),
                                  lib.ptr(rng), n_fft, hop, win, 80, 4.0, 20.0, lib.ptr(mel32), None, 32 * frames, None, None, 0.0, 0))
# (2) the DFT-as-GEMM form (5 launches) through the Python surface
us_gemm = timed(lambda: Audio.melspectrogram(y, 1025, 12.5, 50, 80, sr, max_abs_value=4, device=dev, return_tensor=True, use_fft=False))
alg = y.numel() * 4 + frames * 80 * 4
tables_fft = (800 + 2 * 2048 + 80 * 1025) * 4
flop_fft = frames * (5 * 1024 * 10 + 8 * 1025 + 2 * 2100)   # 5 N log2 N convention
print(json.dumps({"kernel": "mstts_stft_fft (one launch: window + real FFT in LDS + magnitude + mel + dB/normalise)", "audio_seconds": secs, "frames": frames,
                  "us_per_call": us_fft, "x_realtime": secs / (us_fft * 1e-6), "algorithmic_bytes": alg, "constant_table_bytes": tables_fft,
                  "hbm_roofline_us_at_8TBs": alg / 8e12 * 1e6, "achieved_GBs_on_algorithmic": alg / (us_fft * 1e-6) / 1e9,
                  "frac_of_8TBs": alg / (us_fft * 1e-6) / 8e12, "fft_gflops": flop_fft / (us_fft * 1e-6) / 1e9,
                  "batch32": {"us_per_call": us_fft32, "x_realtime": 32 * secs / (us_fft32 * 1e-6), "achieved_GBs_on_algorithmic": 32 * alg / (us_fft32 * 1e-6) / 1e9,
                              "frac_of_8TBs": 32 * alg / (us_fft32 * 1e-6) / 8e12},
                  "dft_gemm_form_us_per_call": us_gemm}))
