"""WaveGlow inference throughput at the reference sizes (BASELINE.json configs[4]: batch 4 chunks x 40 mel frames)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multi_speaker_tts_amd.waveglow import WaveGlowEngine, WGDims
dev = torch.device("cuda:0")
d = WGDims()
eng = WaveGlowEngine(d, device=dev)
eng.split_in = int(os.environ.get("SPLIT_IN", 0))
N, T = int(os.environ.get("N", 4)), 40
mel = np.clip(np.random.default_rng(0).normal(0, 1.5, (N, T, d.n_mel)), -4, 4).astype(np.float32)
L = (T - 1) * d.up_stride + d.up_k
rows = N * L // d.groups
flop = 0
for f in range(d.flows):
    c = d.channels(f)
    per_row = 2 * (c // 2) * d.ch + 2 * d.groups * d.n_mel * d.layers * 2 * d.ch + d.layers * 2 * d.k * d.ch * 2 * d.ch \
              + (d.layers - 1) * 2 * d.ch * 2 * d.ch + 2 * d.ch * d.ch + 2 * d.ch * c
    flop += per_row * rows
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    w = eng.infer(mel, seed=it)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("batch %d x %d frames -> %d samples: %.1f ms, %.0f samples/s (%.1fx real time at 22.05 kHz), %.1f TFLOP/s"
          % (N, T, N * L, dt * 1e3, N * L / dt, N * L / dt / 22050, flop / dt / 1e12))
