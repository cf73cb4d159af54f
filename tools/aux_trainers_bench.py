"""Step time of the two auxiliary trainers at the reference's sizes (synthetic patterns)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multi_speaker_tts_amd.Taco1_Mel_to_Spect import Mel_to_Spect
from multi_speaker_tts_amd.Speaker_Embedding import Speaker_Embedding
dev = "cuda:0"
m = Mel_to_Spect(device=dev)
pat = m.Synthetic_Pattern(batch_Size=128, length=400)
for _ in range(2):
    m.Train_Step(pat)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    r = m.Train_Step(pat)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("Taco1 mel->spectrogram trainer, batch 128 x 400 frames: %.1f ms/step, %.0f frames/s (loss %.4f)" % (dt * 1e3, 128 * 400 / dt, r["Loss"]))
del m
s = Speaker_Embedding(device=dev)
pat = s.Synthetic_Pattern(seed=1)
for _ in range(2):
    s.Train_Step(pat)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    r = s.Train_Step(pat)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
N, T = pat["Mel"].shape[:2]
print("GE2E speaker-encoder trainer, %d utterances x %d frames: %.1f ms/step, %.0f utterances/s (loss %.4f)" % (N, T, dt * 1e3, N / dt, r["Loss"]))
