"""Arena aliasing stress: train steps over RANDOM batch shapes at the reference widths with the arena poisoned (NaN over a set's whole extent
whenever a set of another shape is activated) - any kernel that reads memory its own step did not write shows up as a non-finite loss at the
first step of the offending shape.  usage: shape_stress.py [steps] [seed]"""
import os
import sys
os.environ["MSTTS_ARENA_POISON"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from multi_speaker_tts_amd.engine import TrainEngine
from multi_speaker_tts_amd.params import Dims

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = np.random.default_rng(seed)
dev = torch.device("cuda:0")
d = Dims()
eng = TrainEngine(d, device=dev, seed=1234, arena_hint=(32, 256, 721))
bad = []
prev = None
for i in range(steps):
    B = 32 if g.random() < 0.8 else int(g.integers(1, 33))
    Te = int(g.integers(10, 200))
    L = int(g.integers(40, 720))
    tok = g.integers(2, d.n_tok, size=(B, Te)).astype(np.int32); tok[:, 0] = 0
    tl = g.integers(max(2, Te // 2), Te + 1, size=B).astype(np.int32); tl[g.integers(0, B)] = Te
    for b in range(B):
        tok[b, tl[b] - 1:] = 1
    mel = np.clip(g.normal(0, 1.5, size=(B, L, d.n_mel)), -4, 4).astype(np.float32)
    ml = g.integers(max(1, L // 2), L + 1, size=B).astype(np.int32); ml[g.integers(0, B)] = L
    for b in range(B):
        mel[b, ml[b]:] = 0
    spk = g.normal(0, 1, size=(B, d.spk)); spk = (spk / np.sqrt((spk ** 2).sum())).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev).contiguous()
    batch = {"Token": t(tok), "Token_Length": t(tl), "Mel": t(mel), "Mel_Length": t(ml), "Speaker_Embedding": t(spk)}
    w = eng.train_step(batch)
    sc = eng.scalars(w)
    ok = all(np.isfinite(v) for v in sc.values()) and bool(torch.isfinite(eng.params.train).all())
    if not ok:
        bad.append((i, (B, Te, L), prev, sc))
        print("NON-FINITE at step", i, "shape", (B, Te, L), "previous shape", prev, sc, flush=True)
        break
    prev = (B, Te, L)
print("steps", i + 1, "non-finite", len(bad), "fallbacks", eng.persist_fallbacks, eng.persist_bwd_fallbacks, eng.persist_enc_fallbacks,
      "non-persistent plans", eng.non_persistent_plans, "arena generation", eng._arena.generation, "last loss", sc["Loss"])
