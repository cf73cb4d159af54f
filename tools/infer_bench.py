"""Free-running decoder throughput (BASELINE.json configs[3] shape: batch 16, mixed-length texts)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multi_speaker_tts_amd.inference import InferEngine
from multi_speaker_tts_amd.params import Dims
dev = torch.device("cuda:0")
d = Dims()
eng = InferEngine(d, device=dev, chunk=100)
bt, bo = eng.P("decoder/decoder/linear_projection/dense/bias")
bt.view(-1)[bo + d.n_mel] = -100.0          # never raise the stop flag: run to max_steps
B, Te, S = int(os.environ.get("B", 16)), 128, int(os.environ.get("S", 400))
g = np.random.default_rng(0)
tok = g.integers(2, d.n_tok, size=(B, Te)).astype(np.int32); tok[:, 0] = 0
lens = g.integers(40, Te + 1, size=B).astype(np.int32); lens[0] = Te
for b in range(B):
    tok[b, lens[b] - 1:] = 1
spk = g.normal(0, 1, (B, d.spk)); spk = (spk / np.sqrt((spk ** 2).sum())).astype(np.float32)
t = lambda a: torch.from_numpy(a).to(dev).contiguous()
values, keys = eng.encoder(t(tok), t(lens), t(spk))
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lin, stop, al, n = eng.decode(values, keys, t(lens), seed=1, max_steps=S)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("decode: %d steps, batch %d: %.1f ms  (%.1f us/step, %.0f mel-frames/s)" % (n, B, dt * 1e3, dt / n * 1e6, B * n / dt))
# whole inference forward of BASELINE configs[3] (speaker encoder on 5 x 64-frame mels per utterance, text encoder, free-running decoder to
# max_steps, postnet, Taco1 mel -> spectrogram), results copied to the host as MSTTS_SV.Inference does
spk_mel = g.normal(0, 1, (5 * B, 64, d.n_mel)).astype(np.float32)
pattern = {"Token": tok, "Token_Length": lens, "Speaker_Embedding_Mel": spk_mel}
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = eng.forward(pattern, seed=1, max_steps=S, with_vocoder=True)
    dt = time.perf_counter() - t0
    n = res["Linear"].shape[1]
    print("forward (speaker encoder + Tacotron2 + Taco1 vocoder, host copies included): %d frames x batch %d: %.1f ms  (%.0f mel-frames/s)"
          % (n, B, dt * 1e3, B * n / dt))
