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
print("persistent launches %d, fallbacks %d, status (arrivals, abort, left in order, finished rows, steps) %r" % (eng.persist_infer_launches, eng.persist_infer_fallbacks, eng.persist_infer_status))
if eng.persist_infer_launches:
    # in-kernel stage stamps (PROF instantiation, 100 MHz wall clock, mean over the 256 workgroups and all steps)
    NAMES = ["loop top", "wait ctx", "ctx product", "wait prenet", "prenet product, publish P0", "shadow: h1 product", "wait partials 0", "cell-0 update, publish",
             "wait m0", "m0 product, publish P1", "shadow: h0 product", "wait partials 1", "cell-1 update, publish", "wait m1 slice", "stage Q product, publish",
             "wait query granules", "energies, publish", "wait energies", "softmax, context, prenet-1, publish", "wait prenet-1 row", "prenet-2, publish", "-", "-", "-"]
    eng.persist_infer_stamps = torch.zeros(256, 24, dtype=torch.int64, device=dev)
    lin, stop, al, n = eng.decode(values, keys, t(lens), seed=1, max_steps=S)
    torch.cuda.synchronize()
    st = eng.persist_infer_stamps.cpu().numpy().astype(np.float64)
    eng.persist_infer_stamps = None
    us = st.mean(axis=0) / 100.0 / n
    for i, nm in enumerate(NAMES):
        if nm != "-":
            print("  stage %2d %-38s %6.2f us" % (i, nm, us[i]))
    att = us[13:19].sum()
    print("  frame %.2f us; attention stage (m1 leaves its producers -> context published: stages 13-18) %.2f us = %.3f of 8 TB/s for %d rows x 460 288 B"
          % (us.sum(), att, B * 460288 / (att * 1e-6) / 8e12, B))
# whole inference forward of BASELINE configs[3] (speaker encoder on 5 x 64-frame mels per utterance, text encoder, free-running decoder to
# max_steps, postnet, Taco1 mel -> spectrogram), results copied to the host as MSTTS_SV.Inference does
spk_mel = g.normal(0, 1, (5 * B, 64, d.n_mel)).astype(np.float32)
pattern = {"Token": tok, "Token_Length": lens, "Speaker_Embedding_Mel": spk_mel}
for it in range(4):          # (results land in page-locked blocks that are recycled once the caller drops them: steady from the third call on)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = eng.forward(pattern, seed=1, max_steps=S, with_vocoder=True)
    dt = time.perf_counter() - t0
    n = res["Linear"].shape[1]
    print("forward (speaker encoder + Tacotron2 + Taco1 vocoder, host copies included): %d frames x batch %d: %.1f ms  (%.0f mel-frames/s)"
          % (n, B, dt * 1e3, B * n / dt))
eng.profile_phases = True
res = eng.forward(pattern, seed=1, max_steps=S, with_vocoder=True)
print("  phases (a synchronisation behind each): " + ", ".join("%s %.2f ms" % kv for kv in eng.phase_ms.items()) + "; result bytes %.1f MB" % (sum(v.nbytes for v in res.values()) / 1e6))
