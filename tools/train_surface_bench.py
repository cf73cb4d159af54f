"""Does `Tacotron2.Train_Step` - the surface a user of the reference trains through - run at the speed of the engine underneath it, on the
data the reference trains on?  (/root/reference Feeder.py:89-184, MSTTS_SV.py:266-273.)

  1. writes synthetic pattern files in the reference's on-disk format (Pattern_Generate.py:66-76,245-274: pickled {'Token','Mel','Text','Dataset'}
     + METADATA.PICKLE) with the reference's length distribution: wav lengths uniform in hp.Train.Use_Wav_Length_Range (0.5 .. 9 s -> 40 .. 720
     mel frames at 12.5 ms), ~14 characters per second of speech;
  2. SURFACE leg: Tacotron2(is_Training=True).Train_Step() x N through the real Feeder - length-sorted files, Batch_Size groups, shuffled
     group order, every batch padded to its own maximum (a different (T_enc, T_dec) every step), patterns from the producer thread's queue,
     uploaded through page-locked staging, losses read one step late;
  3. ENGINE leg: the SAME N batches, already resident on the device, through the frozen speaker stack + TrainEngine.train_step.

Prints one JSON object: ms per step of both legs, their ratio, the shapes.  usage: train_surface_bench.py [--steps 50] [--warmup 3] [--keep DIR]"""
import argparse
import json
import os
import pickle
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch


def write_patterns(root, n_files, seed=0):
    """n_files pattern pickles + METADATA.PICKLE under `root` (the reference's format and length distribution)."""
    from multi_speaker_tts_amd import Hyper_Parameters as hp
    from multi_speaker_tts_amd import Pattern_Generate as PG
    g = np.random.default_rng(seed)
    lo, hi = hp.Train.Use_Wav_Length_Range
    for i in range(n_files):
        ms = g.uniform(lo, hi)
        frames = int(np.clip(round(ms / hp.Sound.Frame_Shift), lo / hp.Sound.Frame_Shift + 1, hi / hp.Sound.Frame_Shift - 1))
        chars = max(3, int(round(ms / 1000.0 * g.uniform(11.0, 17.0))))
        tok = g.integers(2, hp.Encoder.Embedding.Token_Size, size=chars).astype(np.int32)
        mel = np.clip(g.normal(0, 1.5, size=(frames, hp.Sound.Mel_Dim)), -4, 4).astype(np.float32)
        with open(os.path.join(root, "SYN.P_%05d.PICKLE" % i), "wb") as f:
            pickle.dump({"Token": tok, "Mel": mel, "Text": "x" * chars, "Dataset": "SYN"}, f, protocol=2)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        PG.Metadata_Generate(pattern_path=root)


def run(steps=50, warmup=3, keep=None, device="cuda:0", seed=0, quiet=True):
    from multi_speaker_tts_amd import Hyper_Parameters as hp
    from multi_speaker_tts_amd.MSTTS_SV import Tacotron2
    root = keep or tempfile.mkdtemp(prefix="mstts_surface_")
    os.makedirs(root, exist_ok=True)
    saved = {k: getattr(hp.Train, k) for k in ("Pattern_Path", "Main_Train_Dataset_List", "Use_Pre_in_Main_Train", "Max_Pattern_Queue")}
    try:
        B = int(hp.Train.Batch_Size)
        n_files = (steps + warmup + 2) * B
        if not os.path.exists(os.path.join(root, hp.Train.Metadata_File.upper())):
            write_patterns(root, n_files, seed)
        hp.Train.Pattern_Path, hp.Train.Main_Train_Dataset_List, hp.Train.Use_Pre_in_Main_Train = root, ["SYN"], False
        import contextlib
        import io
        sink = io.StringIO() if quiet else sys.stdout
        with contextlib.redirect_stdout(sink):
            t = Tacotron2(is_Training=True, device=device, allow_random_init=True)
        taken = []
        real_get = t.feeder.Get_Train_Pattern

        def spy(*a, **k):
            p = real_get(*a, **k)
            if p is not None:
                taken.append(p)
            return p
        t.feeder.Get_Train_Pattern = spy
        # ---- surface leg
        for _ in range(warmup):
            t.Train_Step()
        torch.cuda.synchronize()
        first = len(taken) - (1 if t._prefetched is not None else 0)          # patterns consumed by the warm-up steps
        deadline = time.time() + 20.0
        while len(t.feeder.pattern_Queue) < min(steps, hp.Train.Max_Pattern_Queue) - 1 and time.time() < deadline:
            time.sleep(0.05)                                                    # (the reference starts training on a filled queue too)
        for k in t.host_seconds:
            t.host_seconds[k] = 0
        t0 = time.perf_counter()
        results = [t.Train_Step() for _ in range(steps)]
        torch.cuda.synchronize()
        surface_ms = 1e3 * (time.perf_counter() - t0) / steps
        eng = t.train_engine
        losses, first_bad = [], None
        for i, r in enumerate(results):
            try:
                losses.append(float(r["Loss"]))
            except FloatingPointError:
                losses.append(float("nan"))
                if first_bad is None:
                    first_bad = i
        if first_bad is not None:
            shp = lambda p: (int(p["Token"].shape[0]), int(p["Token"].shape[1]), int(p["Mel"].shape[1]))
            seq = [shp(p) for p in taken[first:first + steps]]
            raise FloatingPointError("non-finite loss from timed step %d on (shape %r, the one before %r); fallbacks fwd %d bptt %d enc %d, speaker redos %d, "
                                     "disabled steps %d, parameters finite: %s" % (first_bad, seq[first_bad], seq[first_bad - 1] if first_bad else None,
                                     eng.persist_fallbacks, eng.persist_bwd_fallbacks, eng.persist_enc_fallbacks, eng.speaker_ticket_redos,
                                     eng.persist_disabled_steps, bool(torch.isfinite(eng.params.train).all())))
        host = {k: (1e3 * v / steps if k not in ("steps", "prefetched") else v) for k, v in t.host_seconds.items()}
        host["queue_length_at_end"] = len(t.feeder.pattern_Queue)
        host["feeder_workers"] = getattr(t.feeder, "_workers", 0)
        eng, inf = t.train_engine, t.infer_engine
        counters = dict(decoder_forward=eng.persist_fallbacks, decoder_bptt=eng.persist_bwd_fallbacks, encoder_bilstm=eng.persist_enc_fallbacks,
                        non_persistent_plans=eng.non_persistent_plans, speaker_ticket_redos=eng.speaker_ticket_redos, arena_generation=eng._arena.generation)
        t.feeder.close()
        pats = taken[first:first + steps]
        assert len(pats) == steps
        shapes = [(int(p["Token"].shape[0]), int(p["Token"].shape[1]), int(p["Mel"].shape[1])) for p in pats]
        # ---- engine leg: the same batches, device-resident, same speaker stack in front
        from multi_speaker_tts_amd.masks import MaskSet, step_seed
        dev = torch.device(device)
        res = [{k: torch.as_tensor(np.asarray(p[k])).to(dev).contiguous() for k in ("Token", "Token_Length", "Mel", "Mel_Length", "Speaker_Embedding_Mel")} for p in pats]
        masks = {}

        def engine_step(b):
            nb = int(b["Speaker_Embedding_Mel"].shape[0])
            if nb not in masks:
                masks[nb] = MaskSet(eng.d, 1, 1, 1, True, dev, speaker_windows=nb)
            masks[nb].draw(step_seed(eng.seed, eng.global_step))
            inf._keep = []
            emb, ticket = inf.speaker_embedding(b["Speaker_Embedding_Mel"], masks=masks[nb], defer=True)
            batch = {k: b[k] for k in ("Token", "Token_Length", "Mel", "Mel_Length")}
            batch["Speaker_Embedding"] = emb.clone()
            if ticket is not None:
                batch["_speaker_ticket"] = ticket
            eng.train_step(batch)
        for b in res[:warmup]:
            engine_step(b)
        eng._plans.clear()                   # every shape is new to this leg too, as it was to the surface leg
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in res:
            engine_step(b)
        torch.cuda.synchronize()
        engine_ms = 1e3 * (time.perf_counter() - t0) / steps
        frames = sum(s[0] * s[2] for s in shapes)
        return {"steps": steps, "warmup": warmup, "batch_size": B,
                "surface_ms_per_step": surface_ms, "engine_ms_per_step": engine_ms, "surface_over_engine": surface_ms / engine_ms,
                "surface_mel_frames_per_s": frames / (surface_ms * 1e-3 * steps), "engine_mel_frames_per_s": frames / (engine_ms * 1e-3 * steps),
                "distinct_shapes": len(set(shapes)), "tokens_min_max": [min(s[1] for s in shapes), max(s[1] for s in shapes)],
                "frames_min_max": [min(s[2] for s in shapes), max(s[2] for s in shapes)], "mean_padded_frames": float(np.mean([s[2] for s in shapes])),
                "loss_first_last": [losses[0], losses[-1]], "counters": counters, "surface_host_ms_per_step": host,
                "what": "Tacotron2.Train_Step through the real Feeder (synthetic pattern files in the reference's format, wav lengths uniform in "
                        "Use_Wav_Length_Range, length-sorted batches padded to their own maximum: a new shape every step) beside the same batches "
                        "device-resident through the speaker stack + TrainEngine.train_step"}
    finally:
        for k, v in saved.items():
            setattr(hp.Train, k, v)
        if keep is None:
            shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--keep", default=None)
    a = ap.parse_args()
    print(json.dumps(run(a.steps, a.warmup, a.keep)))
