"""Drop-in for the reference's speaker-encoder trainer class ``Speaker_Embedding.Speaker_Embedding.Speaker_Embedding``
(Speaker_Embedding/Speaker_Embedding.py:16-160): ``Speaker_Embedding(is_Training).Restore() / .Train() / .Train_Step(pattern) /
.Inference(mel_List)``.  `speaker_embedding.pt` under hp.Speaker_Embedding.Checkpoint_Path is exactly what
``MSTTS_SV.Tacotron2.Speaker_Embedding_Load`` reads (MSTTS_SV.py:223-227).  A training pattern is {'Mel': [Batch_Speaker *
Batch_per_Speaker, frames, 80]} speaker-major with one random frame count from hp.Speaker_Embedding.Train.Frame_Range per batch
(Speaker_Embedding/Feeder.py:66-100); without one a synthetic pattern of that shape is used.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from . import Hyper_Parameters as hp
from . import Feeder as _Feeder
from .params import Dims
from .speaker_trainer import SpeakerTrainEngine, learning_rate

TRAIN_KEYS = ("Global_Step", "Learning_Rate", "Loss", "Train_OP")


class Speaker_Embedding:
    def __init__(self, is_Training=True, device="cuda", seed=1234, dims: Dims = None):
        self.is_Training = is_Training
        self.device = device
        self.engine = SpeakerTrainEngine(dims, device=device, seed=seed)
        self.params = self.engine.params
        self.train_Tensor_Dict = {k: k for k in TRAIN_KEYS} if is_Training else None
        self.inference_Tensor_Dict = {k: k for k in ("Global_Step", "Embedding")}
        self._infer = None

    def _file(self):
        return os.path.join(hp.Speaker_Embedding.Checkpoint_Path.replace("\\", "/"), "speaker_embedding.pt")

    def Restore(self):
        f = self._file()
        if not os.path.exists(f):
            print("There is no checkpoint.")
            return
        state = torch.load(f, map_location="cpu")
        self.params.load({k: v for k, v in state.items() if k.startswith("speaker_embedding")})
        if "__adam_m__" in state:
            self.params.adam_m.copy_(state["__adam_m__"]); self.params.adam_v.copy_(state["__adam_v__"])
            self.engine.wb.copy_(state["__loss_vars__"]); self.engine.wb_m.copy_(state["__loss_m__"]); self.engine.wb_v.copy_(state["__loss_v__"])
        self.engine.global_step = int(state.get("__global_step__", 0))
        print("Checkpoint '%s' is loaded." % f)

    def Save(self):
        f = self._file()
        os.makedirs(os.path.dirname(f), exist_ok=True)
        state = {k: torch.from_numpy(v) for k, v in self.params.export().items() if k.startswith("speaker_embedding")}
        e = self.engine
        state.update({"__adam_m__": self.params.adam_m.cpu(), "__adam_v__": self.params.adam_v.cpu(), "__loss_vars__": e.wb.cpu(),
                      "__loss_m__": e.wb_m.cpu(), "__loss_v__": e.wb_v.cpu(), "__global_step__": e.global_step})
        torch.save(state, f)

    def Synthetic_Pattern(self, speakers=None, per_speaker=None, seed=1234):
        """Speaker-dependent synthetic mels (a per-speaker offset pattern plus noise), so the loss has something to learn."""
        tr = hp.Speaker_Embedding.Train
        S, P = speakers or tr.Batch_Speaker, per_speaker or tr.Batch_per_Speaker
        g = np.random.default_rng(seed)
        T = int(g.integers(tr.Frame_Range[0], tr.Frame_Range[1] + 1))
        d = self.engine.d
        base = g.normal(0, 1.0, (S, 1, 1, d.n_mel))
        mel = np.clip(base + g.normal(0, 1.0, (S, P, T, d.n_mel)), -4, 4).astype(np.float32)
        return {"Mel": mel.reshape(S * P, T, d.n_mel), "Batch_per_Speaker": P}

    def Train_Step(self, pattern=None):
        pattern = pattern or self.Synthetic_Pattern()
        dev = torch.device(self.device)
        mel = torch.as_tensor(np.asarray(pattern["Mel"], np.float32)).to(dev).contiguous()
        step = self.engine.global_step
        w = self.engine.train_step(mel, int(pattern.get("Batch_per_Speaker", hp.Speaker_Embedding.Train.Batch_per_Speaker)))
        return {"Global_Step": step, "Learning_Rate": learning_rate(step), "Loss": float(w.out3[0]), "Train_OP": None}

    def Train(self, max_steps=None, pattern_fn=None):
        while max_steps is None or self.engine.global_step < max_steps:
            t0 = time.time()
            r = self.Train_Step(pattern_fn() if pattern_fn else None)
            print("\t\t".join(["Time: {:0.3f}".format(time.time() - t0), "Global step: {}".format(r["Global_Step"]),
                               "Learning rate: {:0.6f}".format(r["Learning_Rate"]), "Loss: {:0.5f}".format(r["Loss"])]))
            if (r["Global_Step"] + 1) % hp.Speaker_Embedding.Train.Checkpoint_Save_Timing == 0:
                self.Save()

    def Inference(self, mel_List):
        """Embeddings of whole utterances: 5 windows of 64 frames each, mean of the last-frame outputs, whole-tensor l2
        normalisation (Speaker_Embedding/Modules.py:127-137; Feeder.py:105-160).  mel_List: [T_i, 80] arrays."""
        from .inference import InferEngine
        if self._infer is None:
            self._infer = InferEngine(self.engine.d, device=self.device, params=self.params)
        self._infer._keep = []
        win = torch.as_tensor(_Feeder.speaker_windows(mel_List)).to(torch.device(self.device), torch.float32).contiguous()
        return self._infer.speaker_embedding(win).cpu().numpy()
