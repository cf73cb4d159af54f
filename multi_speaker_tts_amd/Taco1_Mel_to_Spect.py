"""Drop-in for the reference's vocoder trainer class ``Taco1_Mel_to_Spect.Taco1_Mel_to_Spect.Mel_to_Spect``
(Taco1_Mel_to_Spect/Taco1_Mel_to_Spect.py:15-205): ``Mel_to_Spect().Restore() / .Train() / .Train_Step(pattern)``.

The saved file (`mel_to_spectrogram.pt` under hp.Taco1_Mel_to_Spect.Checkpoint_Path) is exactly what
``MSTTS_SV.Tacotron2.Vocoder_Load`` reads, as in the reference where the TTS model restores the vocoder scope from the
vocoder trainer's checkpoint directory (MSTTS_SV.py:229-234).  Patterns are dicts {'Mel': [B,T,80], 'Spectrogram': [B,T,1025]}
(the reference's Taco1 feeder pads pickled (mel, spectrogram) pairs to the batch maximum); without one a synthetic pattern
of the trainer's batch shape is used.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from . import Hyper_Parameters as hp
from .params import Dims
from .taco1_trainer import Taco1TrainEngine, learning_rate

TRAIN_KEYS = ("Global_Step", "Learning_Rate", "Loss", "Train_OP")


class Mel_to_Spect:
    def __init__(self, device="cuda", seed=1234, dims: Dims = None):
        self.device = device
        self.engine = Taco1TrainEngine(dims, device=device, seed=seed)
        self.params = self.engine.params
        self.train_Tensor_Dict = {k: k for k in TRAIN_KEYS}
        self.inference_Tensor_Dict = {k: k for k in ("Global_Step", "Mel", "Spectrogram")}

    def _file(self):
        return os.path.join(hp.Taco1_Mel_to_Spect.Checkpoint_Path.replace("\\", "/"), "mel_to_spectrogram.pt")

    def Restore(self):
        f = self._file()
        if not os.path.exists(f):
            print("There is no checkpoint.")
            return
        state = torch.load(f, map_location="cpu")
        self.params.load({k: v for k, v in state.items() if k.startswith("mel_to_spectrogram")})
        if "__adam_m__" in state:
            self.params.adam_m.copy_(state["__adam_m__"]); self.params.adam_v.copy_(state["__adam_v__"])
        self.engine.global_step = int(state.get("__global_step__", 0))
        print("Checkpoint '%s' is loaded." % f)

    def Save(self):
        f = self._file()
        os.makedirs(os.path.dirname(f), exist_ok=True)
        state = {k: torch.from_numpy(v) for k, v in self.params.export().items() if k.startswith("mel_to_spectrogram")}
        state.update({"__adam_m__": self.params.adam_m.cpu(), "__adam_v__": self.params.adam_v.cpu(), "__global_step__": self.engine.global_step})
        torch.save(state, f)

    def Synthetic_Pattern(self, batch_Size=None, length=200, seed=1234):
        g = np.random.default_rng(seed)
        B = batch_Size or hp.Taco1_Mel_to_Spect.Train.Batch_Size
        d = self.engine.d
        return {"Mel": np.clip(g.normal(0, 1.5, (B, length, d.n_mel)), -4, 4).astype(np.float32),
                "Spectrogram": g.uniform(0, 1, (B, length, d.n_spec)).astype(np.float32)}

    def Train_Step(self, pattern=None):
        """One iteration of the reference's `while True` body (:118-124)."""
        pattern = pattern or self.Synthetic_Pattern()
        dev = torch.device(self.device)
        t = lambda a: torch.as_tensor(np.asarray(a, np.float32)).to(dev).contiguous()
        step = self.engine.global_step
        w = self.engine.train_step(t(pattern["Mel"]), t(pattern["Spectrogram"]))
        res = self.engine.scalars(w)
        res.update({"Global_Step": step, "Learning_Rate": learning_rate(step), "Train_OP": None})
        return res

    def Train(self, max_steps=None, pattern_fn=None):
        while max_steps is None or self.engine.global_step < max_steps:
            t0 = time.time()
            r = self.Train_Step(pattern_fn() if pattern_fn else None)
            print("\t\t".join(["Time: {:0.3f}".format(time.time() - t0), "Global step: {}".format(r["Global_Step"]),
                               "Learning rate: {:0.5f}".format(r["Learning_Rate"]), "Loss: {:0.5f}".format(r["Loss"])]))
            if (r["Global_Step"] + 1) % hp.Taco1_Mel_to_Spect.Train.Checkpoint_Save_Timing == 0:
                self.Save()
