"""Drop-in for the reference driver class ``MSTTS_SV.Tacotron2`` (MSTTS_SV.py:20-505) on the MI355X.

Same constructor and methods - ``Tacotron2(is_Training)``, ``Restore()``, ``Train()``,
``Inference(path_List, text_List, file_Prefix)`` - and the same result-dict keys
(``train_Tensor_Dict`` / ``inference_Tensor_Dict`` names, MSTTS_SV.py:194-216).  The TensorFlow session is
replaced by ``engine.TrainEngine`` / ``inference.InferEngine`` (libmstts_hip.so calls on one HIP stream).
Out of scope here: matplotlib PNG export.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from . import Hyper_Parameters as hp
from . import Feeder as _Feeder
from .engine import TrainEngine, learning_rate
from .inference import InferEngine
from .params import Dims

SPEAKER_SIDE_STREAM = os.environ.get("MSTTS_SPEAKER_SIDE_STREAM", "1") != "0"      # Train_Step: the frozen speaker stack on its own stream beside the step's encoder

TRAIN_KEYS = ("Global_Step", "Learning_Rate", "Loss", "Linear_Loss", "Postnet_Loss", "Stop_Loss", "Weight_Regularization_Loss", "Train_OP")
INFERENCE_KEYS = ("Global_Step", "Linear", "Mel", "Stop", "Attention_History", "Spectrogram")


class StepResult(dict):
    """What Train_Step returns: the reference's train_Tensor_Dict results (MSTTS_SV.py:194-203,270-273).  Global_Step and Learning_Rate are
    known on the host; the four loss words are still on their way from the device when the step's launches have been enqueued, and reading
    them there would drain the device between two steps.  They are fetched on FIRST ACCESS of any loss key (or of the dict as a whole):
    Tacotron2.Train touches step k's losses after step k + 1 has been enqueued - same log line, one step later; a caller that reads
    them at once simply waits for the step, as with tf.Session.run."""
    LOSS_KEYS = ("Loss", "Linear_Loss", "Postnet_Loss", "Stop_Loss", "Weight_Regularization_Loss")

    def __init__(self, known, handle):
        super().__init__(known)
        self._handle = handle

    def _fill(self):
        if self._handle is not None:
            h, self._handle = self._handle, None
            res = h.get()
            super().update(res)
            if not np.isfinite(res["Loss"]):
                raise FloatingPointError("non-finite loss at global step %d: %r" % (super().__getitem__("Global_Step"), res))

    def __getitem__(self, k):
        if k in self.LOSS_KEYS:
            self._fill()
        return super().__getitem__(k)

    def get(self, k, default=None):
        if k in self.LOSS_KEYS:
            self._fill()
        return super().get(k, default)

    def __contains__(self, k):
        return k in self.LOSS_KEYS or super().__contains__(k)

    def keys(self):
        self._fill(); return super().keys()

    def values(self):
        self._fill(); return super().values()

    def items(self):
        self._fill(); return super().items()

    def __iter__(self):
        self._fill(); return super().__iter__()

    def __len__(self):
        self._fill(); return super().__len__()

    def __repr__(self):
        self._fill(); return super().__repr__()

    def copy(self):
        self._fill(); return dict(self)


class _BatchUploader:
    """Host -> device path of a training pattern (MSTTS_SV.py:270-273 feeds numpy arrays through feed_dict): page-locked staging blocks and
    device blocks, two of each in rotation and sized for the largest pattern seen (grown, never shrunk), the copies on their own stream.
    Tacotron2.Train_Step stages the NEXT pattern right after it has enqueued the current step, so the 8 MB of a batch cross PCIe under the
    step's tail instead of in front of the next step; nothing is allocated per step."""
    FIELDS = (("Token", np.int32), ("Token_Length", np.int32), ("Mel", np.float32), ("Mel_Length", np.int32),
              ("Speaker_Embedding", np.float32), ("Speaker_Embedding_Mel", np.float32))

    def __init__(self, device, presize=None):
        """presize: {field: elements} - blocks created at that size up front (page-locking memory is a system call: not in the middle of training)."""
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.slots = [{"host": {}, "dev": {}, "uploaded": None, "released": None} for _ in range(2)]
        self.next = 0
        for name, np_dtype in self.FIELDS:
            if presize and presize.get(name):
                for slot in self.slots:
                    self._block(slot, name, int(presize[name]), np_dtype)

    def _block(self, slot, name, n, np_dtype):
        h = slot["host"].get(name)
        if h is None or h.numel() < n:
            cap = max(n, int(1.25 * h.numel()) if h is not None else n)
            tdt = torch.from_numpy(np.zeros(1, np_dtype)).dtype
            slot["host"][name] = torch.zeros(cap, dtype=tdt).pin_memory()
            # (empty, not zeros: a fill kernel would run on the CALLER's stream, unordered with the copy stream - it could land behind the first
            #  upload and wipe the batch.  That race was real: parameters went non-finite in the first steps of 2 of 6 in-bench runs, r06 notes.)
            slot["dev"][name] = torch.empty(cap, dtype=tdt, device=self.device)
            self.stream.wait_stream(torch.cuda.current_stream(self.device))      # ... and whatever the allocator's previous owner of these bytes still has in flight
        return slot["host"][name], slot["dev"][name]

    def stage(self, pattern):
        """Copy the pattern's arrays into a slot's page-locked blocks and enqueue their upload on the copy stream.  Returns the batch
        (device views + the event `uploaded`); the consumer's stream must wait for that event, and calls release(batch) when it has
        enqueued its last read."""
        slot = self.slots[self.next]
        self.next ^= 1
        if slot["uploaded"] is not None:
            slot["uploaded"].synchronize()                   # (its previous upload has left the page-locked blocks - two steps ago)
        batch, todo = {}, []
        for name, np_dtype in self.FIELDS:
            if name not in pattern:
                continue
            a = np.ascontiguousarray(pattern[name], dtype=np_dtype)
            h, dv = self._block(slot, name, a.size, np_dtype)
            h[:a.size].view(a.shape).numpy()[...] = a
            todo.append((h, dv, a.size))
            batch[name] = dv[:a.size].view(a.shape)
        with torch.cuda.stream(self.stream):
            if slot["released"] is not None:
                self.stream.wait_event(slot["released"])     # the step that read this slot's device blocks last has finished with them
            for h, dv, n in todo:
                dv[:n].copy_(h[:n], non_blocking=True)
            slot["uploaded"] = torch.cuda.Event()
            slot["uploaded"].record()
        batch["_uploaded"], batch["_slot"] = slot["uploaded"], slot
        return batch

    def release(self, batch):
        ev = torch.cuda.Event()
        ev.record()
        batch["_slot"]["released"] = ev


class Tacotron2:
    def __init__(self, is_Training=False, device="cuda", seed=1234, dims: Dims = None, allow_random_init=False):
        """allow_random_init: the reference REQUIRES trained speaker-encoder and vocoder checkpoints and raises ValueError
        without them (MSTTS_SV.py:223-242); pass True to run on their random initialisation instead (tests, benchmarks).

        Data parallel: started one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, as
        torch.distributed.run exports them) the instance joins the RCCL group, takes rank 0's variables, feeds every rank a
        different shard of each epoch's batches, all-reduces the gradients inside Train_Step and lets only rank 0 write
        checkpoints (after averaging the BN moving statistics)."""
        from . import dist as _dist
        self.is_Training = is_Training
        self.allow_random_init = allow_random_init
        self.rank, local_rank, self.world = _dist.env_ranks()
        if self.world > 1:
            if str(device) == "cuda":
                device = "cuda:%d" % local_rank
            torch.cuda.set_device(torch.device(device))
            _dist.init_process_group(device=device)
        self.device = device
        self.feeder = _Feeder.Feeder(is_Training=is_Training, device=device, rank=self.rank, world=self.world)
        # the workspace arena is sized once for the largest batch the feeder can produce: hp.Train.Batch_Size rows, as many frames as
        # Use_Wav_Length_Range allows, and the persistent decoder kernels' 256 encoder positions (longer texts grow it on demand)
        hint = (int(hp.Train.Batch_Size), 256, int(hp.Train.Use_Wav_Length_Range[1] / hp.Sound.Frame_Shift) + 1) if is_Training else None
        self.train_engine = TrainEngine(dims, device=device, seed=seed, rank=self.rank, world=self.world, arena_hint=hint)
        self._uploader = None
        self._prefetched = None              # (is_Pre_Train, pattern, staged batch) taken ahead for the next Train_Step
        self.host_seconds = {"feeder_wait": 0.0, "stage": 0.0, "step_enqueue": 0.0, "steps": 0, "prefetched": 0}   # where Train_Step's host time goes (tools/train_surface_bench.py)
        self.params = self.train_engine.params
        self.infer_engine = InferEngine(self.train_engine.d, device=device, seed=seed, params=self.params)
        self.train_Tensor_Dict = {k: k for k in TRAIN_KEYS} if is_Training else None
        self.inference_Tensor_Dict = {k: k for k in INFERENCE_KEYS}
        self.Speaker_Embedding_Load()
        self.Vocoder_Load()
        self._reducer = None
        if self.world > 1:
            self.train_engine.broadcast_state(src=0)
            self._reducer = _dist.GradAllReduce(self.params.grad, self.world)

    # ---- checkpoints: torch files holding the two flat slabs; the reference's own TF V2 bundles are read / written by tf_checkpoint.py
    @property
    def global_step(self):
        return self.train_engine.global_step

    def _ckpt_dir(self):
        return hp.Checkpoint_Path.replace("\\", "/")

    def Speaker_Embedding_Load(self):
        self._load_scope(hp.Speaker_Embedding.Checkpoint_Path, "speaker_embedding")

    def Vocoder_Load(self):
        """MSTTS_SV.py:229-242: the vocoder named by hp.Use_Vocoder is restored from its own checkpoint directory."""
        self.waveglow = None
        if hp.Use_Vocoder.upper() == "Taco1_Mel_to_Spect".upper():
            self._load_scope(hp.Taco1_Mel_to_Spect.Checkpoint_Path, "mel_to_spectrogram")
        elif hp.Use_Vocoder.upper() == "WaveGlow".upper():
            from .waveglow import WaveGlowEngine, WGDims
            f = os.path.join(hp.WaveGlow.Checkpoint_Path.replace("\\", "/"), "waveglow.pt")
            values = None
            if os.path.exists(f):
                values = {k: np.asarray(v) for k, v in torch.load(f, map_location="cpu").items()}
                print("waveglow checkpoint '%s' is loaded." % f)
            elif not self.allow_random_init:
                raise ValueError("There is no WaveGlow checkpoint.")          # MSTTS_SV.py:239-240
            else:
                print("No waveglow checkpoint at '%s': keeping the random initialisation (allow_random_init)." % f)
            self.waveglow = WaveGlowEngine(WGDims.from_hp(hp), device=self.device, values=values)
        else:
            raise ValueError("hp.Use_Vocoder must be 'Taco1_Mel_to_Spect' or 'WaveGlow'")

    def _load_scope(self, path, scope):
        f = os.path.join(path.replace("\\", "/"), "%s.pt" % scope)
        if not os.path.exists(f):
            from . import tf_checkpoint as tfc
            prefix = tfc.latest_checkpoint(path.replace("\\", "/"))
            if prefix is not None:                       # a checkpoint of the reference's own sub-model trainer
                vars_ = tfc.read_checkpoint(prefix)
                self.params.load({k: v for k, v in vars_.items() if k.startswith(scope)})
                print("%s TF checkpoint '%s' is loaded." % (scope, prefix))
                return
            if not self.allow_random_init:       # MSTTS_SV.py:226-227,239-240: a Tacotron2 without its trained sub-models is an error
                raise ValueError("There is no {} checkpoint.".format("speaker embedding" if scope == "speaker_embedding" else "Mel to Spect"))
            print("No %s checkpoint at '%s': keeping the random initialisation (allow_random_init)." % (scope, f))
            return
        values = torch.load(f, map_location="cpu")
        self.params.load({k: v for k, v in values.items() if k.startswith(scope)})
        print("%s checkpoint '%s' is loaded." % (scope, f))

    def Restore(self):
        d = self._ckpt_dir()
        steps = sorted(int(n.split("-")[1].split(".")[0]) for n in os.listdir(d) if n.startswith("CHECKPOINT-") and n.endswith(".pt")) if os.path.isdir(d) else []
        if not steps:
            from . import tf_checkpoint as tfc
            prefix = tfc.latest_checkpoint(d) if os.path.isdir(d) else None
            if prefix is None:
                print("There is no checkpoint.")
                return
            self.Import_TF_Checkpoint(prefix)
            return
        f = os.path.join(d, "CHECKPOINT-%d.pt" % steps[-1])
        state = torch.load(f, map_location="cpu")
        self.params.load(state["variables"])
        self.params.adam_m.copy_(state["adam_m"]); self.params.adam_v.copy_(state["adam_v"])
        self.train_engine.global_step = int(state["global_step"])
        self.train_engine.refresh_derived()
        print("Checkpoint '%s' is loaded." % f)

    def Import_TF_Checkpoint(self, prefix):
        """Restore from a checkpoint written by the reference's tf.train.Saver (MSTTS_SV.py:30-40,244-251): variables by
        their TF names, Adam slots `<name>/Adam`, `<name>/Adam_1` (also under the optimizer's `loss/` scope), `global_step`."""
        from . import tf_checkpoint as tfc
        vars_ = tfc.read_checkpoint(prefix)
        names = [n for n, _, _ in self.params.table if self.params.trainable[n]]
        found = {n: vars_[n] for n in names if n in vars_}
        found.update({n: vars_[n] for n, _, _ in self.params.table if n in vars_ and not n.startswith(("speaker_embedding", "mel_to_spectrogram"))})
        self.params.load(found)
        n_slots = 0
        for n in names:
            for store, suffix in ((self.params.adam_m, "/Adam"), (self.params.adam_v, "/Adam_1")):
                for key in (n + suffix, "loss/" + n + suffix):
                    if key in vars_:
                        o, sz = self.params.offset[n], int(np.prod(self.params.shape[n]))
                        store[o:o + sz].copy_(torch.from_numpy(np.asarray(vars_[key], np.float32).reshape(-1)))
                        n_slots += 1
                        break
        if "global_step" in vars_:
            self.train_engine.global_step = int(np.asarray(vars_["global_step"]).reshape(-1)[0])
        self.train_engine.refresh_derived()
        missing = [n for n in names if n not in vars_]
        print("TF checkpoint '%s' is loaded: %d of %d trainable variables, %d optimizer slots, global step %d."
              % (prefix, len(names) - len(missing), len(names), n_slots, self.global_step))
        if missing:
            print("  not in the checkpoint (kept as initialised): %s%s" % (", ".join(missing[:5]), " ..." if len(missing) > 5 else ""))

    def Export_TF_Checkpoint(self, directory=None):
        """Write the tacotron variables, Adam slots (under the optimizer's `loss/` scope, with the beta power accumulators) and
        global_step as a TF V2 checkpoint `CHECKPOINT-<step>` with the variable names of the reference's Saver (MSTTS_SV.py:30-40,
        289).  PARITY UNPINNED: no real TF checkpoint listing exists here to compare the name set against."""
        from . import tf_checkpoint as tfc
        d = directory or self._ckpt_dir()
        out = {k: v for k, v in self.params.export().items() if not k.startswith(("speaker_embedding", "mel_to_spectrogram", "waveglow"))}
        m, v = self.params.adam_m.cpu().numpy(), self.params.adam_v.cpu().numpy()
        for n, _, _ in self.params.table:
            if self.params.trainable[n]:
                o, sz = self.params.offset[n], int(np.prod(self.params.shape[n]))
                # the reference builds its AdamOptimizer inside tf.variable_scope('loss') (MSTTS_SV.py:127,171-176), so the Saver's
                # slot variables are loss/<var>/Adam, loss/<var>/Adam_1 and the two power accumulators loss/beta{1,2}_power
                out["loss/" + n + "/Adam"] = m[o:o + sz].reshape(self.params.shape[n])
                out["loss/" + n + "/Adam_1"] = v[o:o + sz].reshape(self.params.shape[n])
        b1, b2, _ = self.train_engine.adam
        out["loss/beta1_power"] = np.array(b1 ** (self.global_step + 1), np.float32)     # TF keeps beta^t for the NEXT step t = step + 1
        out["loss/beta2_power"] = np.array(b2 ** (self.global_step + 1), np.float32)
        out["global_step"] = np.array(self.global_step, np.int64)
        prefix = os.path.join(d, "CHECKPOINT-%d" % self.global_step)
        tfc.write_checkpoint(prefix, out)
        return prefix

    def Save(self, keep=5):
        """tf.train.Saver(max_to_keep=5).save (MSTTS_SV.py:287-289).  Data parallel: the BN moving statistics are averaged over
        the ranks first (a collective: every rank must call Save), then rank 0 alone writes."""
        if self.world > 1:
            self.train_engine.sync_statistics()
            if self.rank != 0:
                return
        d = self._ckpt_dir()
        os.makedirs(d, exist_ok=True)
        tacotron = {k: torch.from_numpy(v) for k, v in self.params.export().items()
                    if not k.startswith(("speaker_embedding", "mel_to_spectrogram", "waveglow"))}
        torch.save({"variables": tacotron, "adam_m": self.params.adam_m.cpu(), "adam_v": self.params.adam_v.cpu(),
                    "global_step": self.global_step}, os.path.join(d, "CHECKPOINT-%d.pt" % self.global_step))
        old = sorted(int(n.split("-")[1].split(".")[0]) for n in os.listdir(d) if n.startswith("CHECKPOINT-") and n.endswith(".pt"))
        for s in old[:-keep]:
            os.remove(os.path.join(d, "CHECKPOINT-%d.pt" % s))

    # ---- training (MSTTS_SV.py:253-293)
    def _to_device_batch(self, pattern, staged=None):
        """Pattern (numpy, Feeder.py:132-172) -> device batch.  The arrays travel through page-locked staging blocks on the copy stream
        (_BatchUploader; `staged`: already on their way); the compute stream waits for the upload, not the host."""
        dev = torch.device(self.device)
        if self._uploader is None:
            B, inf = int(hp.Train.Batch_Size), hp.Speaker_Embedding.Inference
            frames = int(hp.Train.Use_Wav_Length_Range[1] / hp.Sound.Frame_Shift) + 2
            self._uploader = _BatchUploader(dev, presize={"Token": B * 256, "Token_Length": B, "Mel": B * frames * hp.Sound.Mel_Dim, "Mel_Length": B,
                                                          "Speaker_Embedding_Mel": B * inf.Sample_Nums * inf.Mel_Frame * hp.Sound.Mel_Dim} if self.is_Training else None)
        up = staged if staged is not None else self._uploader.stage(pattern)
        torch.cuda.current_stream(dev).wait_event(up["_uploaded"])
        batch = {k: up[k] for k in ("Token", "Token_Length", "Mel", "Mel_Length")}
        batch["_upload"] = up
        if "Speaker_Embedding" in up:
            batch["Speaker_Embedding"] = up["Speaker_Embedding"]
        else:   # frozen speaker encoder forward (MSTTS_SV.py:49-56) with TRAINING-mode zoneout: Is_Training is fed to this stack too
            from .masks import MaskSet, step_seed
            self.infer_engine._keep = []
            mel = up["Speaker_Embedding_Mel"]
            nb = int(mel.shape[0])
            if getattr(self, "_spk_masks_nb", None) != nb:
                self._spk_masks = MaskSet(self.train_engine.d, 1, 1, 1, True, dev, rank=self.rank, speaker_windows=nb)
                self._spk_masks_nb = nb
            # (no host sync here: the stack's persistent launches are checked at the train step's own sync point, engine.forward)
            inf = self.infer_engine
            if SPEAKER_SIDE_STREAM and dev.type == "cuda":
                # the stack (a dense layer and three 64-step recurrent layers on 5 windows per utterance) meets the step only at the memory's speaker
                # columns: it runs on its own stream beside the step's masks, encoder convolutions and BiLSTM; engine.forward waits for
                # `_speaker_event` in front of the speaker tile.  The embedding is handed to the compute stream's allocator bookkeeping (record_stream).
                if inf._spk_stream is None:
                    inf._spk_stream = torch.cuda.Stream(device=dev)
                main = torch.cuda.current_stream(dev)
                with torch.cuda.stream(inf._spk_stream):
                    inf._spk_stream.wait_event(up["_uploaded"])
                    self._spk_masks.draw(step_seed(self.train_engine.seed, self.global_step))
                    emb, ticket = inf.speaker_embedding(mel, masks=self._spk_masks, defer=True)
                    emb = emb.clone()
                    emb.record_stream(main)
                    done = torch.cuda.Event()
                    done.record()
                batch["_speaker_event"] = done
            else:
                self._spk_masks.draw(step_seed(self.train_engine.seed, self.global_step))
                emb, ticket = inf.speaker_embedding(mel, masks=self._spk_masks, defer=True)
                emb = emb.clone()
            batch["Speaker_Embedding"] = emb
            if ticket is not None:
                batch["_speaker_ticket"] = ticket
        return batch

    def _prefetch(self, is_Pre_Train):
        """Take the NEXT pattern from the feeder if it has one ready and start its upload.  Runs on a helper thread WHILE train_step runs:
        the main thread spends most of a step blocked in its two status syncs (the interpreter lock is free then), so the 0.6 ms of copying the
        arrays into the page-locked blocks cost the step nothing, and the copy itself runs on the copy stream under the step."""
        if self._prefetched is not None or self.feeder.pattern_Queue is None:
            return
        pattern = self.feeder.Get_Train_Pattern(is_Pre_Train=is_Pre_Train, block=False)
        if pattern is not None:
            t0 = time.perf_counter()
            self._prefetched = (is_Pre_Train, pattern, self._uploader.stage(pattern))
            self.host_seconds["stage"] += time.perf_counter() - t0
            self.host_seconds["prefetched"] += 1

    def Train_Step(self, pattern=None, is_Pre_Train=False):
        """One iteration of the reference's `while True` body (MSTTS_SV.py:268-273); returns the train_Tensor_Dict results (StepResult: the
        losses are fetched when first read).  In a data-parallel job the gradients are all-reduced inside the step and the returned
        losses are the mean over the ranks."""
        staged = None
        feeder_driven = pattern is None
        if pattern is None:
            if self._prefetched is not None:
                pre, p, up = self._prefetched
                self._prefetched = None
                if pre == is_Pre_Train:
                    pattern, staged = p, up
                else:                        # the step asks for the other queue (pre-training -> main training): hand the pattern back
                    self.feeder.Unget_Train_Pattern(p, is_Pre_Train=pre)
            if pattern is None:
                t0 = time.perf_counter()
                pattern = self.feeder.Get_Train_Pattern(is_Pre_Train=is_Pre_Train)
                self.host_seconds["feeder_wait"] += time.perf_counter() - t0
        step = self.global_step
        t0 = time.perf_counter()
        batch = self._to_device_batch(pattern, staged)
        if staged is None:
            self.host_seconds["stage"] += time.perf_counter() - t0
        up = batch.pop("_upload")
        import threading
        helper = None
        if feeder_driven and self._prefetched is None and self.feeder.pattern_Queue is not None:
            dev = torch.device(self.device)

            def ahead():
                torch.cuda.set_device(dev)               # (the current device is per thread)
                self._prefetch(is_Pre_Train)
            helper = threading.Thread(target=ahead, daemon=True)
            helper.start()
        t0 = time.perf_counter()
        w = self.train_engine.train_step(batch, all_reduce=self._reducer)
        self.host_seconds["step_enqueue"] += time.perf_counter() - t0
        self.host_seconds["steps"] += 1
        self._uploader.release(up)
        handle = self.train_engine.scalars_async(w, average=self.world > 1)
        res = StepResult({"Global_Step": step, "Learning_Rate": learning_rate(step), "Train_OP": None}, handle)
        if helper is not None:
            helper.join()
        return res

    def Inference_WaveGlow(self, path_List, text_List, file_Prefix=None, speaker_Mel_List=None, masks=None, export=True, noise_seed=None):
        """MSTTS_SV.py:325-389 + Export_Inference_WaveGlow :449-466: Tacotron2 forward, the mels cut into
        hp.WaveGlow.Inference.Mel_Split_Length-frame chunks, vocoded hp.WaveGlow.Inference.Batch_Size chunks at a time,
        stitched per utterance, cut at the stop token (in samples of Export_Sample_Rate)."""
        from . import waveglow as WG
        pattern = self.feeder.Get_Inference_Pattern(path_List, text_List, speaker_Mel_List=speaker_Mel_List)
        res = self.infer_engine.forward(pattern, masks=masks, with_vocoder=False)
        res["Global_Step"] = self.global_step
        prefix = file_Prefix or "GS_{}".format(self.global_step)
        wavs = WG.vocode(self.waveglow, list(res["Mel"]), hp.WaveGlow.Inference.Mel_Split_Length, hp.WaveGlow.Inference.Batch_Size,
                         noise_seed=noise_seed)
        res["Wav"] = wavs
        cut = []
        for i, text in enumerate(text_List):
            s = _Feeder.stop_cut(res["Stop"][i])
            n = WG.export_length(res["Stop"][i], hp.Sound.Frame_Shift, hp.WaveGlow.Export_Sample_Rate)
            cut.append({"Linear": res["Linear"][i, :s], "Mel": res["Mel"][i, :s], "Stop": res["Stop"][i, :s],
                        "Attention_History": res["Attention_History"][i, :len(text) + 2, :s], "Wav": wavs[i][:n]})
        res["Cut"] = cut
        if export:
            out_dir = os.path.join(hp.Inference_Path, "WAV").replace("\\", "/")
            try:
                from scipy.io import wavfile
                os.makedirs(out_dir, exist_ok=True)
                for i, c in enumerate(cut):
                    wavfile.write(os.path.join(out_dir, "{}.IDX_{}.WAV".format(prefix, i)), hp.WaveGlow.Export_Sample_Rate, c["Wav"].astype(np.float32))
            except OSError as e:
                print("Inference export skipped: {}".format(e))
        return res

    def Run_Inference(self, sentence_file="Inference_Sentence_in_Train.txt"):
        """MSTTS_SV.py:254-265: synthesise the `wav path <TAB> sentence` lines of Inference_Sentence_in_Train.txt.  The reference
        fails when the file or a wav is missing; a missing file is reported and skipped here (rank 0 only in a data-parallel job)."""
        if self.rank != 0:
            return None
        if not os.path.exists(sentence_file):
            print("'{}' not found: in-training inference skipped.".format(sentence_file))
            return None
        paths, sentences = [], []
        with open(sentence_file, "r") as f:
            for line in f.readlines():
                if not line.strip():
                    continue
                embedding_Path, sentence = line.strip().split("\t")
                paths.append(embedding_Path)
                sentences.append(sentence)
        return self.Inference(paths, sentences)

    def Train(self, max_steps=None, pattern_fn=None, run_inference=True):
        """MSTTS_SV.py:253-293: an inference pass first, then loop forever (or `max_steps`): pre-train patterns while
        hp.Train.Use_Pre_in_Main_Train and the step is below hp.Train.Pre_Step, the reference's log line, a checkpoint every
        hp.Train.Checkpoint_Save_Timing steps and an inference pass every hp.Train.Inference_Timing steps."""
        if run_inference:
            self.Run_Inference()
        n = 0
        current = self.global_step
        pending = None                       # (StepResult, wall time, mode) of the step whose log line is still owed

        def log(entry):
            r, seconds, mode = entry
            if self.rank == 0:               # (reading r's losses waits for THAT step's four words only - the next step is already running)
                print("\t\t".join(["Time: {:0.3f}".format(seconds), "Global step: {}".format(r["Global_Step"]),
                                   "Mode: {}".format(mode),
                                   "Learning rate: {:0.5f}".format(r["Learning_Rate"]), "Linear loss: {:0.5f}".format(r["Linear_Loss"]),
                                   "Postnet loss: {:0.5f}".format(r["Postnet_Loss"]), "Stop loss: {:0.5f}".format(r["Stop_Loss"]),
                                   "WR loss: {:0.5f}".format(r["Weight_Regularization_Loss"])]))
            else:
                r["Loss"]                    # every rank checks its step for a non-finite loss
        while max_steps is None or n < max_steps:
            t0 = time.time()
            pre = bool(hp.Train.Use_Pre_in_Main_Train and current < hp.Train.Pre_Step)
            r = self.Train_Step(pattern_fn() if pattern_fn else None, is_Pre_Train=pre)
            if pending is not None:
                log(pending)
            pending = (r, time.time() - t0, "Pre-train" if current < hp.Train.Pre_Step else "Main")
            if (r["Global_Step"] + 1) % hp.Train.Checkpoint_Save_Timing == 0:
                self.Save()
            if run_inference and (r["Global_Step"] + 1) % hp.Train.Inference_Timing == 0:
                self.Run_Inference()
            current = r["Global_Step"]
            n += 1
        if pending is not None:
            log(pending)

    # ---- inference (MSTTS_SV.py:295-323,391-400)
    def Inference(self, path_List, text_List, file_Prefix=None, speaker_Mel_List=None, masks=None, export=True):
        if len(text_List) != (len(path_List) if speaker_Mel_List is None else len(speaker_Mel_List)):
            raise ValueError("path_List and text_List must have the same length")
        if hp.Use_Vocoder.upper() == "WaveGlow".upper():          # MSTTS_SV.py:295-299
            return self.Inference_WaveGlow(path_List, text_List, file_Prefix, speaker_Mel_List=speaker_Mel_List, masks=masks, export=export)
        pattern = self.feeder.Get_Inference_Pattern(path_List, text_List, speaker_Mel_List=speaker_Mel_List)
        res = self.infer_engine.forward(pattern, masks=masks)
        res["Global_Step"] = self.global_step
        prefix = file_Prefix or "GS_{}".format(self.global_step)
        cut = []
        for i, text in enumerate(text_List):       # Export_Inference_Mel_to_Spectrogram's cut rule
            s = _Feeder.stop_cut(res["Stop"][i])
            cut.append({"Linear": res["Linear"][i, :s], "Mel": res["Mel"][i, :s], "Stop": res["Stop"][i, :s],
                        "Attention_History": res["Attention_History"][i, :len(text) + 2, :s], "Spectrogram": res["Spectrogram"][i, :s]})
        res["Cut"] = cut
        if export:
            out_dir = os.path.join(hp.Inference_Path, "NPZ").replace("\\", "/")
            wav_dir = os.path.join(hp.Inference_Path, "WAV").replace("\\", "/")
            try:
                os.makedirs(out_dir, exist_ok=True)
                os.makedirs(wav_dir, exist_ok=True)
                for i, c in enumerate(cut):
                    np.savez_compressed(os.path.join(out_dir, "{}.IDX_{}.npz".format(prefix, i)), **c)
                    self._export_wav(c["Spectrogram"], os.path.join(wav_dir, "{}.IDX_{}.WAV".format(prefix, i)))
            except OSError as e:      # the reference's default path is a Windows drive letter
                print("Inference export skipped: {}".format(e))
        return res

    @staticmethod
    def _export_wav(spectrogram, path):
        """Export_Inference_Mel_to_Spectrogram's WAV leg (MSTTS_SV.py:403-414): Griffin-Lim on the cut spectrogram."""
        from . import Audio
        name = os.path.basename(path)
        if spectrogram.shape[0] <= 1:
            print("WAV '{}' exporting failed. The exported spectrogram is too short.".format(name))
            return
        try:
            if spectrogram.shape[1] != hp.Sound.Spectrogram_Dim:
                raise ValueError("spectrogram width {} != hp.Sound.Spectrogram_Dim".format(spectrogram.shape[1]))
            from scipy.io import wavfile
            wavfile.write(path, hp.Sound.Sample_Rate, Audio.Griffin_Lim(spectrogram).astype(np.float32))
        except Exception as e:       # the reference swallows and reports every export error
            print("Wav exporting failed: {}".format(e))
