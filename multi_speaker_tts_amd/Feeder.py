"""Host-side data contract of the reference's ``Feeder`` (Feeder.py:33-41,62-87,186-233) without
TensorFlow: the same tensor names / dtypes / padding, tokenisation and speaker-window extraction,
plus synthetic train patterns of the benchmark shape.  Mel extraction of speaker wavs runs on the
GPU through ``Audio.melspectrogram``; wav file reading/resampling/trimming is scipy plumbing (the
reference uses librosa.core.load + librosa.effects.trim, which are not available here).
"""
from __future__ import annotations

import json
import os

import numpy as np

from . import Hyper_Parameters as hp

_HERE = os.path.dirname(os.path.abspath(__file__))
PLACEHOLDERS = ("Is_Training", "Token", "Token_Length", "Mel", "Mel_Length", "Speaker_Embedding_Mel")


def load_token_dict(path=None):
    with open(path or os.path.join(_HERE, "Token_Index_Dict.json"), "r") as f:
        return json.load(f)


_TOKENS = None


def _tokens():
    global _TOKENS
    if _TOKENS is None:
        _TOKENS = load_token_dict()
    return _TOKENS


def tokenize(text_List, token_dict=None):
    """text -> int32 [B, T_max] (<S> chars <E>, upper-cased, right-padded with <E>) and lengths.
    Unknown characters raise KeyError exactly like the reference's dict lookup (Feeder.py:192)."""
    table = token_dict or _tokens()
    start, end = table["<S>"], table["<E>"]
    rows = []
    for text in text_List:
        ids = [start]
        for letter in text.upper():
            ids.append(table[letter])
        ids.append(end)
        rows.append(ids)
    width = max(len(r) for r in rows)
    token = np.full((len(rows), width), end, dtype=np.int32)
    for i, r in enumerate(rows):
        token[i, :len(r)] = r
    return token, np.asarray([len(r) for r in rows], dtype=np.int32)


def _required_frames():
    inf = hp.Speaker_Embedding.Inference
    return inf.Sample_Nums * (inf.Mel_Frame - inf.Overlap_Frame) + inf.Overlap_Frame


def window_starts(n_frames):
    """Start frames of the Sample_Nums windows taken from the middle of a mel (None: too short)."""
    inf = hp.Speaker_Embedding.Inference
    need = _required_frames()
    if n_frames < need:
        return None
    first = int((n_frames - need) / 2)
    return [first + k * inf.Overlap_Frame for k in range(inf.Sample_Nums)]


def speaker_windows(mel_List):
    """[T_i, mel] list -> float32 [len * Sample_Nums, Mel_Frame, mel] (Feeder.Speaker_Embedding_Mel)."""
    inf = hp.Speaker_Embedding.Inference
    out = np.zeros((len(mel_List), inf.Sample_Nums, inf.Mel_Frame, hp.Sound.Mel_Dim), dtype=np.float32)
    for i, mel in enumerate(mel_List):
        starts = window_starts(mel.shape[0])
        if starts is None:
            head = mel[:inf.Mel_Frame]
            out[i, :, :head.shape[0]] = head          # every window gets the same (zero-padded) head
        else:
            for k, s0 in enumerate(starts):
                out[i, k] = mel[s0:s0 + inf.Mel_Frame]
    return out.reshape(-1, inf.Mel_Frame, hp.Sound.Mel_Dim)


def stop_cut(stop):
    """Export cut (MSTTS_SV.py:395): index of the first frame whose sigmoid(stop) > 0.5, else the length."""
    stop = np.asarray(stop)
    hit = np.nonzero(stop > 0.5)[0]
    return int(hit[0]) if hit.size else int(stop.shape[0])


def read_sphere(path, start_time=None, end_time=None):
    """NIST SPHERE file (TIMIT's .WAV, TEDLIUM's .sph; the reference reads these through `sphfile`, Pattern_Generate.py:80-83):
    1024-byte-granular ASCII header `NIST_1A / <header bytes> / key -type value ... / end_head`, then uncompressed PCM.
    Returns (sample_rate, int16/int32 array [n] or [n, channels]), optionally cut to [start_time, end_time) seconds."""
    with open(path, "rb") as f:
        magic = f.readline().strip()
        if magic != b"NIST_1A":
            raise ValueError("'{}' is not a NIST SPHERE file".format(path))
        header_bytes = int(f.readline().strip())
        f.seek(0)
        header = f.read(header_bytes).decode("ascii", "replace")
        fields = {}
        for line in header.splitlines()[2:]:
            parts = line.strip().split(None, 2)
            if not parts or parts[0] == "end_head":
                break
            if len(parts) == 3:
                fields[parts[0]] = int(parts[2]) if parts[1] == "-i" else float(parts[2]) if parts[1] == "-r" else parts[2]
        coding = str(fields.get("sample_coding", "pcm"))
        if "shorten" in coding or "ulaw" in coding or "alaw" in coding:
            raise ValueError("SPHERE sample_coding '{}' is not supported (uncompressed pcm only)".format(coding))
        width = int(fields.get("sample_n_bytes", 2))
        if width not in (2, 4):          # (1-byte SPHERE pcm is offset-coded and 3-byte samples have no NumPy dtype; neither occurs in TIMIT / TEDLIUM)
            raise ValueError("SPHERE sample_n_bytes {} is not supported (16- or 32-bit pcm only)".format(width))
        order = "<" if str(fields.get("sample_byte_format", "01")).startswith("01") else ">"
        raw = np.frombuffer(f.read(), dtype=np.dtype("%si%d" % (order, width)))
    ch = int(fields.get("channel_count", 1))
    n = int(fields.get("sample_count", raw.shape[0] // ch))
    data = raw[:n * ch].astype(np.int16 if width == 2 else np.int32)
    if ch > 1:
        data = data.reshape(-1, ch)
    rate = int(fields["sample_rate"])
    if start_time is not None or end_time is not None:
        a = int(round((start_time or 0.0) * rate))
        b = data.shape[0] if end_time is None else int(round(end_time * rate))
        data = data[a:b]
    return rate, data


def read_audio(path):
    """(sample_rate, samples) of a .wav (RIFF or - as in TIMIT - SPHERE behind a .WAV name), .sph, or, when the optional `soundfile`
    module is present, .flac / .m4a file (the reference goes through librosa.core.load, Pattern_Generate.py:36)."""
    ext = os.path.splitext(path)[1].lower()
    with open(path, "rb") as f:
        head = f.read(8)
    if head.startswith(b"NIST_1A"):
        return read_sphere(path)
    if ext == ".wav" or head.startswith(b"RIFF"):
        from scipy.io import wavfile
        return wavfile.read(path)
    try:
        import soundfile
    except ImportError:
        raise ValueError("'{}': decoding {} needs the optional `soundfile` module; convert the corpus to .wav".format(path, ext))
    data, rate = soundfile.read(path, dtype="float32")
    return rate, data


def load_wav(path, sample_rate=None, top_db=15.0, frame=32, hop=16):
    """Speaker wav -> float mono at hp.Sound.Sample_Rate, silence-trimmed, scaled by 0.99
    (Feeder.py:213-215 plumbing: scipy.io.wavfile + polyphase resampling + a frame_length=32 /
    hop_length=16 RMS trim standing in for librosa.effects.trim)."""
    from scipy.signal import resample_poly
    sr = sample_rate or hp.Sound.Sample_Rate
    rate, data = read_audio(path)
    if data.dtype.kind == "i":
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max)
    elif data.dtype.kind == "u":
        data = (data.astype(np.float32) - 128.0) / 128.0
    data = data.astype(np.float32)
    if data.ndim > 1:
        data = data.mean(axis=1)
    if rate != sr:
        g = np.gcd(int(rate), int(sr))
        data = resample_poly(data, sr // g, rate // g).astype(np.float32)
    if data.shape[0] >= frame:
        n = 1 + (data.shape[0] - frame) // hop
        idx = np.arange(frame)[None, :] + hop * np.arange(n)[:, None]
        rms = np.sqrt((data[idx] ** 2).mean(axis=1))
        db = 20.0 * np.log10(np.maximum(rms, 1e-10) / max(rms.max(), 1e-10))
        keep = np.nonzero(db > -top_db)[0]
        if keep.size:
            data = data[keep[0] * hop: min(data.shape[0], (keep[-1] + 1) * hop)]
    return data * 0.99


def metadata_path():
    return os.path.join(hp.Train.Pattern_Path, hp.Train.Metadata_File.upper()).replace("\\", "/")


def check_metadata(md):
    """Feeder.py:46-56: the pattern set must have been generated with the current hyper parameters."""
    if not all([len(md["Token_Index_Dict"]) == hp.Encoder.Embedding.Token_Size, md["Spectrogram_Dim"] == hp.Sound.Spectrogram_Dim,
                md["Mel_Dim"] == hp.Sound.Mel_Dim, md["Frame_Shift"] == hp.Sound.Frame_Shift, md["Frame_Length"] == hp.Sound.Frame_Length,
                md["Sample_Rate"] == hp.Sound.Sample_Rate]):
        raise ValueError("The metadata information and hyper parameter setting are not consistent.")


def train_file_order(md, is_Pre_Train=False):
    """Feeder.py:89-112: files of the selected datasets whose mel length lies in Use_Wav_Length_Range (ms / Frame_Shift),
    sorted by mel length when hp.Train.Pattern_Sorting_by_Mel_Length (stable, like `sorted`)."""
    wanted = hp.Train.Pre_Train_Dataset_List if is_Pre_Train else hp.Train.Main_Train_Dataset_List
    files = [f for f in md["File_List"] if md["Dataset_Dict"][f] in wanted]
    lo, hi = hp.Train.Use_Wav_Length_Range[0] / hp.Sound.Frame_Shift, hp.Train.Use_Wav_Length_Range[1] / hp.Sound.Frame_Shift
    sel = [(f, md["Mel_Length_Dict"][f]) for f in files if lo <= md["Mel_Length_Dict"][f] <= hi]
    if hp.Train.Pattern_Sorting_by_Mel_Length:
        sel = sorted(sel, key=lambda x: x[1])
    return [f for f, _ in sel]


def epoch_batches(path_List, rng):
    """Feeder.py:114-123: one pass = (shuffle the files unless they are length-sorted) -> consecutive Batch_Size groups ->
    shuffle the groups.  `rng` needs .shuffle (random or numpy RandomState)."""
    path_List = list(path_List)
    if not hp.Train.Pattern_Sorting_by_Mel_Length:
        rng.shuffle(path_List)
    batches = [path_List[x:x + hp.Train.Batch_Size] for x in range(0, len(path_List), hp.Train.Batch_Size)]
    rng.shuffle(batches)
    return batches


def load_pattern_batch(file_names, token_dict, pattern_path=None):
    """Feeder.py:132-172: pickled {'Token','Mel','Text','Dataset'} dicts -> one padded training pattern."""
    import pickle
    root = pattern_path or hp.Train.Pattern_Path
    token_List, mel_List = [], []
    for name in file_names:
        with open(os.path.join(root, name).replace("\\", "/"), "rb") as f:
            pd = pickle.load(f)
        token_List.append(np.hstack([token_dict["<S>"], pd["Token"], token_dict["<E>"]]).astype(np.int32))
        mel_List.append(np.asarray(pd["Mel"], np.float32))
    B = len(file_names)
    token = np.full((B, max(t.shape[0] for t in token_List)), token_dict["<E>"], np.int32)
    mel = np.zeros((B, max(m.shape[0] for m in mel_List), hp.Sound.Mel_Dim), np.float32)
    for i, (t, m) in enumerate(zip(token_List, mel_List)):
        token[i, :t.shape[0]] = t
        mel[i, :m.shape[0]] = m
    return {"Is_Training": True, "Token": token, "Token_Length": np.array([t.shape[0] for t in token_List], np.int32), "Mel": mel,
            "Mel_Length": np.array([m.shape[0] for m in mel_List], np.int32), "Speaker_Embedding_Mel": speaker_windows(mel_List)}


_SHM_KEYS = ("Token", "Token_Length", "Mel", "Mel_Length", "Speaker_Embedding_Mel")
_worker_shm = {}                             # loader processes: shared-memory blocks attached so far, by name


def _load_pattern_batch_into(names, token_dict, root, shm_name, shm_size):
    """Loader process: load_pattern_batch, the arrays written into the shared-memory block `shm_name` (a 10 MB batch through the pool's result
    pipe costs the parent more than loading it itself would).  Returns ("shm", [(key, dtype, shape, offset)]) or - a batch that does not
    fit the block - ("value", pattern)."""
    import mmap
    p = load_pattern_batch(names, token_dict, pattern_path=root)
    need = sum((p[k].nbytes + 63) // 64 * 64 for k in _SHM_KEYS)
    if need > shm_size:
        return "value", p
    buf = _worker_shm.get(shm_name)
    if buf is None:
        # (mapped as a file, not through multiprocessing.shared_memory: attaching there registers the block with the resource tracker a second
        #  time, and the tracker then complains about the parent's unlink)
        fd = os.open("/dev/shm/" + shm_name.lstrip("/"), os.O_RDWR)
        try:
            buf = _worker_shm[shm_name] = mmap.mmap(fd, shm_size)
        finally:
            os.close(fd)
    layout, off = [], 0
    for k in _SHM_KEYS:
        a = np.ascontiguousarray(p[k])
        np.ndarray(a.shape, a.dtype, buffer=buf, offset=off)[...] = a
        layout.append((k, a.dtype.str, a.shape, off))
        off += (a.nbytes + 63) // 64 * 64
    return "shm", layout


class Feeder:
    """Same constructor / pattern API as the reference class; `placeholder_Dict` maps the reference's
    placeholder names to themselves (there is no graph), patterns are dicts keyed by those names.

    Training patterns come from the reference's on-disk format when hp.Train.Pattern_Path holds a METADATA.PICKLE
    (Pattern_Generate.py:245-274): a daemon thread keeps up to hp.Train.Max_Pattern_Queue padded batches ready
    (Feeder.py:89-184).  Without pattern files the feeder serves the synthetic benchmark pattern."""

    def __init__(self, is_Training=False, device="cuda", seed=None, rank=0, world=1):
        """rank / world: data-parallel sharding - every rank walks the same shuffled batch list of an epoch (so the seed must be
        shared: it defaults to 1234 when world > 1) and takes the batches rank, rank + world, ..."""
        self.is_Training = is_Training
        self.device = device
        self.rank, self.world = rank, world
        if world > 1 and seed is None:
            seed = 1234
        self._producer_error = {}
        self.placeholder_Dict = {name: name for name in PLACEHOLDERS}
        self.metadata_Dict = {"Token_Index_Dict": load_token_dict()}
        self.pattern_Queue = None
        self._seed = seed
        self._stop = False
        if is_Training and os.path.exists(metadata_path()):
            self.Metadata_Load()
            self._start_producers()

    def Metadata_Load(self):
        import pickle
        with open(metadata_path(), "rb") as f:
            self.metadata_Dict = pickle.load(f)
        check_metadata(self.metadata_Dict)

    def _start_producers(self):
        from collections import deque
        from threading import Thread
        # The reference's pattern files are protocol-2 pickles (Pattern_Generate.py:66-76); under Python 3 a protocol-2 NumPy array is a
        # latin-1 round trip of its bytes, ~1 ms per file, 30-40 ms per batch of 32 - more than a train step of an average batch takes on
        # this GPU (the reference's producer thread, Feeder.py:89-172, had seconds per step to hide in).  So the producer thread hands the
        # batches to a few WORKER PROCESSES (forked: they only ever run NumPy and pickle, never the GPU runtime) and keeps the reference's
        # order: batch k enters the queue before batch k + 1.  MSTTS_FEEDER_WORKERS=0 loads in the producer thread itself.
        n = os.environ.get("MSTTS_FEEDER_WORKERS")
        self._workers = int(n) if n is not None else (4 if (os.cpu_count() or 1) >= 8 else 0)
        self._pool, self._shm = None, []
        if self._workers > 0:
            import multiprocessing
            from concurrent.futures import ProcessPoolExecutor
            # shared-memory blocks (allocated per producer thread, Train_Pattern_Generate): one per batch in flight (+ 1), sized for the largest batch the length filter admits with texts of up to 256 tokens
            frames = int(hp.Train.Use_Wav_Length_Range[1] / hp.Sound.Frame_Shift) + 2
            inf = hp.Speaker_Embedding.Inference
            B = int(hp.Train.Batch_Size)
            self._shm_size = 4 * (B * frames * hp.Sound.Mel_Dim + B * inf.Sample_Nums * inf.Mel_Frame * hp.Sound.Mel_Dim + B * 258) + 4096
            self._pool = ProcessPoolExecutor(max_workers=self._workers, mp_context=multiprocessing.get_context("fork"))
            # fork the loaders NOW, from the constructing thread (Tacotron2 builds its feeder before its engines): the executor would otherwise
            # fork them on first use, from the producer thread, in the middle of the main thread's GPU work
            import time as _time
            for fut in [self._pool.submit(_time.sleep, 0.05) for _ in range(self._workers)]:
                fut.result()
        if hp.Train.Use_Pre_in_Main_Train:
            self.pre_Pattern_Queue = deque()
            Thread(target=self.Train_Pattern_Generate, args=[True], daemon=True).start()
        self.pattern_Queue = deque()
        Thread(target=self.Train_Pattern_Generate, args=[False], daemon=True).start()

    def Train_Pattern_Generate(self, is_Pre_Train=False):
        """Producer loop of Feeder.py:89-172 (runs forever in a daemon thread)."""
        import random
        import time
        from collections import deque
        rng = random.Random(self._seed) if self._seed is not None else random
        queue = self.pre_Pattern_Queue if is_Pre_Train else self.pattern_Queue
        order = train_file_order(self.metadata_Dict, is_Pre_Train)
        print("Pre train pattern info" if is_Pre_Train else "Main train pattern info", "\n",
              "Total pattern count: {}".format(len(self.metadata_Dict["Mel_Length_Dict"])), "\n",
              "Use pattern count: {}".format(len(order)), "\n",
              "Excluded pattern count: {}".format(len(self.metadata_Dict["Mel_Length_Dict"]) - len(order)))
        root = hp.Train.Pattern_Path
        if not order:
            self._producer_error[is_Pre_Train] = "no pattern file passes the dataset / length filters (hp.Train.*_Dataset_List, Use_Wav_Length_Range)"
            return
        token_dict = self.metadata_Dict["Token_Index_Dict"]
        in_flight = deque()                  # batches being loaded by the worker processes, in the order they will enter the queue
        n_submitted, blocks = 0, []
        if self._pool is not None:
            from multiprocessing import shared_memory
            blocks = [shared_memory.SharedMemory(create=True, size=self._shm_size) for _ in range(self._workers + 1)]
            self._shm.extend(blocks)

        def emit(item):
            """One loaded batch into the queue (bounded, Feeder.py:126-127); False when the producer has to stop."""
            while len(queue) >= hp.Train.Max_Pattern_Queue and not self._stop:
                time.sleep(0.01)
            if self._stop:
                return False
            try:
                if self._pool is None:
                    queue.append(load_pattern_batch(item, token_dict, pattern_path=root))
                else:
                    fut, shm = item
                    kind, res = fut.result()
                    if kind == "shm":            # copy out of the block: it is handed to the next batch as soon as this one is in the queue
                        res = dict({"Is_Training": True}, **{k: np.ndarray(shape, np.dtype(dt), buffer=shm.buf, offset=off).copy() for k, dt, shape, off in res})
                    queue.append(res)
            except Exception as e:                  # a dead producer must not leave Get_Train_Pattern spinning forever
                self._producer_error[is_Pre_Train] = "{}: {}".format(type(e).__name__, e)
                print("Pattern producer stopped: {}".format(e))
                return False
            return True
        while not self._stop:
            batches = epoch_batches(order, rng)
            mine = batches[self.rank::self.world] if len(batches) >= self.world else batches       # tiny sets: every rank takes all
            for names in mine:
                if self._stop:
                    return
                if self._pool is None:
                    if not emit(names):
                        return
                    continue
                # (at most `workers` batches in flight and workers + 1 blocks in rotation: the block of batch k was copied out before batch
                #  k + workers + 1 is submitted)
                shm = blocks[n_submitted % len(blocks)]
                n_submitted += 1
                try:
                    in_flight.append((self._pool.submit(_load_pattern_batch_into, names, token_dict, root, shm.name, self._shm_size), shm))
                except RuntimeError:                # the pool was shut down (close())
                    return
                if len(in_flight) > self._workers and not emit(in_flight.popleft()):
                    return

    def Unget_Train_Pattern(self, pattern, is_Pre_Train=False):
        """Put a pattern taken ahead of time back at the head of its queue (the driver prefetched it for a step that then asked for the
        other queue: the switch from pre-training to main training, MSTTS_SV.py:268-269)."""
        if self.pattern_Queue is not None:
            (self.pre_Pattern_Queue if is_Pre_Train else self.pattern_Queue).appendleft(pattern)

    def close(self):
        """Stop the producer threads (the reference's daemon threads die with the process) and the loader processes."""
        self._stop = True
        pool, self._pool = getattr(self, "_pool", None), None
        if pool is not None:
            pool.shutdown(wait=True, cancel_futures=True)
        for shm in getattr(self, "_shm", []):
            try:
                shm.close()
                shm.unlink()
            except (OSError, BufferError):
                pass
        self._shm = []

    def Speaker_Embedding_Mel(self, mel_List):
        return speaker_windows(mel_List)

    def Get_Inference_Pattern(self, speaker_Wav_Path_List, text_List, speaker_Mel_List=None):
        """Feeder.py:186-233.  `speaker_Mel_List` ([T,80] arrays) may be given instead of wav paths."""
        from . import Audio
        token, length = tokenize(text_List, self.metadata_Dict["Token_Index_Dict"])
        if speaker_Mel_List is None:
            speaker_Mel_List = [
                np.transpose(Audio.melspectrogram(
                    y=load_wav(path), num_freq=hp.Sound.Spectrogram_Dim, frame_shift_ms=hp.Sound.Frame_Shift,
                    frame_length_ms=hp.Sound.Frame_Length, num_mels=hp.Sound.Mel_Dim, sample_rate=hp.Sound.Sample_Rate,
                    max_abs_value=hp.Sound.Max_Abs_Mel, device=self.device)).astype(np.float32)
                for path in speaker_Wav_Path_List]
        return {
            "Is_Training": False,
            "Token": token,
            "Token_Length": length,
            "Mel": np.zeros((len(text_List), 1, hp.Sound.Mel_Dim), np.float32),
            "Mel_Length": np.zeros(len(text_List), np.int32),
            "Speaker_Embedding_Mel": speaker_windows(speaker_Mel_List),
        }

    def Get_Train_Pattern(self, is_Pre_Train=False, batch_Size=None, token_Length=128, mel_Length=800, seed=1234, block=True):
        """Feeder.py:174-184 when pattern files exist (blocks until the producer has a batch); otherwise the synthetic
        pattern of the benchmark shape.  block=False: None instead of waiting when the producer's queue is empty (the driver's
        prefetch of the NEXT step's batch must not stall the step that is running)."""
        if self.pattern_Queue is not None and batch_Size is None:
            import time
            if is_Pre_Train and not hasattr(self, "pre_Pattern_Queue"):
                raise RuntimeError("pre-train patterns requested but hp.Train.Use_Pre_in_Main_Train is off")
            queue = self.pre_Pattern_Queue if is_Pre_Train else self.pattern_Queue
            if not block and len(queue) == 0:
                return None
            while len(queue) == 0:
                if is_Pre_Train in self._producer_error:
                    raise RuntimeError("training pattern producer stopped: " + self._producer_error[is_Pre_Train])
                time.sleep(0.01)
            return queue.popleft()
        B = batch_Size or hp.Train.Batch_Size
        g = np.random.default_rng(seed)
        token = g.integers(2, hp.Encoder.Embedding.Token_Size, size=(B, token_Length)).astype(np.int32)
        token[:, 0], token[:, -1] = 0, 1
        mel = np.clip(g.normal(0, 1.5, size=(B, mel_Length, hp.Sound.Mel_Dim)), -4, 4).astype(np.float32)
        return {"Is_Training": True, "Token": token, "Token_Length": np.full(B, token_Length, np.int32), "Mel": mel,
                "Mel_Length": np.full(B, mel_Length, np.int32), "Speaker_Embedding_Mel": speaker_windows(list(mel))}
