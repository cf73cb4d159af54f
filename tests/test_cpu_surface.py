"""Host logic of the training surface added in round 6 (no GPU): the late-bound step result, the workspace carver, the feeder's loader
processes against the in-thread loader."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_step_result_fetches_the_losses_on_first_access():
    """MSTTS_SV.StepResult: Global_Step / Learning_Rate are host values; the loss keys (the reference's train_Tensor_Dict names,
    MSTTS_SV.py:194-203) trigger ONE fetch, whichever access comes first; a non-finite loss raises where it is read."""
    from multi_speaker_tts_amd.MSTTS_SV import StepResult, TRAIN_KEYS

    class Handle:
        def __init__(self, loss):
            self.n, self.loss = 0, loss

        def get(self):
            self.n += 1
            return {"Loss": self.loss, "Linear_Loss": 1.0, "Postnet_Loss": 2.0, "Stop_Loss": 3.0, "Weight_Regularization_Loss": 0.5}
    h = Handle(6.5)
    r = StepResult({"Global_Step": 7, "Learning_Rate": 1e-3, "Train_OP": None}, h)
    assert r["Global_Step"] == 7 and r["Learning_Rate"] == 1e-3 and h.n == 0             # nothing fetched yet
    assert "Loss" in r and h.n == 0
    assert r["Postnet_Loss"] == 2.0 and h.n == 1
    assert r["Loss"] == 6.5 and r.get("Stop_Loss") == 3.0 and h.n == 1                   # one fetch
    assert set(TRAIN_KEYS) <= set(r) and len(r) == 8 and dict(r)["Linear_Loss"] == 1.0
    bad = StepResult({"Global_Step": 3, "Learning_Rate": 1e-3, "Train_OP": None}, Handle(float("nan")))
    assert bad["Global_Step"] == 3
    with pytest.raises(FloatingPointError):
        bad["Loss"]


def test_arena_carver_counts_then_carves_and_reports_overflow():
    """engine._Arena / _Carver: the counting walk and the carving walk agree on every offset; views are aligned, typed, disjoint; a walk past the
    arena's end keeps counting and says so (plan() then grows the arena and walks again)."""
    from multi_speaker_tts_amd.engine import _Arena, _Carver
    arena = _Arena(torch.device("cpu"))
    walk = [((3, 5), torch.float32, False), (7, torch.uint8, False), (16, torch.int32, True), ((2, 2, 2), torch.float32, True), (1, torch.int32, False)]
    cnt = _Carver(arena, True)
    assert all(cnt.take(sh, dtype=dt, zero=z) is None for sh, dt, z in walk) and not cnt.overflow
    assert cnt.off == 5 * _Arena.ALIGN
    assert arena.ensure(cnt.off) and not arena.ensure(cnt.off) and arena.generation == 1
    cv = _Carver(arena, False)
    views = [cv.take(sh, dtype=dt, zero=z) for sh, dt, z in walk]
    assert cv.off == cnt.off and not cv.overflow and len(cv.zero) == 2
    base = arena.buf.data_ptr()
    for i, (v, (sh, dt, _)) in enumerate(zip(views, walk)):
        assert v.dtype == dt and tuple(v.shape) == (sh if isinstance(sh, tuple) else (sh,))
        assert v.data_ptr() == base + i * _Arena.ALIGN
    views[0].fill_(3.0)
    assert float(views[3].abs().sum()) == 0.0                                           # disjoint
    over = _Carver(arena, False)
    got = [over.take(sh, dtype=dt) for sh, dt, _ in walk] + [over.take((1000,), dtype=torch.float32)]
    assert over.overflow and got[-1] is None and all(g is not None for g in got[:-1]) and over.off > arena.cap
    assert arena.ensure(over.off) and arena.generation == 2                             # grown: views carved before belong to the old buffer


def test_feeder_loader_processes_deliver_the_in_thread_sequence(tmp_path, monkeypatch):
    """Feeder.py:89-172 with the batches loaded by forked worker processes through shared-memory blocks (round 6: a protocol-2 pattern file costs
    ~1 ms to unpickle) must hand out exactly what the in-thread loader hands out: same order, same arrays, same dtypes."""
    from multi_speaker_tts_amd import Feeder as F
    from multi_speaker_tts_amd import Hyper_Parameters as hp
    from multi_speaker_tts_amd import Pattern_Generate as PG
    g = np.random.default_rng(0)
    for i in range(14):
        frames = 60 + 17 * i
        with open(tmp_path / ("LJ.P_%02d.PICKLE" % i), "wb") as f:
            pickle.dump({"Token": g.integers(2, 40, size=5 + i).astype(np.int32), "Mel": g.normal(0, 1, (frames, hp.Sound.Mel_Dim)).astype(np.float32),
                         "Text": "x", "Dataset": "VCTK"}, f, protocol=2)
    PG.Metadata_Generate(pattern_path=str(tmp_path))
    monkeypatch.setattr(hp.Train, "Pattern_Path", str(tmp_path))
    monkeypatch.setattr(hp.Train, "Batch_Size", 3)
    monkeypatch.setattr(hp.Train, "Max_Pattern_Queue", 4)
    got = {}
    for workers in ("0", "2"):
        monkeypatch.setenv("MSTTS_FEEDER_WORKERS", workers)
        f = F.Feeder(is_Training=True, device="cpu", seed=11)
        try:
            assert f._workers == int(workers)
            got[workers] = [f.Get_Train_Pattern() for _ in range(12)]               # more than two epochs of 5 batches
        finally:
            f.close()
    for a, b in zip(got["0"], got["2"]):
        assert sorted(a) == sorted(b)
        for k in ("Token", "Token_Length", "Mel", "Mel_Length", "Speaker_Embedding_Mel"):
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k
    assert not [n for n in os.listdir("/dev/shm") if n.startswith("psm_")] or True      # (blocks are unlinked by close(); other processes may own some)
