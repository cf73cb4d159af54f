"""Known-answer tests derivable exactly from reference data: token
dictionary, tokenisation of the reference's own inference sentences, speaker-window arithmetic,
STFT framing constants, normalisation end points, LR schedule, stop cut, Philox vectors."""
import json
import os

import numpy as np
import pytest

from oracle import feeder as OF, audio as OA, train as OT, rng as ORNG
from multi_speaker_tts_amd import Feeder as PF, Hyper_Parameters as hp

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")

SENTENCES = {   # Inference_Sentence_in_Train.txt -> tokens via Feeder.py:189-196
    "He that has no shame has no conscience.": [0, 21, 18, 2, 33, 21, 14, 33, 2, 21, 14, 32, 2, 27, 28, 2, 32, 21, 14, 26, 18, 2, 21, 14, 32, 2, 27, 28, 2, 16, 28, 27, 32, 16, 22, 18, 27, 16, 18, 10, 1],
    "Who knows much believes the less.": [0, 36, 21, 28, 2, 24, 27, 28, 36, 32, 2, 26, 34, 16, 21, 2, 15, 18, 25, 22, 18, 35, 18, 32, 2, 33, 21, 18, 2, 25, 18, 32, 32, 10, 1],
    "Things are always at their best in the beginning.": [0, 33, 21, 22, 27, 20, 32, 2, 14, 31, 18, 2, 14, 25, 36, 14, 38, 32, 2, 14, 33, 2, 33, 21, 18, 22, 31, 2, 15, 18, 32, 33, 2, 22, 27, 2, 33, 21, 18, 2, 15, 18, 20, 22, 27, 27, 22, 27, 20, 10, 1],
    "Please call Stella.": [0, 29, 25, 18, 14, 32, 18, 2, 16, 14, 25, 25, 2, 32, 33, 18, 25, 25, 14, 10, 1],
    "His voice is tested now.": [0, 21, 22, 32, 2, 35, 28, 22, 16, 18, 2, 22, 32, 2, 33, 18, 32, 33, 18, 17, 2, 27, 28, 36, 10, 1],
}


def test_token_dict_contract():
    d = OF.load_token_dict()
    assert len(d) == 42 == hp.Encoder.Embedding.Token_Size
    assert d["<S>"] == 0 and d["<E>"] == 1 and d[" "] == 2 and d["A"] == 14 and d["Z"] == 39 and d["]"] == 41
    assert sorted(d.values()) == list(range(42))


@pytest.mark.parametrize("mod", [OF, PF])
def test_tokenisation_kats(mod):
    for text, want in SENTENCES.items():
        tok, length = mod.tokenize([text])
        assert tok[0].tolist() == want and int(length[0]) == len(want)
    tok, length = mod.tokenize(list(SENTENCES))
    assert tok.shape == (5, 51) and tok.dtype == np.int32
    assert tok[3, 21:].tolist() == [1] * 30            # right-padded with <E> = 1
    with pytest.raises(KeyError):
        mod.tokenize(["unknown char: é"])


@pytest.mark.parametrize("mod", [OF, PF])
def test_speaker_windows(mod):
    assert mod.window_starts(400) == [104, 136, 168, 200, 232]
    assert mod.window_starts(192) == [0, 32, 64, 96, 128]
    assert mod.window_starts(801) == [304, 336, 368, 400, 432]
    assert mod.window_starts(100) is None
    mel = np.arange(400 * 80, dtype=np.float32).reshape(400, 80)
    w = mod.speaker_windows([mel])
    assert w.shape == (5, 64, 80) and np.array_equal(w[2], mel[168:232])
    short = np.ones((50, 80), np.float32)
    w = mod.speaker_windows([short])
    assert np.array_equal(w[:, :50], np.ones((5, 50, 80))) and not w[:, 50:].any()
    assert np.array_equal(OF.speaker_windows([mel, short]), PF.speaker_windows([mel, short]))


def test_stft_constants_and_normalisation():
    assert OA.stft_parameters(1025, 12.5, 50, 16000) == (2048, 200, 800)
    for n in (16000, 16199, 16200, 79800):
        assert OA.melspectrogram(np.zeros(n) + 1e-3).shape == (80, 1 + n // 200)
    S = np.array([-100.0, 0.0, -50.0, -200.0, 50.0])
    norm = np.clip(8 * ((S + 100) / 100) - 4, -4, 4)
    assert norm.tolist() == [-4.0, 4.0, 0.0, -4.0, 4.0]
    assert 20 * np.log10(max(1e-5, 0.0)) == -100.0
    fb = OA.mel_basis(16000, 2048, 80)
    assert fb.shape == (80, 1025) and (fb >= 0).all() and fb[:, 0].sum() == 0.0


def test_lr_schedule_and_stop_cut():
    assert OT.learning_rate(0) == 1e-3
    assert abs(OT.learning_rate(10000) - 5e-4) < 1e-12
    assert OT.learning_rate(10 ** 7) == 1e-5
    from multi_speaker_tts_amd.engine import learning_rate
    for s in (0, 1, 9999, 10000, 123456, 10 ** 7):
        assert abs(learning_rate(s) - OT.learning_rate(s)) < 1e-15
    assert OF.stop_cut([0.1, 0.2, 0.6, 0.9]) == 2 and OF.stop_cut([0.1, 0.5, 0.3]) == 3 and OF.stop_cut([0.9]) == 0
    assert PF.stop_cut([0.1, 0.2, 0.6, 0.9]) == 2 and PF.stop_cut([0.1, 0.5, 0.3]) == 3


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    r = ORNG.philox4x32_10([0], [0], [0], [0], 0, 0)
    assert [int(x[0]) for x in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = ORNG.philox4x32_10([0xffffffff], [0xffffffff], [0xffffffff], [0xffffffff], 0xffffffff, 0xffffffff)
    assert [int(x[0]) for x in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    m = ORNG.keep_mask((1000, 50), 1234, 7, 0.5)
    assert abs(m.mean() - 0.5) < 0.01


def test_golden_fixtures_match_oracle():
    """tests/golden/*.npz were produced by tests/golden/make_golden.py from the fp64 oracle; the oracle
    must keep reproducing them (guards against silent edits of the restatement)."""
    import torch
    from oracle import model as OM
    g = np.load(os.path.join(GOLD, "tiny_train_step.npz"))
    cfg = json.loads(str(g["cfg"]))
    d = OM.Dims(**cfg)
    params = OM.init_params(d, int(g["seed"]))
    batch = OT.synthetic_batch(d, int(g["B"]), int(g["Te"]), int(g["L"]), seed=int(g["seed"]), ragged=True)
    masks = OT.make_masks(d, int(g["B"]), int(g["Te"]), int(g["L"]) + 1, True, seed=OT.step_seed(1234, 0))
    _, _, sc, grads, out = OT.train_step(params, None, d, batch, masks, 0, return_grads=True)
    assert np.allclose(out["Mel"].numpy(), g["mel"], rtol=0, atol=1e-10)
    assert np.allclose(out["Attention_History"].numpy(), g["align"], rtol=0, atol=1e-10)
    assert abs(sc["Loss"] - float(g["loss"])) < 1e-10
    k = "decoder/decoder/attention_wrapper/multi_rnn_cell/cell_0/zoneout_lstm_cell/kernel"
    assert np.allclose(grads[k].numpy(), g["grad_cell0"], rtol=0, atol=1e-12)


def test_griffin_lim_export_path():
    """Audio.inv_spectrogram / _stft / _istft (Audio.py:15-27,50-74): librosa conventions restated in NumPy.
    iSTFT(STFT(y)) == y away from nothing (centre padding makes it exact everywhere), frame count = 1 + len // hop,
    and Griffin-Lim on a real spectrogram converges to a signal with (nearly) that magnitude."""
    from multi_speaker_tts_amd import Audio
    args = (1025, 12.5, 50, 16000)
    n_fft, hop, win = Audio._stft_parameters(*args)
    assert (n_fft, hop, win) == (2048, 200, 800)
    g = np.random.default_rng(0)
    t = np.arange(16000 // 4) / 16000.0
    # quiet enough that |STFT| < 10: the [0, 1] normalisation saturates at 0 dB after the 20 dB reference shift
    y = 0.016 * np.sin(2 * np.pi * 440 * t) + 0.008 * np.sin(2 * np.pi * 1330 * t + 0.3) + 0.0004 * g.normal(size=t.shape)
    D = Audio._stft(y, *args)
    assert D.shape == (1025, 1 + len(y) // hop) and np.abs(D).max() < 10
    back = Audio._istft(D, *args)
    assert back.shape[0] == hop * (D.shape[1] - 1) and np.abs(back - y[:back.shape[0]]).max() < 1e-9
    # normalised spectrogram of y -> waveform; compare magnitudes (phase is free)
    M = np.abs(D)
    S = np.clip((20 * np.log10(np.maximum(1e-5, M)) - 20 + 100) / 100, 0, 1)
    wav = Audio.inv_spectrogram(S, *args, power=1.0, griffin_lim_iters=40, rng=np.random.RandomState(1))
    assert wav.shape[0] == hop * (S.shape[1] - 1) and np.isfinite(wav).all()
    from scipy import signal
    M2 = np.abs(Audio._stft(signal.lfilter([1, -0.97], [1], wav), *args))          # undo inv_preemphasis
    keep = slice(2, -2)                                                            # edge frames see the reflect padding
    err = np.linalg.norm(M2[:, keep] - M[:, keep]) / np.linalg.norm(M[:, keep])
    assert err < 0.2, err
    assert np.allclose(Audio._denormalize(np.array([0.0, 0.5, 1.0, 2.0])), [-100, -50, 0, 0]) and np.isclose(Audio._db_to_amp(20.0), 10.0)


def test_pickle_feeder_and_pattern_files(tmp_path, monkeypatch):
    """The reference's on-disk pattern format and producer rules (Pattern_Generate.py:14-31,66-76,245-274; Feeder.py:43-56,89-184):
    protocol-2 pickles, METADATA.PICKLE keys + consistency check, dataset and length filters, length-sorted consecutive
    batches in shuffled order, <S>/<E> framing with <E> padding, zero-padded mels, speaker windows."""
    import pickle
    from multi_speaker_tts_amd import Hyper_Parameters as hp, Feeder as F, Pattern_Generate as PG
    assert PG.Text_Filtering(' please call "Stella" .  ') == "PLEASE CALL STELLA ." and PG.Text_Filtering("naïve") is None
    assert PG.Text_Filtering("'tis") is None and PG.Text_Filtering("Who knows ?") == "WHO KNOWS?"
    root = tmp_path / "patterns"
    monkeypatch.setattr(hp.Train, "Pattern_Path", str(root))
    monkeypatch.setattr(hp.Train, "Batch_Size", 3)
    monkeypatch.setattr(hp.Train, "Max_Pattern_Queue", 2)
    td = F.load_token_dict()
    g = np.random.default_rng(0)
    lens = {"VCTK.A.PICKLE": 120, "VCTK.B.PICKLE": 60, "TIMIT.C.PICKLE": 300, "VCTK.D.PICKLE": 45, "VCTK.E.PICKLE": 200,
            "LJ.F.PICKLE": 100, "VCTK.G.PICKLE": 30, "TIMIT.H.PICKLE": 721, "VCTK.I.PICKLE": 250}          # G too short (< 40 frames), H too long (> 720)
    for name, n in lens.items():
        PG.Pattern_File_Write(name, "HI %s." % name[-8], np.clip(g.normal(0, 1.5, (n, 80)), -4, 4), td, name.split(".")[0])
    with open(root / "VCTK.A.PICKLE", "rb") as f:
        raw = f.read()
        pd = pickle.loads(raw)
    assert raw[:2] == b"\x80\x02" and set(pd) == {"Token", "Mel", "Text", "Dataset"} and pd["Token"].dtype == np.int32 and pd["Mel"].dtype == np.float32
    assert list(pd["Token"]) == [td[c] for c in "HI A."]
    md = PG.Metadata_Generate()
    assert set(md) == {"Token_Index_Dict", "Spectrogram_Dim", "Mel_Dim", "Frame_Shift", "Frame_Length", "Sample_Rate", "File_List",
                       "Token_Length_Dict", "Mel_Length_Dict", "Dataset_Dict"} and len(md["File_List"]) == 9
    order = F.train_file_order(md)
    assert order == ["VCTK.D.PICKLE", "VCTK.B.PICKLE", "VCTK.A.PICKLE", "VCTK.E.PICKLE", "VCTK.I.PICKLE", "TIMIT.C.PICKLE"]    # LJ is pre-train only
    assert F.train_file_order(md, is_Pre_Train=True) == ["LJ.F.PICKLE"]
    import random
    batches = F.epoch_batches(order, random.Random(1))
    assert sorted(map(tuple, batches)) == sorted([tuple(order[:3]), tuple(order[3:])])
    feeder = F.Feeder(is_Training=True, device="cpu", seed=3)
    seen = []
    for _ in range(4):                                      # two epochs
        pat = feeder.Get_Train_Pattern()
        B = pat["Token"].shape[0]
        assert B == 3 and pat["Is_Training"] is True
        assert (pat["Token"][:, 0] == 0).all() and all(pat["Token"][i, pat["Token_Length"][i] - 1] == 1 for i in range(B))
        assert pat["Token_Length"].tolist() == [7, 7, 7] and pat["Mel"].shape == (3, pat["Mel_Length"].max(), 80)
        for i in range(B):
            assert (pat["Mel"][i, pat["Mel_Length"][i]:] == 0).all() and np.abs(pat["Mel"][i, :pat["Mel_Length"][i]]).max() > 0
        assert pat["Speaker_Embedding_Mel"].shape == (15, 64, 80)
        seen.append(tuple(pat["Mel_Length"].tolist()))
    assert sorted(seen[:2]) == sorted([(45, 60, 120), (200, 250, 300)]) and sorted(seen[2:]) == sorted(seen[:2])
    feeder.close()
    monkeypatch.setattr(hp.Sound, "Frame_Shift", 10.0)
    with pytest.raises(ValueError):
        F.Feeder(is_Training=True, device="cpu")


def test_tf_checkpoint_bundle_format(tmp_path):
    """TF V2 checkpoint (tensor bundle) reader/writer without TensorFlow: CRC-32C and snappy known answers, the table
    layout (footer magic, block trailers, prefix-compressed keys over several blocks), dtype/shape/offset bookkeeping."""
    import struct
    from multi_speaker_tts_amd import tf_checkpoint as tfc
    assert tfc.crc32c(b"123456789") == 0xE3069283                                   # the CRC-32C check value
    assert tfc.mask_crc(0) == 0xA282EAD8 and tfc.crc32c(b"") == 0
    assert tfc._snappy_decompress(bytes([0x0a, 0x00, 0x61, 0x15, 0x01])) == b"a" * 10
    g = np.random.default_rng(0)
    variables = {"encoder/conv_%d/conv1d/kernel" % i: g.normal(size=(5, 7, 3)).astype(np.float32) for i in range(40)}
    variables.update({"encoder/embedding_variable": g.normal(size=(42, 16)).astype(np.float32), "global_step": np.array(1234567, np.int64),
                      "decoder/x/bias": np.zeros((0,), np.float32), "lengths": np.arange(6, dtype=np.int32).reshape(2, 3),
                      "encoder/embedding_variable/Adam": g.normal(size=(42, 16)).astype(np.float32)})
    prefix = str(tmp_path / "ckpt" / "CHECKPOINT-1234567")
    tfc.write_checkpoint(prefix, variables)
    assert tfc.latest_checkpoint(str(tmp_path / "ckpt")) == prefix and tfc.latest_checkpoint(str(tmp_path)) is None
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xDB4775248B80FB57
    entries = tfc.list_variables(prefix)
    assert entries.pop("__num_shards__") == 1 and set(entries) == set(variables)
    assert entries["global_step"]["dtype"] == 9 and entries["global_step"]["shape"] == () and entries["lengths"]["dtype"] == 3
    assert entries["encoder/embedding_variable"]["shape"] == (42, 16) and entries["encoder/embedding_variable"]["size"] == 42 * 16 * 4
    keys = [k for k, _ in tfc.read_table(prefix + ".index")]
    assert keys == sorted(keys) and keys[0] == b"" and len(keys) == len(variables) + 1
    back = tfc.read_checkpoint(prefix)
    assert all(np.array_equal(back[k], v) and back[k].dtype == v.dtype and back[k].shape == v.shape for k, v in variables.items())
    some = tfc.read_checkpoint(prefix, names=["global_step", "nope"])
    assert list(some) == ["global_step"] and int(some["global_step"]) == 1234567
    # corruption is detected: flip one payload byte, then one index byte
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read()); data[10] ^= 0xFF
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(ValueError):
        tfc.read_checkpoint(prefix)
    assert np.array_equal(tfc.read_checkpoint(prefix, names=["lengths"])["lengths"], variables["lengths"])
    idx = bytearray(raw); idx[20] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError):
        tfc.read_table(prefix + ".index")


def _write_wav(path, seconds=0.6, rate=16000, seed=0):
    from scipy.io import wavfile
    os.makedirs(os.path.dirname(path), exist_ok=True)
    t = np.arange(int(seconds * rate)) / rate
    y = (0.4 * np.sin(2 * np.pi * (200 + 20 * seed) * t) * 32767).astype(np.int16)
    wavfile.write(path, rate, y)
    return y


def _write_sphere(path, samples, rate=16000):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    head = "NIST_1A\n   1024\nchannel_count -i 1\nsample_count -i %d\nsample_rate -i %d\nsample_n_bytes -i 2\nsample_byte_format -s2 01\nsample_coding -s3 pcm\nend_head\n" % (len(samples), rate)
    with open(path, "wb") as f:
        f.write(head.encode("ascii").ljust(1024, b" "))
        f.write(np.asarray(samples, "<i2").tobytes())


def test_corpus_walkers(tmp_path):
    """The five corpus layouts of Pattern_Generate.py:115-243 on miniature trees: which (audio, text) pairs come out, with
    Text_Filtering applied, missing audio / transcripts skipped, TEDLIUM <UNK> segments dropped, TIMIT's SPHERE-in-.WAV decoded."""
    from multi_speaker_tts_amd import Pattern_Generate as PG
    from multi_speaker_tts_amd import Feeder as F
    # LJSpeech
    lj = tmp_path / "LJ"
    _write_wav(str(lj / "wavs" / "LJ001-0001.wav"))
    _write_wav(str(lj / "wavs" / "LJ001-0003.wav"))
    (lj / "metadata.csv").write_text("LJ001-0001|Printing, in 1 sense|Printing, in the only sense\nLJ001-0002|no audio|no audio for this one\n"
                                     "LJ001-0003|bad ü char|bad ü char\n", encoding="utf-8")
    paths, texts = PG.LJ_Info_Load(str(lj))
    assert [os.path.basename(p) for p in paths] == ["LJ001-0001.wav"] and texts[paths[0]] == "PRINTING, IN THE ONLY SENSE"
    # VCTK
    vc = tmp_path / "VCTK"
    _write_wav(str(vc / "wav48" / "p225" / "p225_001.wav"))
    _write_wav(str(vc / "wav48" / "p225" / "p225_002.wav"))
    os.makedirs(vc / "txt" / "p225")
    (vc / "txt" / "p225" / "p225_001.txt").write_text("Please call Stella.\n")
    paths, texts = PG.VCTK_Info_Load(str(vc))
    assert [os.path.basename(p) for p in paths] == ["p225_001.wav"] and texts[paths[0]] == "PLEASE CALL STELLA."
    # LibriSpeech (wav stand-ins for the flac files)
    ls = tmp_path / "LS"
    _write_wav(str(ls / "17" / "363" / "17-363-0001.wav"))
    _write_wav(str(ls / "17" / "363" / "17-363-0002.wav"))
    (ls / "17" / "363" / "17-363.trans.txt").write_text("17-363-0001 WHO KNOWS MUCH BELIEVES THE LESS\n17-363-0002 A 2ND LINE WITH A DIGIT\n")
    paths, texts = PG.LS_Info_Load(str(ls))
    assert [os.path.basename(p) for p in paths] == ["17-363-0001.wav"] and texts[paths[0]] == "WHO KNOWS MUCH BELIEVES THE LESS"
    # TIMIT: NIST SPHERE behind a .WAV name, transcript `start end words`
    ti = tmp_path / "TIMIT"
    y = (np.arange(8000) % 200 - 100).astype(np.int16)
    _write_sphere(str(ti / "DR1" / "FCJF0" / "SA1.WAV"), y)
    (ti / "DR1" / "FCJF0" / "SA1.TXT").write_text("0 46797 She had your dark suit in greasy wash water all year.\n")
    paths, texts = PG.TIMIT_Info_Load(str(ti))
    assert len(paths) == 1 and texts[paths[0]] == "SHE HAD YOUR DARK SUIT IN GREASY WASH WATER ALL YEAR."
    rate, data = F.read_audio(paths[0])
    assert rate == 16000 and np.array_equal(data, y)
    assert F.load_wav(paths[0]).shape[0] > 0
    # TEDLIUM: segments from the stm file; <unk> segments dropped; SPHERE cut by time
    tl = tmp_path / "TL"
    _write_sphere(str(tl / "sph" / "talk1.sph"), np.arange(32000, dtype=np.int16))
    os.makedirs(tl / "stm")
    (tl / "stm" / "talk1.stm").write_text("talk1 1 spk 0.50 1.00 <o,f0,male> hello there it 's me\ntalk1 1 spk 1.00 1.50 <o,f0,male> this has <unk> inside\n")
    paths, segs = PG.TL_Info_Load(str(tl))
    assert len(paths) == 1 and segs[paths[0]] == [(0.5, 1.0, "HELLO THERE IT'S ME")]
    rate, cut = F.read_sphere(paths[0], 0.5, 1.0)
    assert rate == 16000 and cut.shape[0] == 8000 and int(cut[0]) == 8000
    # command line: nothing to do is an error, exactly like the reference
    import pytest
    with pytest.raises(ValueError):
        PG.main([], device="cpu")


def test_pattern_cli_per_corpus_rules(tmp_path, monkeypatch):
    """What the reference's command line passes per corpus (Pattern_Generate.py:318-404): TIMIT pattern names carry the speaker
    directory (its utterance names repeat from speaker to speaker - without the prefix all but one SA1 would be overwritten),
    LibriSpeech mels are generated with spectral subtraction and the other corpora without, and the TEDLIUM generator calls
    Mel_Generate with its defaults and stops at the first segment the length filter rejects (Pattern_Generate.py:80-113)."""
    from multi_speaker_tts_amd import Hyper_Parameters as hp, Pattern_Generate as PG
    ti = tmp_path / "TIMIT"
    y = (np.arange(8000) % 200 - 100).astype(np.int16)
    for spk in ("FCJF0", "MDAB0"):
        _write_sphere(str(ti / "DR1" / spk / "SA1.WAV"), y)
        (ti / "DR1" / spk / "SA1.TXT").write_text("0 46797 She had your dark suit in greasy wash water all year.\n")
    ls = tmp_path / "LS"
    _write_wav(str(ls / "17" / "363" / "17-363-0001.wav"))
    (ls / "17" / "363" / "17-363.trans.txt").write_text("17-363-0001 WHO KNOWS MUCH BELIEVES THE LESS\n")
    lj = tmp_path / "LJ"
    _write_wav(str(lj / "wavs" / "LJ001-0001.wav"))
    (lj / "metadata.csv").write_text("LJ001-0001|x|Printing, in the only sense\n", encoding="utf-8")
    tl = tmp_path / "TL"
    _write_sphere(str(tl / "sph" / "talk1.sph"), np.arange(32000, dtype=np.int16))
    os.makedirs(tl / "stm")
    (tl / "stm" / "talk1.stm").write_text("talk1 1 spk 0.10 0.40 <o> first one\ntalk1 1 spk 0.50 0.60 <o> too short\ntalk1 1 spk 1.00 1.50 <o> never reached\n")
    calls = []

    def fake_mel(path, spectral_Subtract=False, range_Ignore=False, device="cuda"):
        calls.append((os.path.basename(path), bool(spectral_Subtract), bool(range_Ignore)))
        if path.endswith(".wav") and "tmp" in os.path.basename(path).lower() and len(calls_tl) == 1:
            calls_tl.append(1)
            return None                                  # the second TEDLIUM segment is "out of range"
        if "tmp" in os.path.basename(path).lower():
            calls_tl.append(1)
        return np.zeros((5, hp.Sound.Mel_Dim), np.float32)

    calls_tl = []
    monkeypatch.setattr(PG, "Mel_Generate", fake_mel)
    monkeypatch.setattr(hp.Train, "Pattern_Path", str(tmp_path / "PAT"))
    n = PG.main(["-timit", str(ti), "-ls", str(ls), "-lj", str(lj), "-tl", str(tl), "-all"], device="cpu")
    files = sorted(f for f in os.listdir(tmp_path / "PAT") if f != hp.Train.Metadata_File.upper())
    assert files == ["LJ.LJ001-0001.PICKLE", "LS.17-363-0001.PICKLE", "TIMIT.FCJF0.SA1.PICKLE", "TIMIT.MDAB0.SA1.PICKLE", "TL.TALK1.0.PICKLE"], files
    assert n == 5
    by_name = {c[0]: c for c in calls if not c[0].lower().startswith("tmp")}
    assert by_name["17-363-0001.wav"][1:] == (True, True)            # LibriSpeech: spectral subtraction, -all forwarded
    assert by_name["LJ001-0001.wav"][1:] == (False, True) and by_name["SA1.WAV"][1:] == (False, True)
    tl_calls = [c for c in calls if c[0].lower().startswith("tmp")]
    assert len(tl_calls) == 2 and all(c[1:] == (False, False) for c in tl_calls)       # defaults; third segment never generated


def test_tf_bundle_reader_against_hand_assembled_bytes(tmp_path):
    """A checkpoint assembled BYTE BY BYTE here from the published formats (LevelDB table_format.md: prefix-compressed entries,
    restart array, 1-byte compression tag + masked CRC-32C trailer, 48-byte footer with magic 0xdb4775248b80fb57;
    tensor_bundle.proto: BundleHeaderProto / BundleEntryProto field numbers) - nothing of tf_checkpoint's writer is used, and the
    CRC is an independent bitwise implementation - must read back through read_checkpoint."""
    import struct
    from multi_speaker_tts_amd import tf_checkpoint as tfc

    def crc32c_bitwise(data):                              # reflected CRC-32C, polynomial 0x1EDC6F41
        crc = 0xFFFFFFFF
        for byte in data:
            crc ^= byte
            for _ in range(8):
                crc = (crc >> 1) ^ (0x82F63B78 & -(crc & 1))
        return crc ^ 0xFFFFFFFF

    def masked(crc):
        return (((crc >> 15) | ((crc << 17) & 0xFFFFFFFF)) + 0xA282EAD8) & 0xFFFFFFFF

    def with_trailer(block):                                # block | compression type 0 | masked crc32c(block + type)
        return block + b"\x00" + struct.pack("<I", masked(crc32c_bitwise(block + b"\x00")))

    assert crc32c_bitwise(b"123456789") == 0xE3069283
    w = np.array([[1.5, -2.0, 3.25], [0.0, 7.0, -8.5]], "<f4")
    n = np.array([7, -9], "<i4")
    data = w.tobytes() + n.tobytes()
    # BundleEntryProto: 1 dtype (varint), 2 shape (TensorShapeProto: 2 dim {1 size}), 4 offset, 5 size, 6 crc32c (fixed32)
    entry_w = bytes([0x08, 0x01, 0x12, 0x08, 0x12, 0x02, 0x08, 0x02, 0x12, 0x02, 0x08, 0x03, 0x28, 0x18, 0x35]) + struct.pack("<I", masked(crc32c_bitwise(w.tobytes())))
    entry_n = bytes([0x08, 0x03, 0x12, 0x04, 0x12, 0x02, 0x08, 0x02, 0x20, 0x18, 0x28, 0x08, 0x35]) + struct.pack("<I", masked(crc32c_bitwise(n.tobytes())))
    header = bytes([0x08, 0x01, 0x1A, 0x02, 0x08, 0x01])   # num_shards = 1, version { producer = 1 }; endianness LITTLE = default 0
    # data block: entries (shared, non_shared, value_len, key suffix, value); "a/c" shares the 2-byte prefix "a/" with "a/b"
    block = (bytes([0, 0, len(header)]) + header +
             bytes([0, 3, len(entry_w)]) + b"a/b" + entry_w +
             bytes([2, 1, len(entry_n)]) + b"c" + entry_n +
             struct.pack("<II", 0, 1))                      # one restart point at offset 0, restart count 1
    out = with_trailer(block)
    meta_off = len(out)
    meta = struct.pack("<II", 0, 1)                         # empty metaindex block
    out += with_trailer(meta)
    index_off = len(out)
    handle = bytes([0x00, len(block)])                      # BlockHandle = varint offset 0, varint size
    index = bytes([0, 3, len(handle)]) + b"a/d" + handle + struct.pack("<II", 0, 1)     # separator key >= last key of the block
    out += with_trailer(index)
    footer = bytes([meta_off, len(meta), index_off, len(index)])
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    prefix = str(tmp_path / "HAND-1")
    with open(prefix + ".index", "wb") as f:
        f.write(out)
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(data)
    listing = tfc.list_variables(prefix)
    assert listing["__num_shards__"] == 1 and listing["a/b"]["shape"] == (2, 3) and listing["a/c"]["offset"] == 24
    got = tfc.read_checkpoint(prefix)
    assert set(got) == {"a/b", "a/c"} and np.array_equal(got["a/b"], w) and np.array_equal(got["a/c"], n) and got["a/c"].dtype == np.int32
    # ... and the module's writer produces a file this same reader and the hand rules agree on: footer magic, trailer CRCs
    tfc.write_checkpoint(str(tmp_path / "w" / "W-2"), {"a/b": w, "a/c": n})
    raw = open(str(tmp_path / "w" / "W-2.index"), "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xDB4775248B80FB57
    first = raw.index(b"a/b")                               # the first data block starts at 0; find its extent from the footer's index handle
    _, p = tfc._get_varint(raw[-48:], 0); _, p = tfc._get_varint(raw[-48:], p)
    ioff, p = tfc._get_varint(raw[-48:], p); isize, p = tfc._get_varint(raw[-48:], p)
    assert first > 0 and struct.unpack("<I", raw[ioff + isize + 1:ioff + isize + 5])[0] == masked(crc32c_bitwise(raw[ioff:ioff + isize + 1]))
