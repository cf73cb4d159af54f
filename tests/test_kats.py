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
