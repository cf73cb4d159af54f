"""The persistent FREE-RUNNING decoder loop (csrc/persist_infer.hip, mstts_decoder_infer_persistent: every step of Modules.py:397-443 in
inference mode - own frame -> prenet -> next input, stop gating Modules.py:212-237 - in one launch) against the launch-per-step driver it
replaces (mstts_decoder_infer_steps) on the same engine, weights, inputs and prenet keep-masks, and against the fp64 oracle.  Oracle parity
of the same path at depth, rows stopping at different steps, is tests/test_gpu_depth.py::test_depth_parity_free_running (B = 4, >= 100 steps)
and ::test_depth_parity_free_running_config4 (BASELINE configs[3]'s shape: B = 16, up to 128 tokens, >= 400 steps)."""
import numpy as np
import pytest
import torch

from helpers import dims_pair, rel_err, t2n
from oracle import model as OM, train as OT

pytestmark = pytest.mark.gpu

REFW = dict(dec_lstm=1024, prenet=256, enc_lstm=256, spk=256, n_mel=80)


def _setup(dev, B, Te, max_inf, seed, stop_bias=-8.0, stop_scale=1.0, **more):
    from multi_speaker_tts_amd.inference import InferEngine
    pd, od = dims_pair(max_inf=max_inf, **REFW, **more)
    values = OM.init_params(od, seed)
    g = np.random.default_rng(seed + 1)
    for k in values:
        if k.endswith(("bias", "beta")) and "highway" not in k:
            values[k] = g.normal(0, 0.1, values[k].shape)
    pk, bk = "decoder/decoder/linear_projection/dense/kernel", "decoder/decoder/linear_projection/dense/bias"
    values[pk] = np.array(values[pk]); values[pk][:, -1] *= stop_scale
    values[bk] = np.array(values[bk]); values[bk][-1] = stop_bias
    tok = g.integers(2, od.n_tok, size=(B, Te)).astype(np.int32)
    lengths = np.concatenate([[Te], g.integers(3, Te + 1, B - 1)]).astype(np.int32) if B > 1 else np.array([Te], np.int32)
    for b in range(B):
        tok[b, 0] = 0; tok[b, lengths[b] - 1] = 1; tok[b, lengths[b]:] = 1
    spk = g.normal(0, 1, (B, od.spk)); spk = (spk / np.sqrt((spk ** 2).sum())).astype(np.float32)
    masks = {k: v.numpy() for k, v in OT.make_masks(od, B, Te, od.max_inf + 1, False, seed=seed + 2).items()}
    eng = InferEngine(pd, device=dev, values=values)
    pat = {"Token": tok, "Token_Length": lengths, "Speaker_Embedding": spk}
    return eng, od, values, pat, masks


def _skip_unless_supported(eng, B, Te):
    from multi_speaker_tts_amd import lib
    d = eng.d
    if not lib.load().mstts_persist_infer_supported(B, d.dec_lstm, d.prenet, d.mem, d.att, Te, d.att_k, d.n_mel):
        pytest.skip("persistent free-running loop not available on this device (needs 256 CUs, one workgroup per CU)")


KEYS = ("Linear", "Mel", "Stop_Logit", "Attention_History")


@pytest.mark.parametrize("B,Te,max_inf", [(16, 128, 39), (32, 128, 24), (5, 23, 11), (1, 7, 5), (16, 100, 150),
                                            (16, 160, 20), (32, 256, 12), (4, 131, 30), (17, 129, 9)])      # > 128 tokens: the 256-position instantiations (one and two row tiles)
def test_persistent_equals_launch_per_step(dev, B, Te, max_inf):
    """Nobody stops early (stop bias -8): both drivers run max_inf + 1 steps; every output agrees to fp32 rounding (the persistent loop
    forms the first prenet layer from (m1, alignment) instead of from the frame, and sums its products in another order)."""
    eng, od, values, pat, masks = _setup(dev, B, Te, max_inf, seed=21)
    _skip_unless_supported(eng, B, Te)
    a = eng.forward(pat, masks=masks, with_vocoder=False)
    assert eng.persist_infer_launches == 1 and eng.persist_infer_fallbacks == 0, eng.persist_infer_status
    assert eng.persist_infer_status[4] == max_inf + 1                       # the forced stop at Max_Inference_Length ended the loop
    eng.persist_infer = False
    b = eng.forward(pat, masks=masks, with_vocoder=False)
    assert eng.persist_infer_launches == 1
    assert a["Linear"].shape == b["Linear"].shape == (B, max_inf + 1, od.n_mel)
    tol = 2e-4 if max_inf > 100 else 5e-5
    for k in KEYS:
        assert np.isfinite(a[k]).all(), k
        assert rel_err(a[k], b[k]) < tol, (k, rel_err(a[k], b[k]))


def test_rows_stop_at_different_steps(dev):
    """Stop gating inside the launch (Modules.py:216-219,395,409): a stop bias chosen from a first run so that the rows' stop logits cross
    0 at different steps; the launch must end exactly where the launch-per-step driver's host loop ends (the step at which the LAST row
    has finished), its control words must say so, and both must produce the same frames up to there."""
    B, Te, max_inf = 8, 64, 79
    eng, od, values, pat, masks = _setup(dev, B, Te, max_inf, seed=5, stop_bias=-100.0, stop_scale=-6.0)
    _skip_unless_supported(eng, B, Te)
    eng.persist_infer = False
    raw = eng.forward(pat, masks=masks, with_vocoder=False)["Stop_Logit"] + 100.0          # [B, S] bias-free logits (the trajectory does not depend on the bias)
    assert raw.shape == (B, max_inf + 1)
    best = None
    lo = -np.sort(raw.max(axis=1))[0]                                                       # from here on every row crosses 0 somewhere
    for beta in lo + np.concatenate([np.linspace(0.0005, 0.05, 100), np.linspace(0.05, 3.0, 600)]):
        z = raw + beta
        first = np.array([int(np.argmax(z[b] >= 0)) if (z[b] >= 0).any() else -1 for b in range(B)])
        if (first < 0).any() or len(set(first.tolist())) < 2:
            continue
        S = int(first.max()) + 1
        margin = float(np.abs(z[:, :S]).min())
        if S >= 3 and S < max_inf and margin > 1e-4 and (best is None or (len(set(first.tolist())), margin) > (len(set(best[2].tolist())), best[0])):
            best = (margin, float(beta), first, S)
    assert best is not None, ("no stop bias makes the rows stop at two or more different steps", raw.max(axis=1), raw.argmax(axis=1))
    margin, beta, first, S = best
    values["decoder/decoder/linear_projection/dense/bias"][-1] = beta
    eng.params.load(values)
    b = eng.forward(pat, masks=masks, with_vocoder=False)
    assert b["Linear"].shape[1] == S
    eng.persist_infer = True
    a = eng.forward(pat, masks=masks, with_vocoder=False)
    assert eng.persist_infer_launches == 1 and eng.persist_infer_fallbacks == 0, eng.persist_infer_status
    arrivals, abort, left, finished_rows, steps = eng.persist_infer_status
    assert (abort, left, finished_rows, steps) == (0, 256, B, S), eng.persist_infer_status
    assert a["Linear"].shape == (B, S, od.n_mel)
    for k in KEYS:
        assert rel_err(a[k], b[k]) < 5e-5, (k, rel_err(a[k], b[k]))
    cut = lambda stop: [int(np.argmax(stop[r] > 0.5)) if (stop[r] > 0.5).any() else stop.shape[1] for r in range(B)]     # MSTTS_SV.py:395
    assert cut(a["Stop"]) == cut(b["Stop"]) == first.tolist()
    print("rows stop at %s (margin %.3g), launch ended after %d steps" % (first.tolist(), margin, S))


def test_abort_falls_back_and_recovers(dev):
    """Self-test knob: workgroup 0 raises the abort word at step 3; the launch drains, the engine decodes launch by launch (same result),
    the next decode is one healthy launch again; two consecutive failures start the cool-down."""
    import warnings
    B, Te, max_inf = 6, 40, 9
    eng, od, values, pat, masks = _setup(dev, B, Te, max_inf, seed=9)
    _skip_unless_supported(eng, B, Te)
    ref = eng.forward(pat, masks=masks, with_vocoder=False)
    assert eng.persist_infer_launches == 1
    eng.persist_infer_selftest = 4
    a = eng.forward(pat, masks=masks, with_vocoder=False)
    assert eng.persist_infer_fallbacks == 1 and eng.persist_infer_status[1] == 3 and eng.persist_infer_status[2] < 256
    for k in KEYS:
        assert rel_err(a[k], ref[k]) < 5e-5, k
    eng.persist_infer_selftest = 0
    c = eng.forward(pat, masks=masks, with_vocoder=False)
    assert eng.persist_infer_launches == 2 and eng.persist_infer_fallbacks == 1
    for k in KEYS:
        assert rel_err(c[k], ref[k]) < 5e-6, k                              # (the launch itself is deterministic; the encoder's tail-split products upstream of it are not, to the last bit)
    eng.persist_infer_selftest = 2
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        eng.forward(pat, masks=masks, with_vocoder=False)
        eng.forward(pat, masks=masks, with_vocoder=False)
        assert eng.persist_infer_fallbacks == 3
        eng.forward(pat, masks=masks, with_vocoder=False)                   # cooling down: not attempted
        assert eng.persist_infer_fallbacks == 3 and eng.persist_disabled_decodes == 1
    assert sum("consecutive persistent decoder launches gave up" in str(r.message) for r in rec) == 1


def test_against_oracle_reference_widths(dev):
    """fp64 oracle, reference widths, ragged lengths, injected masks, on the persistent path (zero fallbacks asserted)."""
    B, Te, max_inf = 5, 23, 15
    eng, od, values, pat, masks = _setup(dev, B, Te, max_inf, seed=33)
    _skip_unless_supported(eng, B, Te)
    tm = {k: torch.from_numpy(v) for k, v in masks.items()}
    ob = {"Token": torch.from_numpy(pat["Token"]), "Token_Length": torch.from_numpy(pat["Token_Length"]), "Mel": torch.zeros(B, 1, od.n_mel, dtype=torch.float64),
          "Mel_Length": torch.zeros(B, dtype=torch.int32), "Speaker_Embedding": torch.tensor(pat["Speaker_Embedding"], dtype=torch.float64)}
    with torch.no_grad():
        ref = OM.forward(OM.to_torch(values), od, ob, False, tm, with_vocoder=False)
    got = eng.forward(pat, masks=masks, with_vocoder=False)
    assert eng.persist_infer_launches == 1 and eng.persist_infer_fallbacks == 0
    assert got["Linear"].shape == tuple(ref["Linear"].shape)
    for k in ("Linear", "Mel", "Stop", "Attention_History"):
        assert rel_err(got[k], t2n(ref[k])) < 1e-3, (k, rel_err(got[k], t2n(ref[k])))


def test_whole_forward_takes_the_persistent_recurrences(dev):
    """BASELINE configs[3] end to end (speaker encoder on 5 x 64-frame windows, text encoder, free-running decoder, postnet, Taco1 mel ->
    spectrogram): every recurrence is ONE launch - speaker stack 3, encoder BiLSTM 1, decoder 1, vocoder BiRNN (H = 128) 1 - with no
    fallback, and the result equals the launch-per-step forward."""
    from multi_speaker_tts_amd import lib
    B, Te, max_inf = 4, 40, 30
    eng, od, values, pat, masks = _setup(dev, B, Te, max_inf, seed=13, birnn=128, spk_lstm=256)      # the recurrent widths of the reference
    _skip_unless_supported(eng, B, Te)
    d = eng.d
    assert lib.load().mstts_persist_lstm_fwd_supported_n(B, d.birnn, 2) and lib.load().mstts_persist_lstm_fwd_supported_n(d.spk_samples * B, d.spk_lstm, 1)
    g = np.random.default_rng(3)
    pat = dict(pat)
    del pat["Speaker_Embedding"]
    pat["Speaker_Embedding_Mel"] = g.normal(0, 1, (d.spk_samples * B, d.spk_frames, d.n_mel)).astype(np.float32)
    a = eng.forward(pat, masks=masks, with_vocoder=True)
    L = lib.load()
    expect = (d.spk_lstm_n if L.mstts_persist_lstm_fwd_supported_n(d.spk_samples * B, d.spk_lstm, 1) and d.spk == d.spk_lstm else 0) + \
             (1 if L.mstts_persist_lstm_fwd_supported_n(B, d.enc_lstm, 2) else 0) + (1 if L.mstts_persist_lstm_fwd_supported_n(B, d.birnn, 2) else 0)
    assert eng.persist_lstm_launches == expect and eng.persist_lstm_fallbacks == 0 and eng.persist_infer_launches == 1, (eng.persist_lstm_launches, expect)
    eng.persist_lstm = eng.persist_infer = False
    b = eng.forward(pat, masks=masks, with_vocoder=True)
    assert eng.persist_lstm_launches == expect
    for k in KEYS + ("Spectrogram", "Speaker_Embedding"):
        assert np.isfinite(a[k]).all() and rel_err(a[k], b[k]) < 1e-4, (k, rel_err(a[k], b[k]))
