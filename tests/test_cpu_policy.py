"""Host logic added in round 4 that needs no GPU: the adaptive fallback policy of the persistent launches (engine._persist_begin_step), the
layout contract the train engine's moving-statistics snapshot relies on (ParamStore.n_moving), the variable version counter the inference
engine keys its packed operands on, and the ABI version handshake of the binding."""
import types
import warnings

import numpy as np
import pytest
import torch

from helpers import dims_pair


def _policy_obj():
    from multi_speaker_tts_amd import engine as E
    o = types.SimpleNamespace(_persist_strikes=0, _persist_off=0, _persist_warned=False, persist_disabled_steps=0, persist_last_status=(256, 1, 0))
    o.begin = types.MethodType(E.TrainEngine._persist_begin_step, o)
    return E, o


def test_two_consecutive_fallbacks_start_a_cooldown_and_probe_again(monkeypatch):
    E, o = _policy_obj()
    monkeypatch.setattr(E, "PERSIST_COOLDOWN", 4)
    assert o.begin() is True                               # step 1: healthy
    assert o.begin() is True and o._persist_strikes == 0
    o._step_fell_back = True                               # step 2 fell back
    assert o.begin() is True and o._persist_strikes == 1   # one strike: still trying
    assert o.begin() is True and o._persist_strikes == 0   # a healthy step in between clears the strike
    o._step_fell_back = True
    assert o.begin() is True and o._persist_strikes == 1
    o._step_fell_back = True
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        for i in range(4):                                 # two in a row: four steps without the persistent launches ...
            assert o.begin() is False and o.persist_disabled_steps == i + 1
        assert o.begin() is True                           # ... then a probe
    assert len(rec) == 1 and "consecutive steps fell back" in str(rec[0].message)
    # a second cool-down does not warn again
    o._step_fell_back = True; o.begin(); o._step_fell_back = True
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        assert o.begin() is False
    assert not rec


def test_moving_statistics_are_one_contiguous_range_at_the_front_of_the_frozen_slab():
    from multi_speaker_tts_amd.params import ParamStore
    pd, _ = dims_pair()
    ps = ParamStore(pd, "cpu", seed=3)
    mov = [(ps.offset[n], int(np.prod(ps.shape[n]))) for n, _, _ in ps.table if n.endswith(("moving_mean", "moving_variance"))]
    assert mov and ps.n_moving == sum((n + 3) // 4 * 4 for _, n in mov)
    assert all(o + n <= ps.n_moving for o, n in mov)                               # every moving statistic inside [0, n_moving)
    others = [ps.offset[n] for n, _, _ in ps.table if not ps.trainable[n] and not n.endswith(("moving_mean", "moving_variance"))]
    assert others and min(others) >= ps.n_moving                                    # ... and nothing else
    # snapshot / restore of that range is what engine.forward does around a speculative forward tail
    snap = ps.frozen[:ps.n_moving].clone()
    ps.view("decoder/conv_0/batch_normalization/moving_mean").add_(1.0)
    assert not torch.equal(snap, ps.frozen[:ps.n_moving])
    ps.frozen[:ps.n_moving].copy_(snap)
    assert float(ps.view("decoder/conv_0/batch_normalization/moving_mean").abs().max()) == 0.0


def test_param_version_counts_loads():
    from multi_speaker_tts_amd.params import ParamStore
    pd, _ = dims_pair()
    ps = ParamStore(pd, "cpu", seed=3)
    v0 = ps.version
    ps.load({"encoder/embedding_variable": np.zeros(ps.shape["encoder/embedding_variable"], np.float32)})
    assert ps.version == v0 + 1


def test_abi_version_handshake(monkeypatch):
    from multi_speaker_tts_amd import lib
    L = lib.load()
    assert L.mstts_abi_version() == lib.ABI_VERSION
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "ABI_VERSION", lib.ABI_VERSION + 1)
    with pytest.raises(lib.MsttsError, match="ABI version"):
        lib.load()
    monkeypatch.setattr(lib, "ABI_VERSION", lib.ABI_VERSION - 1)
    monkeypatch.setattr(lib, "_lib", None)
    assert lib.load() is not None


def test_header_declares_the_round_4_entry_points():
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "mstts.h")).read()
    for name in ("mstts_decoder_infer_persistent", "mstts_persist_infer_supported", "mstts_persist_infer_pack", "mstts_persist_infer_ws_bytes",
                 "mstts_persist_lstm_fwd_supported_n", "mstts_persist_lstm_pack_fwd", "mstts_lsa_param_bwd_ws_floats", "mstts_debug_park_cus"):
        assert re.search(r"\b%s\s*\(" % name, text), name
    from multi_speaker_tts_amd import lib
    L = lib.load()
    # shape rules that need no device: outside the reference widths / 32 rows / 256 positions the answer is 0 before any device query
    assert L.mstts_persist_infer_supported(33, 1024, 256, 768, 128, 128, 31, 80) == 0
    assert L.mstts_persist_infer_supported(16, 1024, 256, 768, 128, 257, 31, 80) == 0
    assert L.mstts_persist_infer_supported(16, 512, 256, 768, 128, 128, 31, 80) == 0
    assert L.mstts_persist_fwd_supported(32, 1024, 768, 128, 257, 31) == 0 and L.mstts_persist_bwd_supported(33, 1024, 768, 128, 128, 31) == 0
    assert L.mstts_persist_lstm_fwd_supported_n(16, 64, 2) == 0
    assert L.mstts_lsa_param_bwd_ws_floats(32, 128, 801) == (32 * 4 * 16 * 34 * 128 + 2 * 16 * 34 * 128)
    assert L.mstts_persist_infer_pack_floats() == 256 * 8 * 8 * 64


def test_waveglow_conv_piece_rule_is_a_fixed_function_of_the_shape():
    """WaveGlowEngine picks the dilated convolution's form (one reduction piece / two onto a zeroed buffer) from the shape alone - never from a
    timing - so that one latent seed gives the same samples in every process; the rule reproduces the measured table of tools/wg_conv_ab.py."""
    from multi_speaker_tts_amd.waveglow import _conv_two_pieces
    measured = {1: True, 2: False, 3: False, 4: True, 5: True, 6: False, 8: False, 10: False, 12: True, 16: True, 24: False, 32: False}
    for batch, two in measured.items():
        assert _conv_two_pieces(batch * 1376, 1024) is two, batch
    assert _conv_two_pieces(160, 1024) is True            # the 2-frame parity case: a fraction of the chip
