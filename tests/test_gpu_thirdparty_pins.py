"""HIP kernels against third-party implementations directly (not through the oracle): the STFT / mel launch against a pipeline built
only from scipy + torch + transformers parts, and the dynamic_rnn drivers against torch.nn.LSTM over packed sequences.  The oracle
is pinned to the same third parties on the CPU side (tests/test_cpu_thirdparty_pins.py); these close the triangle on the device."""
import ctypes as C

import numpy as np
import pytest
import torch

from multi_speaker_tts_amd import lib
from tests.helpers import t2n

pytestmark = pytest.mark.gpu


def test_stft_mel_launch_against_scipy_torch_transformers(dev):
    """Audio.melspectrogram (Audio.py:29-32) = lfilter([1, -0.97]) -> librosa.stft(2048, 200, 800) -> Slaney mel bank -> dB -> [-4, 4]."""
    from scipy import signal
    from transformers.audio_utils import mel_filter_bank
    from multi_speaker_tts_amd import Audio
    g = np.random.default_rng(11)
    t = np.arange(16000 * 3) / 16000.0
    y = (0.3 * np.sin(2 * np.pi * 180 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.08 * np.sin(2 * np.pi * 2500 * t)
         + 0.02 * g.normal(size=t.shape)).astype(np.float32)
    pre = signal.lfilter([1, -0.97], [1], y.astype(np.float64))
    D = torch.stft(torch.tensor(pre), 2048, hop_length=200, win_length=800, window=torch.hann_window(800, periodic=True, dtype=torch.float64),
                   center=True, pad_mode="reflect", return_complex=True).abs().numpy()
    fb = mel_filter_bank(1025, 80, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney").T
    ref_mel = np.clip(8 * ((20 * np.log10(np.maximum(1e-5, fb @ D)) + 100) / 100) - 4, -4, 4)
    ref_spec = np.clip((20 * np.log10(np.maximum(1e-5, D)) - 20 + 100) / 100, 0, 1)          # Audio.py:19-22: ref_level_db 20
    got = Audio.melspectrogram(y, 1025, 12.5, 50, 80, 16000, max_abs_value=4, device=dev)
    assert got.shape == ref_mel.shape == (80, 241)
    assert np.abs(got - ref_mel).max() < 2e-3                 # [-4, 4] range: 5e-4 relative, the tolerance of the oracle test
    spec = Audio.spectrogram(y, 1025, 12.5, 50, 16000, device=dev)
    assert spec.shape == ref_spec.shape and np.abs(spec - ref_spec).max() < 1e-3


@pytest.mark.parametrize("B,T,H,cin", [(6, 11, 64, 24), (32, 16, 256, 40)])
def test_lstm_seq_pair_against_torch_packed_bilstm(dev, B, T, H, cin):
    """mstts_lstm_seq_fwd_pair with zoneout rate 0 is tf.nn.bidirectional_dynamic_rnn over plain LSTM cells (forget_bias 1): torch's packed
    bidirectional nn.LSTM in fp64 with the TF kernel re-ordered (i, j, f, o -> i, f, g, o) - zero outputs past each row's length, the
    backward direction started at each row's own last step."""
    L = lib.load()
    g = np.random.default_rng(5)
    lengths_np = np.concatenate([[T], g.integers(1, T + 1, B - 1)]).astype(np.int32)
    lengths = torch.tensor(lengths_np, device=dev)
    x = g.normal(size=(B, T, cin))
    kern = [g.normal(0, 1.0 / np.sqrt(cin + H), size=(cin + H, 4 * H)) for _ in range(2)]
    bias = [g.normal(0, 0.3, size=4 * H) for _ in range(2)]
    ones = torch.ones(T, B, H, dtype=torch.uint8, device=dev)
    out = torch.zeros(B, T, 2 * H, device=dev)
    keep, descs = [], []
    for d in range(2):
        k32 = torch.tensor(kern[d], dtype=torch.float32, device=dev)
        xw = torch.zeros(B, T, 4 * H, device=dev)
        xin = torch.tensor(x, dtype=torch.float32, device=dev)
        lib.gemm(xin, k32, xw, B * T, 4 * H, cin, cin, 4 * H, 4 * H, bias=torch.tensor(bias[d], dtype=torch.float32, device=dev))
        wh = k32[cin:].contiguous()
        whp = torch.zeros(H * 4 * H, device=dev)
        lib.call("mstts_pack_cell_fwd", lib.ptr(wh), 4 * H, lib.ptr(whp), H, H)
        t = dict(xw=xw, wh=wh, whp=whp, hp=torch.zeros(2 * int(L.mstts_cell_act_floats(B, H)), device=dev),
                 c=torch.zeros(T + 1, B, H, device=dev), h=torch.zeros(T + 1, B, H, device=dev), acts=torch.zeros(T, B, 4 * H, device=dev),
                 craw=torch.zeros(T, B, H, device=dev), ws=torch.zeros(int(L.mstts_lstm_seq_ws_floats(B, H, 0)), device=dev))
        q = lib.LstmSeqFwd()
        q.B, q.T, q.H = B, T, H
        q.xw, q.wh, q.wh_ld, q.lengths, q.reverse, q.zoneout = lib.ptr(xw), lib.ptr(wh), 4 * H, lib.ptr(lengths), d, 0.0
        q.zc, q.zh = lib.ptr(ones), lib.ptr(ones)
        q.out, q.out_sb, q.out_st = lib.ptr(out, d * H), T * 2 * H, 2 * H
        q.c_hist, q.h_hist, q.acts, q.c_raw, q.gates_ws = lib.ptr(t["c"]), lib.ptr(t["h"]), lib.ptr(t["acts"]), lib.ptr(t["craw"]), lib.ptr(t["ws"])
        q.wh_p, q.h_p = lib.ptr(whp), lib.ptr(t["hp"])
        keep.append(t)
        descs.append(q)
    lib.call("mstts_lstm_seq_fwd_pair", C.byref(descs[0]), C.byref(descs[1]))
    torch.cuda.synchronize()

    lstm = torch.nn.LSTM(cin, H, batch_first=True, bidirectional=True).double()
    for d, sfx in enumerate(("", "_reverse")):
        k, b = torch.tensor(kern[d]), torch.tensor(bias[d])
        i, j, f, o = k.chunk(4, dim=1)
        w = torch.cat([i, f, j, o], dim=1)
        bi, bj, bf, bo = b.chunk(4)
        with torch.no_grad():
            getattr(lstm, "weight_ih_l0" + sfx).copy_(w[:cin].t())
            getattr(lstm, "weight_hh_l0" + sfx).copy_(w[cin:].t())
            getattr(lstm, "bias_ih_l0" + sfx).copy_(torch.cat([bi, bf + 1.0, bj, bo]))
            getattr(lstm, "bias_hh_l0" + sfx).zero_()
    packed = torch.nn.utils.rnn.pack_padded_sequence(torch.tensor(x), torch.tensor(lengths_np).long(), batch_first=True, enforce_sorted=False)
    with torch.no_grad():
        ref, _ = torch.nn.utils.rnn.pad_packed_sequence(lstm(packed)[0], batch_first=True, total_length=T)
    got = t2n(out).astype(np.float64)
    assert np.abs(got - ref.numpy()).max() < 2e-5
    for b_ in range(B):
        assert np.abs(got[b_, lengths_np[b_]:]).max(initial=0.0) == 0.0
