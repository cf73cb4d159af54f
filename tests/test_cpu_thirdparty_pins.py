"""Pins of the oracle's restated LIBRARY semantics against independent third-party implementations that are installed here.

The reference's arithmetic lives in TensorFlow 1.x and librosa, neither of which can be imported, so the oracle restates
those library ops.  The reference itself still cannot be run - parity stays "unpinned" in the sense of the contract - but every op
below is checked against somebody else's implementation of the same published definition, so that a misreading of the library
semantics in `oracle/` (and in the host-side NumPy export path of `multi_speaker_tts_amd/Audio.py`) cannot go unnoticed:

  scipy.signal   lfilter (the very call of Audio.py:13,16), get_window (what librosa.stft calls), istft
  torch          stft (center / reflect / centred short window: documented to follow librosa), conv1d(padding='same'), batch_norm,
                 LSTMCell, packed (bi)LSTM = dynamic_rnn's length semantics incl. reverse-by-length, Adam, l1 / mse / BCE-with-logits
  transformers   audio_utils.mel_filter_bank(norm='slaney', mel_scale='slaney') - a re-implementation of librosa.filters.mel

Where TF and the third party differ by definition (BN moving variance, Adam's epsilon) the test states the published difference and
checks the oracle sits on the TF side of it."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import audio as OA, model as OM, train as OT, np_ops as ON


def _rng(seed=0):
    return np.random.default_rng(seed)


# ----------------------------------------------------------------------------------------------------------------- audio (a1, f.4)
def test_preemphasis_is_scipy_lfilter():
    """Audio.py:12-13 calls scipy.signal.lfilter([1, -0.97], [1], x); scipy IS installed, so this one is the reference's own call."""
    from scipy import signal
    from multi_speaker_tts_amd import Audio
    x = _rng(1).normal(size=4001)
    assert np.allclose(OA.preemphasis(x), signal.lfilter([1, -0.97], [1], x), rtol=0, atol=1e-14)
    y = signal.lfilter([1, -0.97], [1], x)
    assert np.allclose(Audio.inv_preemphasis(y), x, rtol=0, atol=1e-9)           # Audio.py:15-16 undoes it


def test_window_is_scipy_get_window():
    """librosa.stft builds its window with scipy.signal.get_window('hann', win_length, fftbins=True) and pad_center's it to n_fft."""
    from scipy import signal
    from multi_speaker_tts_amd import Audio
    for n in (800, 400, 50):
        assert np.allclose(OA.hann_periodic(n), signal.get_window("hann", n, fftbins=True), rtol=0, atol=1e-15)
    ref = np.zeros(2048)
    ref[624:624 + 800] = signal.get_window("hann", 800, fftbins=True)
    assert np.allclose(OA.padded_window(2048, 800), ref, rtol=0, atol=1e-15)
    assert np.allclose(Audio._padded_window(2048, 800), ref, rtol=0, atol=1e-15)


@pytest.mark.parametrize("n_fft,hop,win,n", [(2048, 200, 800, 16000), (512, 100, 400, 3333), (256, 64, 256, 1000)])
def test_stft_matches_torch_stft(n_fft, hop, win, n):
    """torch.stft(center=True, pad_mode='reflect', win_length < n_fft) pads the window on both sides to n_fft and frames the
    reflect-padded signal every `hop` samples - the librosa definition Audio.py:62-64 relies on."""
    from multi_speaker_tts_amd import Audio
    y = _rng(2).normal(size=n)
    got = OA.stft(y, n_fft, hop, win)
    ref = torch.stft(torch.tensor(y), n_fft, hop_length=hop, win_length=win, window=torch.tensor(OA.hann_periodic(win)), center=True,
                     pad_mode="reflect", normalized=False, onesided=True, return_complex=True).numpy()
    assert got.shape == ref.shape == (1 + n_fft // 2, 1 + n // hop)
    assert np.abs(got - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())
    if (n_fft, hop, win) == (2048, 200, 800):                                     # the host-side export path uses the same transform
        host = Audio._stft(y, 1025, 12.5, 50, 16000)
        assert np.abs(host - ref).max() < 1e-9 * np.abs(ref).max()


def test_istft_matches_scipy_and_torch():
    """librosa.istft = windowed overlap-add divided by the summed squared window, centre padding removed.  torch.istft states the same
    definition; scipy.signal.istft the same up to its window-sum scaling (undone here)."""
    from scipy import signal
    from multi_speaker_tts_amd import Audio
    n_fft, hop, win = 2048, 200, 800
    y = _rng(3).normal(size=8000)
    D = Audio._stft(y, 1025, 12.5, 50, 16000)
    host = Audio._istft(D, 1025, 12.5, 50, 16000)
    w = OA.hann_periodic(win)
    ref_t = torch.istft(torch.tensor(D), n_fft, hop_length=hop, win_length=win, window=torch.tensor(w), center=True).numpy()
    assert host.shape == ref_t.shape
    assert np.abs(host - ref_t).max() < 1e-9
    assert np.abs(host - y).max() < 1e-9                                          # and it inverts the forward transform
    wp = OA.padded_window(n_fft, win)
    _, ref_s = signal.istft(D / wp.sum(), window=wp, nperseg=n_fft, noverlap=n_fft - hop, nfft=n_fft, boundary=True, input_onesided=True)
    m = min(len(ref_s), len(host))
    assert np.abs(host[:m] - ref_s[:m]).max() < 1e-9


@pytest.mark.parametrize("sr,n_fft,n_mels", [(16000, 2048, 80), (22050, 1024, 40)])
def test_mel_basis_matches_transformers_slaney(sr, n_fft, n_mels):
    """librosa.filters.mel(sr, n_fft, n_mels) with htk=False, norm=1 (Audio.py:82-84) = Slaney scale + Slaney area normalisation;
    transformers.audio_utils.mel_filter_bank is an independent NumPy implementation of that definition."""
    from transformers.audio_utils import mel_filter_bank
    from multi_speaker_tts_amd import Audio
    ref = mel_filter_bank(num_frequency_bins=1 + n_fft // 2, num_mel_filters=n_mels, min_frequency=0.0, max_frequency=sr / 2.0,
                          sampling_rate=sr, norm="slaney", mel_scale="slaney").T
    got = OA.mel_basis(sr, n_fft, n_mels)
    assert got.shape == ref.shape == (n_mels, 1 + n_fft // 2)
    assert np.abs(got - ref).max() < 1e-9 * ref.max()
    host = np.asarray(Audio.mel_filterbank(sr, n_fft, n_mels), np.float64)
    assert np.abs(host - ref).max() < 1e-6 * ref.max()


def test_melspectrogram_against_third_party_pipeline():
    """Audio.py:29-32 end to end out of third-party parts only: scipy lfilter -> torch.stft -> transformers mel bank -> dB -> [-4, 4]."""
    from scipy import signal
    from transformers.audio_utils import mel_filter_bank
    y = 0.3 * _rng(4).normal(size=12000)
    pre = signal.lfilter([1, -0.97], [1], y)
    D = torch.stft(torch.tensor(pre), 2048, hop_length=200, win_length=800, window=torch.hann_window(800, periodic=True, dtype=torch.float64),
                   center=True, pad_mode="reflect", return_complex=True).abs().numpy()
    fb = mel_filter_bank(1025, 80, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney").T
    S = 20 * np.log10(np.maximum(1e-5, fb @ D))
    ref = np.clip(8 * ((S + 100) / 100) - 4, -4, 4)
    got = OA.melspectrogram(y)
    assert got.shape == ref.shape == (80, 61)
    assert np.abs(got - ref).max() < 1e-8


# ------------------------------------------------------------------------------------------------------------ layers (a3, a4, a8, a13)
@pytest.mark.parametrize("K", [1, 2, 3, 5, 8, 16, 31])
def test_conv1d_same_is_torch_same(K):
    """tf.layers.conv1d(padding='same', strides=1): total padding K-1, left (K-1)//2 - torch's padding='same' states the same rule
    (extra sample on the right for even K), with the kernel transposed from TF's [K, Cin, Cout] to torch's [Cout, Cin, K]."""
    g = _rng(K)
    x = torch.tensor(g.normal(size=(2, 11, 3)))
    k = torch.tensor(g.normal(size=(K, 3, 4)))
    b = torch.tensor(g.normal(size=4))
    ref = F.conv1d(x.transpose(1, 2), k.permute(2, 1, 0), b, padding="same").transpose(1, 2)
    assert torch.allclose(OM.conv1d_same(x, k, b), ref, rtol=0, atol=1e-12)
    assert np.allclose(ON.conv1d_same(x.numpy(), k.numpy(), b.numpy()), ref.numpy(), rtol=0, atol=1e-12)


def test_batch_norm_training_against_torch():
    """tf.layers.batch_normalization(training=True, momentum=.99, epsilon=1e-3): output = torch's (both normalise with the biased batch
    variance); moving mean = torch's with momentum 0.01; moving variance: TF feeds the BIASED variance into the average, torch the
    unbiased one - the documented difference (quirk Q11), checked from both sides."""
    g = _rng(5)
    B, T, C = 3, 7, 5
    x = torch.tensor(g.normal(2.0, 3.0, size=(B, T, C)))
    p = {"bn/gamma": torch.tensor(g.normal(size=C)), "bn/beta": torch.tensor(g.normal(size=C)),
         "bn/moving_mean": torch.tensor(g.normal(size=C)), "bn/moving_variance": torch.tensor(g.uniform(0.5, 2, size=C))}
    rm, rv = p["bn/moving_mean"].clone(), p["bn/moving_variance"].clone()
    ref = F.batch_norm(x.reshape(-1, C), rm, rv, p["bn/gamma"], p["bn/beta"], training=True, momentum=0.01, eps=1e-3).reshape(B, T, C)
    stats = {}
    got = OM.batch_norm(x, p, "bn/", True, stats)
    assert torch.allclose(got, ref, rtol=0, atol=1e-12)
    assert torch.allclose(stats["bn/moving_mean"], rm, rtol=0, atol=1e-12)
    n = B * T
    biased = x.reshape(-1, C).var(dim=0, unbiased=False)
    assert torch.allclose(stats["bn/moving_variance"], 0.99 * p["bn/moving_variance"] + 0.01 * biased, rtol=0, atol=1e-12)
    assert torch.allclose(rv, 0.99 * p["bn/moving_variance"] + 0.01 * biased * n / (n - 1), rtol=0, atol=1e-12)
    # inference: both use the moving statistics
    ref_i = F.batch_norm(x.reshape(-1, C), p["bn/moving_mean"], p["bn/moving_variance"], p["bn/gamma"], p["bn/beta"], training=False, eps=1e-3)
    assert torch.allclose(OM.batch_norm(x, p, "bn/", False), ref_i.reshape(B, T, C), rtol=0, atol=1e-12)
    y_np, mean_np, var_np = ON.batch_norm_train(x.numpy(), p["bn/gamma"].numpy(), p["bn/beta"].numpy())
    assert np.allclose(y_np, ref.numpy(), rtol=0, atol=1e-12) and np.allclose(var_np, biased.numpy(), rtol=0, atol=1e-12)


def _torch_lstm_from_tf(kernel, bias, n_in, H, bidirectional=False, kernel_b=None, bias_b=None):
    """A torch.nn.LSTM holding a TF LSTM kernel [n_in + H, 4H] (gate order i, j, f, o; forget_bias 1.0 added at run time, ZoneoutLSTMCell.py:
    239) - torch's order is i, f, g, o and its forget bias is part of the parameter."""
    lstm = torch.nn.LSTM(n_in, H, batch_first=True, bidirectional=bidirectional).double()

    def put(sfx, kern, bia):
        i, j, f, o = kern.chunk(4, dim=1)
        w = torch.cat([i, f, j, o], dim=1)                                   # -> torch gate order
        bi, bj, bf_, bo = bia.chunk(4)
        with torch.no_grad():
            getattr(lstm, "weight_ih_l0" + sfx).copy_(w[:n_in].t())
            getattr(lstm, "weight_hh_l0" + sfx).copy_(w[n_in:].t())
            getattr(lstm, "bias_ih_l0" + sfx).copy_(torch.cat([bi, bf_ + 1.0, bj, bo]))
            getattr(lstm, "bias_hh_l0" + sfx).zero_()
    put("", kernel, bias)
    if bidirectional:
        put("_reverse", kernel_b, bias_b)
    return lstm


def test_cell_without_zoneout_is_torch_lstm_cell():
    """ZoneoutLSTMCell with rate 0 is the plain LSTM of tf.nn.rnn_cell.LSTMCell(forget_bias=1.0): one step against torch.nn.LSTM."""
    g = _rng(6)
    B, n_in, H = 4, 5, 6
    k = torch.tensor(g.normal(0, 0.4, size=(n_in + H, 4 * H)))
    b = torch.tensor(g.normal(0, 0.4, size=4 * H))
    x, c0, h0 = (torch.tensor(g.normal(size=s)) for s in ((B, n_in), (B, H), (B, H)))
    m, c1, h1 = OM.zoneout_lstm_cell(x, c0, h0, k, b, None, None, 0.0, False)
    lstm = _torch_lstm_from_tf(k, b, n_in, H)
    out, (hn, cn) = lstm(x[:, None, :], (h0[None], c0[None]))
    assert torch.allclose(m, out[:, 0], rtol=0, atol=1e-12) and torch.allclose(h1, hn[0], rtol=0, atol=1e-12)
    assert torch.allclose(c1, cn[0], rtol=0, atol=1e-12)
    m2, c2, h2 = ON.zoneout_lstm_cell(x.numpy(), c0.numpy(), h0.numpy(), k.numpy(), b.numpy(), rate=0.0)
    assert np.allclose(m2, out[:, 0].detach().numpy(), rtol=0, atol=1e-12) and np.allclose(c2, cn[0].detach().numpy(), rtol=0, atol=1e-12)


def test_dynamic_rnn_lengths_against_packed_bilstm():
    """tf.nn.bidirectional_dynamic_rnn with sequence_length: outputs past a row's length are zero, the backward direction runs over
    tf.reverse_sequence(x, length) and is reversed back the same way.  torch's packed bidirectional LSTM has exactly these semantics
    (pad_packed_sequence pads with zeros; the reverse direction starts at each row's own last valid step)."""
    g = _rng(7)
    B, T, n_in, H = 4, 9, 3, 5
    lengths = torch.tensor([9, 4, 1, 6])
    x = torch.tensor(g.normal(size=(B, T, n_in)))
    kf, kb = (torch.tensor(g.normal(0, 0.5, size=(n_in + H, 4 * H))) for _ in range(2))
    bf, bb = (torch.tensor(g.normal(0, 0.5, size=4 * H)) for _ in range(2))
    fw = OM.run_lstm(x, lengths, kf, bf, H, None, None, 0.0, False, reverse=False)
    bw = OM.run_lstm(x, lengths, kb, bb, H, None, None, 0.0, False, reverse=True)
    lstm = _torch_lstm_from_tf(kf, bf, n_in, H, bidirectional=True, kernel_b=kb, bias_b=bb)
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, lengths, batch_first=True, enforce_sorted=False)
    ref, _ = torch.nn.utils.rnn.pad_packed_sequence(lstm(packed)[0], batch_first=True, total_length=T)
    assert torch.allclose(fw, ref[:, :, :H], rtol=0, atol=1e-12)
    assert torch.allclose(bw, ref[:, :, H:], rtol=0, atol=1e-12)
    assert float(fw[1, 4:].abs().max()) == 0.0 and float(bw[2, 1:].abs().max()) == 0.0
    # the NumPy restatement's dynamic_rnn, same check
    ones = np.ones((T, B, H))
    fw2 = ON._dynamic_rnn(x.numpy(), lengths.numpy(), kf.numpy(), bf.numpy(), H, ones, ones, 0.0, False)
    bw2 = ON._dynamic_rnn(x.numpy(), lengths.numpy(), kb.numpy(), bb.numpy(), H, ones, ones, 0.0, True)
    assert np.allclose(fw2, ref[:, :, :H].detach().numpy(), rtol=0, atol=1e-12) and np.allclose(bw2, ref[:, :, H:].detach().numpy(), rtol=0, atol=1e-12)


# ----------------------------------------------------------------------------------------------------------------- loss, optimizer (a14)
def test_losses_against_torch_functionals():
    """tf.losses.mean_squared_error / absolute_difference / sigmoid_cross_entropy with default weights = means over all elements."""
    g = _rng(8)
    B, L, n_mel = 3, 6, 4
    mel = torch.tensor(g.normal(size=(B, L, n_mel)))
    out = {"Linear": torch.tensor(g.normal(size=(B, L + 1, n_mel))), "Mel": torch.tensor(g.normal(size=(B, L + 1, n_mel))),
           "Stop_Logit": torch.tensor(g.normal(0, 3, size=(B, L + 1)))}
    lengths = torch.tensor([6, 2, 4], dtype=torch.int32)
    batch = {"Mel": mel, "Mel_Length": lengths}
    got = OT.losses({}, out, batch, wr_rate=0.0, use_l1=True)
    lin, post = out["Linear"][:, :-1], out["Mel"][:, :-1]
    assert torch.allclose(got["Linear_Loss"], F.mse_loss(lin, mel) + F.l1_loss(lin, mel), rtol=0, atol=1e-13)
    assert torch.allclose(got["Postnet_Loss"], F.mse_loss(post, mel) + F.l1_loss(post, mel), rtol=0, atol=1e-13)
    target = (torch.arange(L + 1)[None, :] >= lengths[:, None]).double()
    assert torch.allclose(got["Stop_Loss"], F.binary_cross_entropy_with_logits(out["Stop_Logit"], target), rtol=0, atol=1e-13)
    assert torch.allclose(OT.losses({}, out, batch, wr_rate=0.0, use_l1=False)["Linear_Loss"], F.mse_loss(lin, mel), rtol=0, atol=1e-13)


@pytest.mark.parametrize("t", [1, 2, 7])
def test_tf_adam_against_torch_adam(t):
    """tf.train.AdamOptimizer: theta -= lr sqrt(1-b2^t)/(1-b1^t) m / (sqrt(v) + eps)   ("epsilon hat" of the paper, section 2).
    torch.optim.Adam:          theta -= lr/(1-b1^t) m / (sqrt(v)/sqrt(1-b2^t) + eps).
    They coincide when torch is given eps / sqrt(1-b2^t) at step t - checked over t steps with the epsilon re-set per step -
    and differ by a known factor otherwise (the oracle must sit on the TF side: epsilon outside the bias correction, quirk Q18)."""
    g = _rng(9)
    b1, b2, eps, lr = 0.9, 0.999, 1e-6, 1e-3
    p0 = g.normal(size=17)
    grads = [g.normal(size=17) * 10.0 ** g.integers(-7, 1) for _ in range(t)]       # small gradients make epsilon matter
    tp = torch.nn.Parameter(torch.tensor(p0))
    opt = torch.optim.Adam([tp], lr=lr, betas=(b1, b2), eps=eps)
    p, m, v = torch.tensor(p0), torch.zeros(17, dtype=torch.float64), torch.zeros(17, dtype=torch.float64)
    pn, mn, vn = p0.copy(), np.zeros(17), np.zeros(17)
    for s in range(1, t + 1):
        opt.param_groups[0]["eps"] = eps / np.sqrt(1 - b2 ** s)
        tp.grad = torch.tensor(grads[s - 1])
        opt.step()
        p, m, v = OT.adam_tf(p, torch.tensor(grads[s - 1]), m, v, s, lr, b1, b2, eps)
        pn, mn, vn = ON.tf_adam(pn, grads[s - 1], mn, vn, s - 1, lr, b1, b2, eps)
    assert torch.allclose(p, tp.detach(), rtol=0, atol=1e-15)
    assert np.allclose(pn, tp.detach().numpy(), rtol=0, atol=1e-15)
    # with the SAME epsilon the two definitions differ measurably at step 1 for tiny gradients: the oracle is not torch's Adam
    tq = torch.nn.Parameter(torch.tensor(p0))
    o2 = torch.optim.Adam([tq], lr=lr, betas=(b1, b2), eps=eps)
    tq.grad = torch.full((17,), 1e-7, dtype=torch.float64)
    o2.step()
    q, _, _ = OT.adam_tf(torch.tensor(p0), torch.full((17,), 1e-7, dtype=torch.float64), torch.zeros(17, dtype=torch.float64),
                         torch.zeros(17, dtype=torch.float64), 1, lr, b1, b2, eps)
    d_tf, d_torch = (torch.tensor(p0) - q).abs().max(), (torch.tensor(p0) - tq.detach()).abs().max()
    assert float(d_tf) < 0.8 * float(d_torch)


def test_exponential_decay_against_torch_scheduler():
    """tf.train.exponential_decay(lr0, step, decay_steps, rate, staircase=False) = lr0 rate^(step/decay_steps), floored at 1e-5
    (MSTTS_SV.py:163-172).  torch's ExponentialLR(gamma) multiplies by gamma per step: gamma = rate^(1/decay_steps)."""
    tp = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([tp], lr=1e-3)
    sch = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.5 ** (1.0 / 100))
    for step in range(0, 1000):
        want = max(opt.param_groups[0]["lr"], 1e-5)
        assert abs(OT.learning_rate(step, decay_step=100) - want) < 1e-12
        assert abs(ON.tf_learning_rate(step, decay_step=100) - want) < 1e-12
        opt.step()
        sch.step()


# ------------------------------------------------------------------------------------------------------------- checkpoint format (f.3)
def test_snappy_decoder_against_pyarrow_snappy():
    """TF table blocks may be snappy-compressed (block trailer type 1); tf_checkpoint restates the decoder.  pyarrow ships the real
    snappy codec: whatever it compresses - literals, short and long copies, overlapping copies, >64 KB inputs - must come back."""
    import pyarrow as pa
    from multi_speaker_tts_amd import tf_checkpoint as TC
    if not pa.Codec.is_available("snappy"):
        pytest.skip("pyarrow built without snappy")
    codec = pa.Codec("snappy")
    g = _rng(10)
    cases = [b"", b"a", b"abc" * 1000, bytes(g.integers(0, 256, size=5000, dtype=np.uint8)), b"\x00" * 70000,
             (b"tensor_names/decoder/kernel" * 37 + bytes(g.integers(0, 4, size=3000, dtype=np.uint8))) * 30,
             np.arange(40000, dtype=np.int32).tobytes()]
    for raw in cases:
        assert bytes(TC._snappy_decompress(codec.compress(raw, asbytes=True))) == raw
