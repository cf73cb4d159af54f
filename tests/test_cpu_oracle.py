"""CPU checks that pin the oracle as far as it can be pinned here: the torch restatement against the
independent NumPy restatement, autograd against finite differences, TF-Adam / loss formulas against
hand arithmetic, and the quirk list as executable assertions."""
import numpy as np
import torch

from oracle import model as OM, train as OT, np_ops as ON, audio as OA
from tests.helpers import small_dims


def _rng(seed=0):
    return np.random.default_rng(seed)


def test_cell_torch_vs_numpy():
    g = _rng(1)
    B, In, H = 3, 5, 4
    x, c, h = g.normal(size=(B, In)), g.normal(size=(B, H)), g.normal(size=(B, H))
    k, b = g.normal(size=(In + H, 4 * H)), g.normal(size=4 * H)
    zc, zh = g.integers(0, 2, (B, H)), g.integers(0, 2, (B, H))
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    for training in (True, False):
        m, c2, h2 = OM.zoneout_lstm_cell(t(x), t(c), t(h), t(k), t(b), t(zc), t(zh), 0.1, training)
        mn, cn, hn = ON.zoneout_lstm_cell(x, c, h, k, b, zc if training else None, zh if training else None)
        assert np.allclose(m.numpy(), mn) and np.allclose(c2.numpy(), cn) and np.allclose(h2.numpy(), hn)
    # quirk Q2: inference zoneout is 0.9*new + 0.1*old ; cell output is the un-zoned m
    assert np.allclose(cn, 0.9 * (cn - 0.1 * c) / 0.9 + 0.1 * c)
    m, c2, h2 = ON.zoneout_lstm_cell(x, c, h, k, b)
    assert np.allclose(h2, 0.9 * m + 0.1 * h)


def test_conv_same_padding_even_kernels():
    g = _rng(2)
    x = g.normal(size=(2, 9, 3))
    for K in range(1, 9):
        k = g.normal(size=(K, 3, 4)); b = g.normal(size=4)
        y1 = OM.conv1d_same(torch.tensor(x), torch.tensor(k), torch.tensor(b)).numpy()
        assert np.allclose(y1, ON.conv1d_same(x, k, b))
    # quirk Q12: K=2 pads 0 left / 1 right -> y[t] = x[t] k0 + x[t+1] k1
    k = g.normal(size=(2, 3, 4))
    y = ON.conv1d_same(x, k)
    assert np.allclose(y[:, 0], x[:, 0] @ k[0] + x[:, 1] @ k[1]) and np.allclose(y[:, -1], x[:, -1] @ k[0])


def test_lsa_step_torch_vs_numpy():
    g = _rng(3)
    B, T, A, M, Hq, KS, CH = 2, 11, 6, 5, 7, 5, 3
    d = OM.Dims(att=A, att_k=KS, att_ch=CH, dec_lstm=Hq)
    P = OM.P_LSA
    p = {P + "query_layer/kernel": g.normal(size=(Hq, A)), P + "attention_convolution_dense_layer/conv1d/kernel": g.normal(size=(KS, 1, CH)),
         P + "attention_convolution_dense_layer/conv1d/bias": g.normal(size=CH), P + "attention_convolution_dense_layer/dense/kernel": g.normal(size=(CH, A)),
         P + "score_layer/weight_w": g.normal(size=(1, 1, A)), P + "score_layer/bias_b": g.normal(size=(1, 1, A))}
    keys, values = g.normal(size=(B, T, A)), g.normal(size=(B, T, M))
    lengths = np.array([T, 7]); query = g.normal(size=(B, Hq)); cum = np.abs(g.normal(size=(B, T)))
    mask = np.arange(T)[None] < lengths[:, None]
    a, cn, ctx = OM.lsa_step({k: torch.tensor(v) for k, v in p.items()}, d, torch.tensor(keys), torch.tensor(values),
                             torch.tensor(mask), torch.tensor(query), torch.tensor(cum))
    a2, cn2, ctx2 = ON.lsa_step(keys, values, lengths, query, cum, p[P + "query_layer/kernel"],
                                p[P + "attention_convolution_dense_layer/conv1d/kernel"], p[P + "attention_convolution_dense_layer/conv1d/bias"],
                                p[P + "attention_convolution_dense_layer/dense/kernel"], p[P + "score_layer/weight_w"], p[P + "score_layer/bias_b"])
    assert np.allclose(a.numpy(), a2) and np.allclose(cn.numpy(), cn2) and np.allclose(ctx.numpy(), ctx2)
    assert np.allclose(a2.sum(1), 1.0) and not a2[1, 7:].any()          # quirk Q6: masked softmax


def test_decoder_quirks():
    d = OM.Dims(**small_dims())
    p = OM.to_torch(OM.init_params(d, 3))
    batch = OT.synthetic_batch(d, 2, 6, 4, seed=1, ragged=True)
    masks = OT.make_masks(d, 2, 6, 5, True, seed=7)
    bt = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    out = OM.forward(p, d, bt, True, masks, with_vocoder=False)
    assert out["Linear"].shape == (2, 5, d.n_mel)                       # Q7: max(L)+1 steps
    assert out["Attention_History"].shape == (2, 6, 5)
    k0 = p[OM.P_CELL % 0 + "kernel"]
    assert k0.shape[0] == d.prenet + 2 * d.mem + d.dec_lstm             # Q1: context enters cell 0 twice
    # Q9: prenet dropout is active at inference too -> different masks give different inference outputs
    m1 = OT.make_masks(d, 2, 6, d.max_inf + 1, False, seed=1)
    m2 = OT.make_masks(d, 2, 6, d.max_inf + 1, False, seed=2)
    d2 = OM.Dims(**small_dims(max_inf=6))
    o1 = OM.forward(p, d2, bt, False, m1, with_vocoder=False)["Linear"]
    o2 = OM.forward(p, d2, bt, False, m2, with_vocoder=False)["Linear"]
    assert o1.shape[1] <= 7 and not torch.allclose(o1[:, :min(o1.shape[1], o2.shape[1])], o2[:, :min(o1.shape[1], o2.shape[1])])


def test_loss_and_adam_formulas():
    d = OM.Dims(**small_dims())
    p = OM.to_torch(OM.init_params(d, 4))
    B, L = 2, 3
    mel = torch.randn(B, L, d.n_mel, dtype=torch.float64)
    out = {"Linear": torch.randn(B, L + 1, d.n_mel, dtype=torch.float64), "Mel": torch.randn(B, L + 1, d.n_mel, dtype=torch.float64),
           "Stop_Logit": torch.randn(B, L + 1, dtype=torch.float64)}
    ls = OT.losses(p, out, {"Mel": mel, "Mel_Length": torch.tensor([3, 2])})
    e = (out["Linear"][:, :-1] - mel).numpy()
    assert abs(float(ls["Linear_Loss"]) - ((e ** 2).mean() + np.abs(e).mean())) < 1e-12          # Q8: plain means, no mask
    tgt = np.array([[0, 0, 0, 1], [0, 0, 1, 1]], float)                                           # Q7: stop target 1 from t=L on
    z = out["Stop_Logit"].numpy()
    assert abs(float(ls["Stop_Loss"]) - (np.maximum(z, 0) - z * tgt + np.log1p(np.exp(-np.abs(z)))).mean()) < 1e-12
    wr = sum(float((v ** 2).sum()) / 2 for k, v in p.items() if OM.in_weight_reg(k)) * 1e-6
    assert abs(float(ls["Weight_Regularization_Loss"]) - wr) < 1e-15
    assert OM.in_weight_reg("encoder/conv_0/batch_normalization/beta") and not OM.in_weight_reg("encoder/conv_0/conv1d/bias")   # Q19
    assert not OM.in_weight_reg(OM.P_CELL % 0 + "kernel") and not OM.in_weight_reg("decoder/decoder/linear_projection/dense/kernel")
    # Q18: TF Adam, epsilon outside the bias correction
    t64 = lambda x: torch.tensor([x], dtype=torch.float64)
    pn, m, v = OT.adam_tf(t64(1.0), t64(0.5), t64(0.0), t64(0.0), 1, 1e-3)
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert abs(float(pn) - (1.0 - lr_t * 0.05 / (np.sqrt(0.00025) + 1e-6))) < 1e-12


def test_autograd_matches_finite_differences():
    d = OM.Dims(**small_dims(dec_lstm=8, enc_lstm=4, spk=4, prenet=4, emb=8, enc_conv_ch=8, post_ch=8, n_mel=4, att=8, att_ch=4, att_k=5))
    params = OM.init_params(d, 5)
    batch = OT.synthetic_batch(d, 2, 5, 3, seed=2, ragged=True)
    masks = OT.make_masks(d, 2, 5, 4, True, seed=3)
    _, _, sc, grads, _ = OT.train_step(params, None, d, batch, masks, 0, return_grads=True, update_vocoder_bn=False)
    g = _rng(6)
    for name in (OM.P_CELL % 1 + "kernel", OM.P_LSA + "attention_convolution_dense_layer/conv1d/kernel", "encoder/conv_1/conv1d/kernel",
                 "decoder/conv_2/batch_normalization/gamma", "attention/memory_layer/kernel"):
        idx = tuple(g.integers(0, s) for s in params[name].shape)
        eps = 1e-6
        vals = []
        for sgn in (+1, -1):
            q = {k: v.copy() for k, v in params.items()}
            q[name][idx] += sgn * eps
            _, _, s2 = OT.train_step(q, None, d, batch, masks, 0, update_vocoder_bn=False)
            vals.append(s2["Loss"])
        fd = (vals[0] - vals[1]) / (2 * eps)
        an = float(grads[name][idx])
        assert abs(fd - an) < 1e-5 * max(1.0, abs(an)), (name, fd, an)


def test_speaker_encoder_whole_tensor_norm():
    d = OM.Dims(**small_dims())
    p = OM.to_torch(OM.init_params(d, 6))
    x = torch.randn(3 * d.spk_samples, d.spk_frames, d.n_mel, dtype=torch.float64)
    e = OM.speaker_encoder(p, d, x)
    assert e.shape == (3, d.spk) and abs(float((e ** 2).sum()) - 1.0) < 1e-9        # Q14: whole-tensor l2 norm


def test_waveglow_flow_is_invertible():
    """Pins oracle/waveglow.py against itself: Glow_Train direction followed by Glow_Inference on the emitted latents
    returns the audio (WaveGlow/Modules.py:329-371), for a configuration with two early outputs."""
    from oracle import waveglow as OW
    d = OW.WGDims(n_mel=8, flows=6, groups=8, early_every=2, early_size=2, up_k=16, up_stride=4, layers=3, ch=16, k=3)
    p = OW.to_torch(OW.init_params(d, seed=1))
    g = np.random.default_rng(2)
    N, T = 2, 5
    mel = torch.tensor(g.normal(0, 1, (N, T, d.n_mel)))
    L = (T - 1) * d.up_stride + d.up_k
    audio = torch.tensor(g.normal(0, 0.3, (N, L // d.groups, d.groups)))
    melg = OW.restructure_inference_mel(p, d, mel)
    assert melg.shape == (N, L // d.groups, d.groups * d.n_mel)
    noise = OW.glow_forward(p, d, audio, melg)
    assert noise["z"].shape[-1] == d.z_channels == 4 and set(noise) == {"z", "early_2", "early_4"}
    back = OW.glow_inference(p, d, mel, noise, sigma=1.0)
    assert float((back - audio.reshape(N, -1)).abs().max()) < 1e-9
    # weight norm: unit-norm direction per output channel scaled by g (WaveGlow/Modules.py:32-34)
    w = OW.weight_norm(torch.tensor([2.0, 3.0]), torch.tensor(g.normal(0, 1, (1, 3, 4, 2))))
    assert np.allclose(np.sqrt((w ** 2).sum(dim=(0, 1, 2)).numpy()), [2.0, 3.0])
    # upsampler length and channel bookkeeping of the reference hyper-parameters
    full = OW.WGDims()
    assert full.z_channels == 4 and [full.channels(f) for f in (0, 3, 4, 7, 8, 11)] == [8, 8, 6, 6, 4, 4]
    assert (40 - 1) * full.up_stride + full.up_k == 11008 and 11008 % full.groups == 0


def test_waveglow_chunking_kats():
    """MSTTS_SV.Inference_WaveGlow host arithmetic (MSTTS_SV.py:335-347,452-458) - product and oracle agree with hand values."""
    from oracle import waveglow as OW
    from multi_speaker_tts_amd import waveglow as WG
    mels = [np.zeros((95, 80)), np.zeros((40, 80)), np.zeros((3, 80))]
    for mod in (OW, WG):
        chunks, index = mod.split_mels(mels, 40)
        assert [c.shape[0] for c in chunks] == [40, 40, 15, 40, 3] and index == [(0, 3), (3, 4), (4, 5)]
        assert mod.export_length([.1, .2, .6, .9], 12.5, 22050) == int(2 * 12.5 / 1000 * 22050) == 551
        assert mod.export_length([.1, .2, .3], 12.5, 22050) == int(3 * 12.5 / 1000 * 22050)


def test_second_restatement_of_the_train_step():
    """oracle/np_ops.py restates the whole TRAIN forward (encoder incl. length-reversed BiLSTM, doubled context Q1, max(L)+1 steps
    and teacher-forcing shift Q7, postnet), the losses (Q8, Q19) and TF-Adam (Q18) independently of oracle/model.py / train.py in
    NumPy fp64: both must agree to 1e-9 on ragged inputs, and so must the optimizer step built on model.py's gradients."""
    from oracle import np_ops as NP
    cfg = dict(emb=12, enc_conv_ch=12, enc_lstm=6, spk=8, prenet=7, dec_lstm=10, n_mel=5, post_ch=9, bank_ch=4, proj1_ch=6, birnn=4, n_spec=9, spk_lstm=8)
    d = OM.Dims(**cfg)
    g = np.random.default_rng(17)
    params = OM.init_params(d, 17)
    for k in params:                                     # nothing left at its trivial initial value
        if k.endswith(("bias", "beta", "bias_b", "moving_mean")):
            params[k] = g.normal(0, 0.2, params[k].shape)
        if k.endswith(("gamma", "moving_variance")):
            params[k] = 1.0 + 0.3 * g.random(params[k].shape)
    B, Te, L = 3, 9, 6
    batch = OT.synthetic_batch(d, B, Te, L, seed=17, ragged=True)
    assert len(set(batch["Token_Length"].tolist())) > 1 and len(set(batch["Mel_Length"].tolist())) > 1
    masks = OT.make_masks(d, B, Te, L + 1, True, seed=5)
    new_p, opt, sc, grads, out = OT.train_step(params, None, d, batch, masks, 0, return_grads=True)
    npb = {k: v.numpy() for k, v in batch.items()}
    npm = {k: v.numpy().astype(np.float64) for k, v in masks.items()}
    npp = {k: np.asarray(v, np.float64) for k, v in params.items()}
    mine = NP.tacotron_train_forward(npp, d, npb, npm)
    for key in ("Linear", "Mel", "Stop_Logit", "Attention_History"):
        assert mine[key].shape == tuple(out[key].shape) and np.abs(mine[key] - out[key].numpy()).max() < 1e-9, key
    assert mine["Linear"].shape[1] == int(batch["Mel_Length"].max()) + 1                          # Q7
    ls = NP.tacotron_losses(npp, mine, npb)
    for key in ("Loss", "Linear_Loss", "Postnet_Loss", "Stop_Loss", "Weight_Regularization_Loss"):
        assert abs(ls[key] - sc[key]) < 1e-10, key
    assert all(NP.weight_regularised(k) == OM.in_weight_reg(k) for k in params)
    for k, v in mine["stats"].items():                                                             # BN moving statistics (Q11)
        assert np.abs(v - new_p[k].numpy()).max() < 1e-10, k
    # TF-Adam on model.py's gradients, two consecutive steps of one variable with non-zero slots
    k = "decoder/decoder/prenet_1/dense/kernel"
    gr = grads[k].numpy()
    p1, m1, v1 = NP.tf_adam(npp[k], gr, np.zeros_like(gr), np.zeros_like(gr), 0, NP.tf_learning_rate(0))
    assert np.abs(p1 - new_p[k].numpy()).max() < 1e-12 and np.abs(m1 - opt["m"][k].numpy()).max() < 1e-15
    p2a, m2a, v2a = NP.tf_adam(p1, 0.5 * gr, m1, v1, 1, NP.tf_learning_rate(1))
    p2b, m2b, v2b = OT.adam_tf(torch.tensor(p1), torch.tensor(0.5 * gr), torch.tensor(m1), torch.tensor(v1), 2, OT.learning_rate(1))
    assert np.abs(p2a - p2b.numpy()).max() < 1e-12 and np.abs(v2a - v2b.numpy()).max() < 1e-18
    assert NP.tf_learning_rate(10000) == OT.learning_rate(10000) == 5e-4 and NP.tf_learning_rate(10 ** 7) == 1e-5


def test_bf16_emulation_sensitivity():
    """Why whole-step parity bounds are loose in config-3 arithmetic: the oracle's bf16 emulation is discontinuous (rounding to 8
    mantissa bits), so a 1e-6 relative perturbation of the variables moves its own outputs by ~1e-2 and its gradients by a few per
    cent (relative L2), while the exact-arithmetic oracle moves by ~1e-5.  tests/test_gpu_model.py::test_train_step_parity_bf16_full
    compares at this noise level; the bf16 kernels themselves are pinned to 2e-5 by op tests."""
    def l2(a, b):
        return float(np.sqrt(((a - b) ** 2).sum() / ((b ** 2).sum() + 1e-30)))
    cfg = dict(emb=32, enc_conv_ch=32, enc_lstm=32, spk=64, prenet=32, dec_lstm=64, n_mel=8, post_ch=16, bank_ch=8, proj1_ch=16, birnn=8, n_spec=20, spk_lstm=16)
    d = OM.Dims(**cfg)
    values = OM.init_params(d, 13)
    g = np.random.default_rng(14)
    batch = OT.synthetic_batch(d, 5, 18, 9, seed=13, ragged=True)
    masks = OT.make_masks(d, 5, 18, 10, True, seed=3)
    moved = {}
    try:
        for mode in (True, False):
            OM.RECURRENT_BF16 = OM.GEMM_BF16 = mode
            r0 = OT.train_step(values, None, d, batch, masks, 0, return_grads=True)
            v2 = {k: np.asarray(v) * (1 + 1e-6 * g.normal(size=np.shape(v))) for k, v in values.items()}
            r1 = OT.train_step(v2, None, d, batch, masks, 0, return_grads=True)
            moved[mode] = (l2(r1[4]["Mel"].numpy(), r0[4]["Mel"].numpy()), float(np.median([l2(r1[3][k].numpy(), r0[3][k].numpy()) for k in r0[3]])))
    finally:
        OM.RECURRENT_BF16 = OM.GEMM_BF16 = False
    assert moved[False][0] < 1e-4 and moved[False][1] < 1e-3          # exact arithmetic: smooth
    assert moved[True][0] > 1e-3 and moved[True][1] > 5e-3            # emulated bf16: the rounding flips dominate
