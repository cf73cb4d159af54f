"""Taco1 mel -> spectrogram trainer through the C ABI vs the oracle's autograd restatement."""
import numpy as np
import pytest
import torch

from multi_speaker_tts_amd import lib
from multi_speaker_tts_amd.taco1_trainer import Taco1TrainEngine
from oracle import model as OM, train as OT
from tests.helpers import dims_pair, rel_err, t2n

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def test_small_backward_kernels(dev):
    g = np.random.default_rng(0)
    B, T, C = 3, 7, 5
    x = torch.tensor(g.integers(-2, 3, (B, T, C)).astype(np.float32), device=dev)        # ties on purpose
    dy = torch.tensor(g.normal(size=(B, T, C)).astype(np.float32), device=dev)
    dx = torch.zeros_like(x)
    lib.call("mstts_maxpool2_same_bwd", lib.ptr(x), lib.ptr(dy), lib.ptr(dx), B, T, C)
    xr = x.double().cpu().requires_grad_(True)
    # first-max tie rule: emulate with a tiny decreasing bias along time
    y = torch.maximum(xr, torch.cat([xr[:, 1:] - 1e-9, torch.full_like(xr[:, :1], -float("inf"))], dim=1))
    (y * dy.double().cpu()).sum().backward()
    assert rel_err(t2n(dx), t2n(xr.grad)) < 1e-6
    n = 1000
    hp_, tp_, xx, dd = [torch.tensor(g.normal(size=n).astype(np.float32), device=dev) for _ in range(4)]
    dh, dt, dxx = torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    lib.call("mstts_highway_combine_bwd", lib.ptr(hp_), lib.ptr(tp_), lib.ptr(xx), lib.ptr(dd), lib.ptr(dh), lib.ptr(dt), lib.ptr(dxx), n)
    h64, t64, x64 = [v.double().cpu().requires_grad_(True) for v in (hp_, tp_, xx)]
    Tt = torch.sigmoid(t64)
    ((torch.relu(h64) * Tt + x64 * (1 - Tt)) * dd.double().cpu()).sum().backward()
    assert rel_err(t2n(dh), t2n(h64.grad)) < 1e-5 and rel_err(t2n(dt), t2n(t64.grad)) < 1e-5 and rel_err(t2n(dxx), t2n(x64.grad)) < 1e-5
    p, t = torch.tensor(g.normal(size=5000).astype(np.float32), device=dev), torch.tensor(g.normal(size=5000).astype(np.float32), device=dev)
    t[:7] = p[:7]                                                                     # exact zeros: sign(0) = 0
    loss, dp = torch.zeros(1, device=dev), torch.zeros(5000, device=dev)
    lib.call("mstts_l1_loss_fwd_bwd", lib.ptr(p), lib.ptr(t), 5000, lib.ptr(loss), lib.ptr(dp))
    assert abs(float(loss) - float((p - t).abs().mean())) < 1e-6 and rel_err(t2n(dp), t2n(torch.sign(p - t) / 5000)) < 1e-6


@pytest.mark.parametrize("B,S,kw", [(3, 11, {}), (4, 23, dict(bank_ch=16, proj1_ch=32, birnn=16, n_spec=40)), (2, 9, dict(n_mel=80, bank_ch=32, proj1_ch=64, birnn=32, n_spec=129))])
def test_taco1_train_step_parity(dev, B, S, kw):
    """Forward, loss, every gradient, parameters after TF-Adam and the BN moving statistics of one trainer step, then a second step."""
    pd, od = dims_pair(**kw)
    values = OM.init_params(od, 5)
    g = np.random.default_rng(6)
    for k in values:
        if k.startswith(OM.P_V) and k.endswith(("bias", "beta")) and "highway" not in k:
            values[k] = g.normal(0, 0.1, values[k].shape)
        if k.startswith(OM.P_V) and k.endswith("moving_variance"):
            values[k] = 0.5 + np.abs(g.normal(0, 0.5, values[k].shape))
    mel = np.clip(g.normal(0, 1.5, (B, S, od.n_mel)), -4, 4).astype(np.float32)
    spec = g.uniform(0, 1, (B, S, od.n_spec)).astype(np.float32)
    eng = Taco1TrainEngine(pd, device=dev, values=values)
    params, opt = values, None
    for step in range(2):
        masks = {k: torch.tensor(g.integers(0, 2, (S, B, od.birnn)).astype(np.uint8)) for k in ("v_zc_fw", "v_zh_fw", "v_zc_bw", "v_zh_bw")}
        params, opt, sc, grads, pred = OT.taco1_train_step(params, opt, od, torch.tensor(mel), torch.tensor(spec), masks, step, return_grads=True)
        w = eng.plan(B, S)
        eng.forward(torch.tensor(mel, device=dev), w, masks={k: v.numpy() for k, v in masks.items()})
        eng.loss_and_backward(w, torch.tensor(spec, device=dev))
        got = eng.scalars(w)
        assert rel_err(t2n(w.pred).reshape(B, S, -1), t2n(pred)) < 1e-3
        assert abs(got["Loss"] - sc["Loss"]) < 1e-4 * max(1.0, abs(sc["Loss"])) and abs(got["Weight_Regularization_Loss"] - sc["Weight_Regularization_Loss"]) < 1e-6
        gexp = eng.params.export(grads=True)
        def gerr(k):          # conv1d_9/bias feeds a BatchNorm directly: its true gradient is exactly zero -> absolute test
            ref = t2n(grads[k])
            return rel_err(gexp[k], ref) if np.abs(ref).max() > 1e-9 else float(np.abs(gexp[k]).max() > 1e-5)
        bad = [(k, gerr(k)) for k in grads if gerr(k) > 5e-3]
        assert not bad, bad
        eng.adam_step()
        now = eng.params.export()
        bad = [(k, rel_err(now[k], t2n(params[k]))) for k in now if k.startswith(OM.P_V) and rel_err(now[k], t2n(params[k])) > 2e-3]
        assert not bad, bad
    assert eng.global_step == 2


def test_mel_to_spect_surface(dev, tmp_path, monkeypatch):
    """Mel_to_Spect().Train_Step / Save / Restore, and the saved file is what Tacotron2.Vocoder_Load reads."""
    from multi_speaker_tts_amd import Hyper_Parameters as hp
    from multi_speaker_tts_amd.Taco1_Mel_to_Spect import Mel_to_Spect, TRAIN_KEYS
    from multi_speaker_tts_amd.params import Dims
    monkeypatch.setattr(hp.Taco1_Mel_to_Spect, "Checkpoint_Path", str(tmp_path / "voc"))
    monkeypatch.setattr(hp, "Checkpoint_Path", str(tmp_path / "ckpt"))
    dims = Dims(emb=32, enc_conv_ch=32, enc_lstm=16, spk=256, prenet=16, dec_lstm=32, post_ch=16, bank_ch=8, proj1_ch=16, birnn=8, n_spec=20,
                spk_lstm=256, max_inf=4)
    m = Mel_to_Spect(device=dev, dims=dims)
    pat = m.Synthetic_Pattern(batch_Size=4, length=30)
    losses = [m.Train_Step(pat) for _ in range(30)]
    assert set(TRAIN_KEYS) <= set(losses[0]) and losses[0]["Global_Step"] == 0 and losses[-1]["Global_Step"] == 29
    assert losses[-1]["Loss"] < losses[0]["Loss"]                    # it learns the fixed batch
    m.Save()
    saved = m.params.export()
    m2 = Mel_to_Spect(device=dev, dims=dims, seed=99)
    m2.Restore()
    assert m2.engine.global_step == 30 and all(np.array_equal(saved[k], m2.params.export()[k]) for k in saved if k.startswith("mel_to_spectrogram"))
    from multi_speaker_tts_amd.MSTTS_SV import Tacotron2
    t = Tacotron2(is_Training=False, device=dev, dims=dims, allow_random_init=True)
    mine = t.params.export()
    assert all(np.array_equal(saved[k], mine[k]) for k in saved if k.startswith("mel_to_spectrogram"))
