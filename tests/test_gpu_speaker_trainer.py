"""GE2E speaker-encoder trainer through the C ABI vs the oracle's autograd restatement."""
import numpy as np
import pytest
import torch

from multi_speaker_tts_amd import lib
from multi_speaker_tts_amd.speaker_trainer import SpeakerTrainEngine
from oracle import model as OM, train as OT
from tests.helpers import dims_pair, rel_err, t2n

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("S,P,D", [(4, 3, 16), (32, 10, 256), (7, 5, 100)])
def test_ge2e_loss_kernel(dev, S, P, D):
    g = np.random.default_rng(S)
    N, ld = S * P, D + 8
    x = torch.zeros(N, ld, device=dev); x[:, :D] = torch.tensor(g.normal(0, 1, (N, D)).astype(np.float32))
    wb = torch.tensor([10.0, -5.0, 0, 0], device=dev)
    out, dx = torch.zeros(4, device=dev), torch.zeros(N, ld, device=dev)
    ws = torch.zeros(int(lib.load().mstts_ge2e_ws_floats(N, D, S)), device=dev)
    lib.call("mstts_ge2e_loss_fwd_bwd", lib.ptr(x), ld, S, P, D, lib.ptr(wb), lib.ptr(out), lib.ptr(dx), ld, lib.ptr(ws))
    x64 = x[:, :D].double().cpu().requires_grad_(True)
    w64, b64 = torch.tensor(10.0, dtype=torch.float64, requires_grad=True), torch.tensor(-5.0, dtype=torch.float64, requires_grad=True)
    loss = OT.ge2e_loss(x64, P, w64, b64)
    loss.backward()
    assert abs(float(out[0]) - float(loss.detach())) < 1e-4 * max(1.0, float(loss.detach()))
    assert rel_err(t2n(dx[:, :D]), t2n(x64.grad)) < 1e-3
    assert abs(float(out[1]) - float(w64.grad)) < 1e-4 * max(1e-3, abs(float(w64.grad))) + 1e-6 and abs(float(out[2])) < 1e-6 and abs(float(b64.grad)) < 1e-9
    assert float(dx[:, D:].abs().max()) == 0.0


@pytest.mark.parametrize("S,P,T,kw", [(3, 2, 7, {}), (4, 3, 12, dict(spk=64, spk_lstm=64, n_mel=80)),
                                      (12, 3, 9, dict(spk=256, spk_lstm=256, n_mel=80))])       # reference widths, 36 rows: the persistent launches, 2 row groups
def test_speaker_train_step_parity(dev, S, P, T, kw):
    """Loss, every gradient (incl. the loss's weight / bias) and the parameters after TF-Adam over two trainer steps."""
    pd, od = dims_pair(**kw)
    values = OM.init_params(od, 9)
    g = np.random.default_rng(10)
    for k in values:
        if k.startswith(OM.P_S) and k.endswith("bias"):
            values[k] = g.normal(0, 0.1, values[k].shape)
    N = S * P
    mel = np.clip(g.normal(0, 1.5, (N, T, od.n_mel)), -4, 4).astype(np.float32)
    eng = SpeakerTrainEngine(pd, device=dev, values=values)
    params, lv, opt = values, {"loss/weight": 10.0, "loss/bias": -5.0}, None
    for step in range(2):
        masks = {}
        for i in range(od.spk_lstm_n):
            masks["s_zc_%d" % i] = torch.tensor(g.integers(0, 2, (T, N, od.spk_lstm)).astype(np.uint8))
            masks["s_zh_%d" % i] = torch.tensor(g.integers(0, 2, (T, N, od.spk_lstm)).astype(np.uint8))
        params, lv, opt, sc, grads, out = OT.speaker_train_step(params, lv, opt, od, torch.tensor(mel), P, masks, step, return_grads=True)
        w = eng.plan(N, T)
        eng.forward(torch.tensor(mel, device=dev), w, masks={k: v.numpy() for k, v in masks.items()})
        assert rel_err(t2n(w.x[-1]), t2n(out)) < 1e-3
        eng.loss_and_backward(w, P)
        assert abs(float(w.out3[0]) - sc["Loss"]) < 1e-4 * max(1.0, abs(sc["Loss"]))
        gexp = eng.params.export(grads=True)
        bad = [(k, rel_err(gexp[k], t2n(grads[k]))) for k in gexp if k.startswith(OM.P_S) and rel_err(gexp[k], t2n(grads[k])) > 5e-3]
        assert not bad, bad
        assert abs(float(w.out3[1]) - float(grads["loss/weight"])) < 5e-3 * abs(float(grads["loss/weight"])) + 1e-7
        if od.spk_lstm == 256 and eng.persist_lstm:
            assert w.persist and all(w.phist_valid) and eng.persist_lstm_fallbacks == 0      # this case really ran on the persistent launches
        eng.adam_step(w)
        now = eng.params.export()
        bad = [(k, rel_err(now[k], t2n(params[k]))) for k in now if k.startswith(OM.P_S) and rel_err(now[k], t2n(params[k])) > 2e-3]
        assert not bad, bad
        assert abs(float(eng.wb[0]) - lv["loss/weight"]) < 1e-4 and abs(float(eng.wb[1]) - lv["loss/bias"]) < 1e-4


def test_speaker_embedding_surface(dev, tmp_path, monkeypatch):
    """Speaker_Embedding(is_Training).Train_Step learns a separable synthetic batch; Save / Restore; the file is what
    Tacotron2.Speaker_Embedding_Load reads; Inference gives whole-tensor-normalised embeddings."""
    from multi_speaker_tts_amd import Hyper_Parameters as hp
    from multi_speaker_tts_amd.Speaker_Embedding import Speaker_Embedding, TRAIN_KEYS
    from multi_speaker_tts_amd.params import Dims
    monkeypatch.setattr(hp.Speaker_Embedding, "Checkpoint_Path", str(tmp_path / "se"))
    monkeypatch.setattr(hp, "Checkpoint_Path", str(tmp_path / "ckpt"))
    monkeypatch.setattr(hp.Speaker_Embedding.Train, "Frame_Range", (20, 24))
    dims = Dims(emb=32, enc_conv_ch=32, enc_lstm=16, spk=64, prenet=16, dec_lstm=32, post_ch=16, bank_ch=8, proj1_ch=16, birnn=8, n_spec=20,
                spk_lstm=64, max_inf=4)
    m = Speaker_Embedding(device=dev, dims=dims)
    pat = m.Synthetic_Pattern(speakers=6, per_speaker=4, seed=2)
    res = [m.Train_Step(pat) for _ in range(40)]
    assert set(TRAIN_KEYS) <= set(res[0]) and res[-1]["Global_Step"] == 39 and res[-1]["Loss"] < 0.7 * res[0]["Loss"]
    m.Save()
    saved = m.params.export()
    m2 = Speaker_Embedding(device=dev, dims=dims, seed=5)
    m2.Restore()
    assert m2.engine.global_step == 40 and torch.equal(m2.engine.wb.cpu(), m.engine.wb.cpu())
    assert all(np.array_equal(saved[k], m2.params.export()[k]) for k in saved if k.startswith("speaker_embedding"))
    emb = m2.Inference([np.clip(np.random.default_rng(i).normal(0, 1.5, (230, 80)), -4, 4).astype(np.float32) for i in range(3)])
    assert emb.shape == (3, 64) and abs(float((emb ** 2).sum()) - 1.0) < 1e-4
    from multi_speaker_tts_amd.MSTTS_SV import Tacotron2
    t = Tacotron2(is_Training=False, device=dev, dims=dims, allow_random_init=True)
    mine = t.params.export()
    assert all(np.array_equal(saved[k], mine[k]) for k in saved if k.startswith("speaker_embedding"))
