"""Parity at the DEPTH the headline metric is quoted on (Modules.py:212-237: the teacher-forced loop runs max(Mel_Length) + 1 = 801
steps; Hyper_Parameters.py:53 allows 1000 free-running steps): the HIP path against the fp64 oracle at the reference's decoder widths
over 51 / 201 / 801 decoder steps (B = 4 x 64 tokens) and once at the full headline shape (B = 32 x 128 tokens x 801 steps), and up to 200
free-running steps with rows that stop at different steps.  Everything recurrent in the
HIP path is fp32 with hardware exp-based gates; these tests are where its error growth over the sequence is measured."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import dims_pair, rel_err, t2n
from oracle import model as OM, train as OT
from test_gpu_model import REF, _engine_vs_oracle

pytestmark = pytest.mark.gpu


def _record(tag, payload):
    """Measured curves go to gpurun_out/depth_parity.jsonl when that directory exists (it is what DESIGN.md quotes)."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(root):
        with open(os.path.join(root, "depth_parity.jsonl"), "a") as f:
            f.write(json.dumps(dict(tag=tag, **payload)) + "\n")


def _train_depth_case(dev, B, Te, L, tag, grad_tol=5e-3):
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    OM.RELU_INJECTED.update(elements=0, differ=0)
    eng, w, od, values, batch, sc, grads, out, new_p = _engine_vs_oracle(dev, B, Te, L, True, seed=17, **REF)
    assert not w.persist or (eng.persist_fallbacks == 0 and eng.persist_bwd_fallbacks == 0 and eng.persist_enc_fallbacks == 0)
    errs = {"linear": rel_err(t2n(w.linear), t2n(out["Linear"])), "mel": rel_err(t2n(w.mel_out), t2n(out["Mel"])),
            "stop": rel_err(t2n(w.stop), t2n(out["Stop_Logit"])),
            "align": rel_err(t2n(w.align_hist).transpose(1, 2, 0), t2n(out["Attention_History"]))}
    # error of the mel by decoder step: the growth curve
    dm = np.abs(t2n(w.mel_out).astype(np.float64) - t2n(out["Mel"]).astype(np.float64)).max(axis=(0, 2)) / np.abs(t2n(out["Mel"])).max()
    curve = {int(s): float(dm[: s + 1].max()) for s in (0, 10, 50, 100, 200, 400, 800) if s <= L}
    ggot = eng.params.export(grads=True)
    worst = {}
    for k, gr in grads.items():
        ref = t2n(gr).astype(np.float64)
        mine = ggot[k].astype(np.float64) + (1e-6 * np.asarray(values[k]) if OM.in_weight_reg(k) else 0.0)
        worst[k] = float(np.abs(mine - ref).max() / (np.abs(ref).max() + 1e-9))
    gmax = max(worst.values())
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    got = eng.scalars(w)
    _record(tag, dict(B=B, tokens=Te, L=L, steps=L + 1, persistent=bool(w.persist), forward=errs, mel_err_up_to_step=curve, worst_gradient=gmax,
                      worst_gradients=[[k, v] for k, v in top], relu_injected=dict(OM.RELU_INJECTED)))
    print("%s B %d x %d tokens, depth %d: forward %s, mel error up to step %s, worst gradient %.2e, injected ReLU pattern: %d of %d elements differ "
          "(all inside the kink band)" % (tag, B, Te, L, errs, curve, gmax, OM.RELU_INJECTED["differ"], OM.RELU_INJECTED["elements"]))
    for k, e in errs.items():
        assert e < 1e-3, (k, e, L)
    for k in ("Linear_Loss", "Postnet_Loss", "Stop_Loss", "Loss"):
        assert abs(got[k] - sc[k]) <= 1e-4 * max(1.0, abs(sc[k])), (k, got[k], sc[k])
    print("worst gradients:", top)
    bad = {k: v for k, v in worst.items() if v > grad_tol}
    assert not bad, bad
    return w


@pytest.mark.parametrize("L", [50, 200, 800])
def test_depth_parity_train(dev, L):
    """One train step at B = 4 x 64 tokens x L frames, reference widths, fp32 HIP vs fp64 oracle (forward tensors, loss scalars, every
    gradient).  north_star's bound - mel within 1e-3 relative - must hold at every depth, including the 801 steps of BASELINE configs[1]."""
    _train_depth_case(dev, 4, 64, L, "train")


def test_headline_shape_parity(dev):
    """The exact shape the headline metric is quoted on (BASELINE configs[1]; MSTTS_SV.py:129-161, Hyper_Parameters.py:69, Modules.py:215):
    ONE train step at B = 32 x 128 tokens x 800 frames (801 decoder steps), reference widths, every attention row and every key position of
    the persistent kernels busy, fp32 HIP against the **fp64** oracle: forward <= 1e-3, losses <= 1e-4, zero fallbacks; every gradient <= 5e-3 of
    its maximum EXCEPT the three variables of the attention's location layer, whose bound here is 1e-2: their gradient is one 31 x 128 filter
    gradient summed over 25 632 row-steps x 128 positions by 2 048 workgroups with fp32 atomics (lsa_param_bwd_kernel) - measured 4.6e-3 and
    5.5e-3 of the maximum on `attention_convolution_dense_layer/dense/kernel` in two runs of this test (the order of the atomics differs run to
    run), 1.7e-4 at B = 4; everything else stays below 2e-3.  The oracle's autograd tape at this size needs tens of GB of host memory: skipped on
    a host without it."""
    import psutil
    need = 96 << 30
    if psutil.virtual_memory().available < need:
        pytest.skip("fp64 oracle tape of the full shape needs ~%d GB of host memory" % (need >> 30))
    w = _train_depth_case(dev, 32, 128, 800, "headline_shape", grad_tol=1e-2)
    assert w.persist and w.persist_bwd and w.persist_enc
    import json
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "depth_parity.jsonl")
    if os.path.exists(root):         # the tight bound for everything that is not the location layer (read back from the record just written)
        rec = [json.loads(l) for l in open(root) if l.strip()][-1]
        if rec.get("tag") == "headline_shape":
            for name, v in rec["worst_gradients"]:
                assert v <= 5e-3 or "attention_convolution_dense_layer" in name, (name, v)


def test_depth_parity_free_running(dev):
    """200 free-running decoder steps at the reference widths, rows stopping at DIFFERENT steps (Modules.py:216-219: a row is finished
    once its stop logit is >= 0; the loop ends when every row is; MSTTS_SV.py:395: each row is cut at its own first stop).  The decoder's
    trajectory does not depend on the stop bias (finished rows keep computing, impute_finished = False), so the bias is chosen from a
    first oracle pass such that every row stops, at different steps, with the widest margin around the threshold."""
    from multi_speaker_tts_amd.inference import InferEngine
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    pd, od = dims_pair(dec_lstm=1024, prenet=256, enc_lstm=256, spk=256, n_mel=80, max_inf=199)
    B, Te = 4, 64
    values = OM.init_params(od, 41)
    g = np.random.default_rng(12)
    for k in values:
        if k.endswith(("bias", "beta")) and "highway" not in k:
            values[k] = g.normal(0, 0.1, values[k].shape)
    pk = "decoder/decoder/linear_projection/dense/kernel"
    values[pk] = np.array(values[pk]); values[pk][:, -1] *= -6.0         # a livelier stop logit; flipped, so that the start of the sequence
                                                                         # (where the random-weight logit peaks) is its minimum
    batch = OT.synthetic_batch(od, B, Te, 4, seed=10, ragged=True)
    spk = g.normal(0, 1, (B, od.spk)); spk = spk / np.sqrt((spk ** 2).sum())
    masks = OT.make_masks(od, B, Te, od.max_inf + 1, False, seed=78)
    ob = {"Token": batch["Token"], "Token_Length": batch["Token_Length"], "Mel": torch.zeros(B, 1, od.n_mel, dtype=torch.float64),
          "Mel_Length": torch.zeros(B, dtype=torch.int32), "Speaker_Embedding": torch.tensor(spk, dtype=torch.float64)}
    bk = "decoder/decoder/linear_projection/dense/bias"
    values[bk] = np.array(values[bk]); values[bk][-1] = -100.0
    with torch.no_grad():
        raw = t2n(OM.forward(OM.to_torch(values), od, ob, False, masks, with_vocoder=False)["Stop_Logit"]) + 100.0      # [B, 200] bias-free logits
    assert raw.shape == (B, od.max_inf + 1)
    # the bias: every row crosses, the rows' first crossings differ, the margin |logit| around every decision is as wide as possible
    best = None
    for beta in -np.sort(raw.max(axis=1))[0] + np.linspace(0.0005, 0.05, 100):
        z = raw + beta
        first = np.array([int(np.argmax(z[b] >= 0)) if (z[b] >= 0).any() else -1 for b in range(B)])
        if (first < 0).any() or len(set(first.tolist())) < 3:
            continue
        S = int(first.max()) + 1
        margin = float(np.abs(z[:, :S]).min())
        if S >= 60 and (best is None or margin > best[0]):
            best = (margin, float(beta), first, S)
    assert best is not None, "no stop bias makes the rows stop at three or more different steps beyond step 60"
    margin, beta, first, S = best
    values[bk][-1] = beta
    with torch.no_grad():
        ref = OM.forward(OM.to_torch(values), od, ob, False, masks, with_vocoder=False)
    assert ref["Linear"].shape[1] == S
    eng = InferEngine(pd, device=dev, values=values)
    got = eng.forward({"Token": batch["Token"].numpy(), "Token_Length": batch["Token_Length"].numpy(), "Speaker_Embedding": spk.astype(np.float32)},
                      masks={k: v.numpy() for k, v in masks.items()}, with_vocoder=False)
    from multi_speaker_tts_amd import lib as _lib
    if _lib.load().mstts_persist_infer_supported(B, pd.dec_lstm, pd.prenet, pd.mem, pd.att, Te, pd.att_k, pd.n_mel):
        assert eng.persist_infer_launches == 1 and eng.persist_infer_fallbacks == 0, eng.persist_infer_status      # the whole loop was ONE launch
    errs = {k: rel_err(got[k], t2n(ref[k])) for k in ("Linear", "Mel", "Stop", "Attention_History")}
    _record("free_running", dict(persistent_launches=eng.persist_infer_launches, steps=S, first_stop_step=first.tolist(), threshold_margin=margin, errors=errs))
    print("free running: %d steps, rows stop at %s (margin %.3g), errors %s" % (S, first.tolist(), margin, errs))
    assert got["Linear"].shape == (B, S, od.n_mel), (got["Linear"].shape, S)
    for k, e in errs.items():
        assert e < 1e-3, (k, e)
    cut = lambda stop: [int(np.argmax(stop[b] > 0.5)) if (stop[b] > 0.5).any() else stop.shape[1] for b in range(B)]     # MSTTS_SV.py:395
    assert cut(got["Stop"]) == cut(t2n(ref["Stop"])) == first.tolist()
