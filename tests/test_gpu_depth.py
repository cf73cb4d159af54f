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
    # the same backward pass a second time on the same forward state: the two gradient slabs differ only by the order of the split-K atomics
    # (recorded: a transient error in one pass would show here whether or not it crosses the parity bound)
    g_first = eng.params.grad.clone()
    eng.loss_and_backward(w)
    torch.cuda.synchronize()
    repeat_diff = float((eng.params.grad - g_first).abs().max() / g_first.abs().max())
    _record(tag + "_backward_twice", dict(B=B, tokens=Te, L=L, max_abs_diff_over_max_abs=repeat_diff))
    print("backward pass twice on the same forward state: max |difference| / max |gradient| = %.2e" % repeat_diff)
    assert repeat_diff < 1e-4, repeat_diff
    bad = {k: v for k, v in worst.items() if v > grad_tol}
    if bad:
        # tell a transient error from a systematic one before failing: the same backward pass again on the same forward state, then the whole step
        first = {k: ggot[k].astype(np.float64) for k in bad}
        keep = {n: getattr(w, n).clone() for n in ("d_post", "d_linear", "d_pj")}
        keep.update({"post_dz[%d]" % i: t.clone() for i, t in enumerate(w.post_dz)})
        eng.loss_and_backward(w)
        torch.cuda.synchronize()
        again = eng.params.export(grads=True)
        moved = {n: float((getattr(w, n.split("[")[0])[int(n[-2])] if "[" in n else getattr(w, n)).double().sub(t.double()).abs().max() / (t.abs().max() + 1e-30))
                 for n, t in keep.items()}
        print("RECHECK second backward pass on the same forward state: gradient moved by",
              {k: float(np.abs(again[k] - first[k]).max() / (np.abs(first[k]).max() + 1e-30)) for k in bad}, "intermediates moved by", moved,
              "fallbacks", eng.persist_fallbacks, eng.persist_bwd_fallbacks, eng.persist_enc_fallbacks)
        _record(tag + "_recheck", dict(bad=bad, moved=moved))
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
    the persistent kernels busy, fp32 HIP against the **fp64** oracle: forward <= 1e-3, losses <= 1e-4, zero fallbacks, every gradient <= 1e-2 of
    its maximum.  (The 5e-3 bound of the smaller cases does not carry over: at this size every weight gradient is an fp32 sum over 25 632 rows
    with heavy cancellation - measured over three runs of this test: location-layer dense kernel 4.6e-3 / 5.5e-3 / 4.8e-3, the postnet convolution
    kernels 4.1-4.2e-3, everything else below 2e-3, against 1.7e-4 worst at B = 4; run to run the atomics of the split-K products move them by
    ~1e-3.  A wrong gradient is off by tens of percent.)  The oracle's autograd tape at this size needs tens of GB of host memory: skipped on
    a host without it."""
    import psutil
    need = 96 << 30
    if psutil.virtual_memory().available < need:
        pytest.skip("fp64 oracle tape of the full shape needs ~%d GB of host memory" % (need >> 30))
    w = _train_depth_case(dev, 32, 128, 800, "headline_shape", grad_tol=1e-2)
    assert w.persist and w.persist_bwd and w.persist_enc
