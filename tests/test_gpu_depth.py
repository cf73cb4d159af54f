"""Parity at the DEPTH the headline metric is quoted on (Modules.py:212-237: the teacher-forced loop runs max(Mel_Length) + 1 = 801
steps; Hyper_Parameters.py:53 allows 1000 free-running steps): the HIP path against the fp64 oracle at the reference's decoder widths
over 51 / 201 / 801 decoder steps (B = 4 x 64 tokens) and once at the full headline shape (B = 32 x 128 tokens x 801 steps), and the
free-running, stop-gated loop (one persistent launch) over >= 100 steps at B = 4 and >= 400 steps at BASELINE configs[3]'s shape (B = 16,
mixed text lengths up to 128 tokens) with rows that stop at different steps.  Everything recurrent in the
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


def _grad_errors(eng, values, grads):
    ggot = eng.params.export(grads=True)
    worst = {}
    for k, gr in grads.items():
        ref = t2n(gr).astype(np.float64)
        mine = ggot[k].astype(np.float64) + (1e-6 * np.asarray(values[k]) if OM.in_weight_reg(k) else 0.0)
        worst[k] = float(np.abs(mine - ref).max() / (np.abs(ref).max() + 1e-9))
    return worst


def _three_way(eng, w, od, values, batch, grads, worst):
    """Where does the gradient error at this shape come from?  Three more columns next to HIP vs the fp64 oracle with the HIP path's L1 sign
    pattern injected (`worst`): (a) HIP vs the fp64 oracle WITHOUT that injection - the bound round 4 had to loosen to 1e-2; (b) the oracle's own
    autograd evaluated in fp32 (same inputs, masks and ReLU pattern, no L1 injection) vs its fp64 evaluation - what a plain fp32 evaluation of
    the same graph loses; (c) the same fp32 evaluation vs fp64 with the fp32 evaluation's OWN sign pattern handed to fp64 - its rounding error
    proper.  If (a) ~ (b) >> `worst` ~ (c), the 4-5e-3 is the sign of d|x|/dx flipping on a few dozen elements, not an arithmetic defect."""
    plain = {k: v for k, v in w.oracle_masks.items() if not k.startswith("l1_sign_")}
    _, _, _, g64_plain, _ = OT.train_step(values, None, od, batch, plain, 0, return_grads=True)
    _, _, _, g32, out32 = OT.train_step(values, None, od, batch, plain, 0, dtype=torch.float32, return_grads=True)
    own = dict(plain)
    mel = batch["Mel"].to(torch.float32)
    own["l1_sign_linear"] = torch.sign(out32["Linear"][:, :-1] - mel)
    own["l1_sign_post"] = torch.sign(out32["Mel"][:, :-1] - mel)
    _, _, _, g64_own, _ = OT.train_step(values, None, od, batch, own, 0, return_grads=True)
    hip_plain = _grad_errors(eng, values, g64_plain)
    f = lambda a, b: float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-9))
    fp32_plain = {k: f(g32[k], g64_plain[k]) for k in g32}
    fp32_own = {k: f(g32[k], g64_own[k]) for k in g32}
    flips = {n: int((w.oracle_masks["l1_sign_" + n].double() != own["l1_sign_" + n].double()).sum()) for n in ("linear", "post")} if "l1_sign_linear" in w.oracle_masks else None
    names = sorted(hip_plain, key=lambda k: -hip_plain[k])[:8]
    table = [dict(variable=k, hip_vs_fp64_l1_injected=worst[k], hip_vs_fp64_plain=hip_plain[k], fp32_oracle_vs_fp64_plain=fp32_plain[k],
                  fp32_oracle_vs_fp64_own_l1_pattern=fp32_own[k]) for k in names]
    summary = dict(worst_hip_vs_fp64_l1_injected=max(worst.values()), worst_hip_vs_fp64_plain=max(hip_plain.values()),
                   worst_fp32_oracle_vs_fp64_plain=max(fp32_plain.values()), worst_fp32_oracle_vs_fp64_own_l1_pattern=max(fp32_own.values()),
                   l1_signs_hip_vs_fp32_oracle_differ=flips)
    return table, summary


def _train_depth_case(dev, B, Te, L, tag, grad_tol=5e-3, three_way=False):
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    OM.RELU_INJECTED.update(elements=0, differ=0)
    OT.L1_INJECTED.update(elements=0, differ=0)
    eng, w, od, values, batch, sc, grads, out, new_p = _engine_vs_oracle(dev, B, Te, L, True, seed=17, **REF)
    l1_injected = dict(OT.L1_INJECTED)
    assert not w.persist or (eng.persist_fallbacks == 0 and eng.persist_bwd_fallbacks == 0 and eng.persist_enc_fallbacks == 0)
    errs = {"linear": rel_err(t2n(w.linear), t2n(out["Linear"])), "mel": rel_err(t2n(w.mel_out), t2n(out["Mel"])),
            "stop": rel_err(t2n(w.stop), t2n(out["Stop_Logit"])),
            "align": rel_err(t2n(w.align_hist).transpose(1, 2, 0), t2n(out["Attention_History"]))}
    # error of the mel by decoder step: the growth curve
    dm = np.abs(t2n(w.mel_out).astype(np.float64) - t2n(out["Mel"]).astype(np.float64)).max(axis=(0, 2)) / np.abs(t2n(out["Mel"])).max()
    curve = {int(s): float(dm[: s + 1].max()) for s in (0, 10, 50, 100, 200, 400, 800) if s <= L}
    ggot = eng.params.export(grads=True)
    worst = _grad_errors(eng, values, grads)
    gmax = max(worst.values())
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    got = eng.scalars(w)
    _record(tag, dict(B=B, tokens=Te, L=L, steps=L + 1, persistent=bool(w.persist), forward=errs, mel_err_up_to_step=curve, worst_gradient=gmax,
                      worst_gradients=[[k, v] for k, v in top], relu_injected=dict(OM.RELU_INJECTED), l1_sign_injected=l1_injected))
    if three_way:
        table, summary = _three_way(eng, w, od, values, batch, grads, worst)
        _record(tag + "_gradient_error_three_way", dict(B=B, tokens=Te, L=L, summary=summary, table=table))
        print("three-way gradient table (error over the tensor's max):", summary)
        for row in table:
            print("   ", row)
    print("%s B %d x %d tokens, depth %d: forward %s, mel error up to step %s, worst gradient %.2e, injected ReLU pattern: %d of %d elements differ "
          "(all inside the kink band)" % (tag, B, Te, L, errs, curve, gmax, OM.RELU_INJECTED["differ"], OM.RELU_INJECTED["elements"]))
    for k, e in errs.items():
        assert e < 1e-3, (k, e, L)
    # the injected kink patterns are held to a COUNT, not only to a band - at 4 M elements the +-2e-3 band of the L1 terms holds
    # thousands of elements, and a kernel that mis-signed d|x| (or mis-gated a ReLU) near zero on a visible share of them would pass a
    # band-only check.  Measured: 3 of 4 096 000 L1 signs at the headline shape, 0-2 ReLU gates; allowed: a handful plus 1e-5 of the elements.
    for name, cnt in (("L1 sign", l1_injected), ("ReLU", dict(OM.RELU_INJECTED))):
        assert cnt["differ"] <= 8 + 1e-5 * cnt["elements"], "%s pattern: %d of %d injected elements differ from the oracle's own" % (name, cnt["differ"], cnt["elements"])
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
    _train_depth_case(dev, 4, 64, L, "train", three_way=(L == 200))


@pytest.mark.parametrize("L", [200])
def test_depth_parity_train_bf16(dev, monkeypatch, L):
    """BASELINE config 3 arithmetic at depth, on the persistent bf16 launches: B = 4 x 64 tokens x L frames at the reference widths against the
    bf16-EMULATING oracle (oracle.model.GEMM_BF16 + RECURRENT_BF16) with the bounds of test_gpu_model.py::test_train_step_parity_bf16_full, whose
    docstring says why a whole step in this mode is compared in relative L2: the emulation must be clearly closer to the HIP path than exact
    arithmetic is (at this depth: at least twice as close on the decoder output - measured 2.9 x at 201 steps, 1.2e-3 against 3.4e-3; the
    short-sequence test asks for 3 x), forward tensors <= 2e-3 / 2e-2, gradients at the oracle's own noise level."""
    from test_gpu_model import _l2
    from helpers import to_dev
    from multi_speaker_tts_amd.engine import TrainEngine
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    pd, od = dims_pair(**REF)
    values = OM.init_params(od, 17)
    g = np.random.default_rng(18)
    for k in values:
        if k.endswith(("bias", "beta", "bias_b")):
            values[k] = g.normal(0, 0.1, values[k].shape)
        if k.endswith("gamma"):
            values[k] = 1.0 + g.normal(0, 0.1, values[k].shape)
    B, Te = 4, 64
    batch = OT.synthetic_batch(od, B, Te, L, seed=17, ragged=True)
    masks = OT.make_masks(od, B, Te, L + 1, True, seed=OT.step_seed(1234, 0))
    eng = TrainEngine(pd, device=dev, values=values, recurrent_dtype="bf16", gemm_dtype="bf16")
    if not (eng.persist and eng.persist_bf16):
        pytest.skip("persistent bf16 loops not available on this device")
    w = eng.plan(B, Te, L)
    eng.forward(to_dev(batch, dev), w, seed=OT.step_seed(1234, 0))
    eng.loss_and_backward(w)
    torch.cuda.synchronize()
    assert w.persist and w.persist_bwd and w.pdesc.recurrent_bf16 == 1 and eng.persist_fallbacks == 0 and eng.persist_bwd_fallbacks == 0
    omasks = dict(masks)
    for i in range(od.enc_conv_n):
        omasks["relu_enc_%d" % i] = (w.enc_a[i] > 0).reshape(B, Te, od.enc_conv_ch).cpu()
    monkeypatch.setattr(OM, "KINK_BAND", 5e-2)
    ggot = eng.params.export(grads=True)
    res, relu_counts = {}, {}
    for mode in ("emulated", "exact"):
        monkeypatch.setattr(OM, "RECURRENT_BF16", mode == "emulated")
        monkeypatch.setattr(OM, "GEMM_BF16", mode == "emulated")
        OM.RELU_INJECTED.update(elements=0, differ=0)
        _, _, sc, grads, out = OT.train_step(values, None, od, batch, omasks, 0, return_grads=True)
        relu_counts[mode] = dict(OM.RELU_INJECTED)
        gl2 = {k: _l2(ggot[k].astype(np.float64) + (1e-6 * np.asarray(values[k]) if OM.in_weight_reg(k) else 0.0), t2n(gr)) for k, gr in grads.items()}
        res[mode] = dict(linear=_l2(t2n(w.linear), t2n(out["Linear"])), mel=_l2(t2n(w.mel_out), t2n(out["Mel"])),
                         align=_l2(t2n(w.align_hist).transpose(1, 2, 0), t2n(out["Attention_History"])), loss=sc["Loss"],
                         grads_median=float(np.median(list(gl2.values()))), grads_worst=max(gl2.items(), key=lambda kv: kv[1]))
    em, ex = res["emulated"], res["exact"]
    _record("train_bf16", dict(B=B, tokens=Te, L=L, steps=L + 1, persistent_bf16=True, vs_emulating_oracle=em, vs_exact_oracle=ex, relu_injected=relu_counts))
    print("bf16 depth %d: vs emulating oracle %s; vs exact oracle %s; injected ReLU pattern differs from the oracle's own on %s" % (L, em, ex, relu_counts))
    # with the band widened to 5e-2 for bf16 products the band alone could hide a wrong activation on a visible share of the
    # elements - so the COUNT of elements on which the HIP path's encoder ReLU pattern differs from the oracle's own is bounded too: the
    # bf16-emulating oracle rounds where the HIP path rounds, so they may part only on pre-activations within a bf16 product's rounding of zero
    # (<= 0.1 % of the elements); the exact-arithmetic oracle is further away (<= 1 %)
    assert relu_counts["emulated"]["elements"] == B * Te * od.enc_conv_ch * od.enc_conv_n
    assert relu_counts["emulated"]["differ"] <= 1e-3 * relu_counts["emulated"]["elements"], relu_counts
    assert relu_counts["exact"]["differ"] <= 1e-2 * relu_counts["exact"]["elements"], relu_counts
    assert em["linear"] < 2e-3 and em["align"] < 2e-3 and em["mel"] < 2e-2, em
    assert em["linear"] < ex["linear"] / 2 and em["mel"] < ex["mel"], (em, ex)
    assert abs(eng.scalars(w)["Loss"] - em["loss"]) <= 1e-3 * max(1.0, abs(em["loss"]))
    assert em["grads_worst"][1] < 0.15 and em["grads_median"] < 6e-2, em


def test_headline_shape_parity(dev):
    """The exact shape the headline metric is quoted on (BASELINE configs[1]; MSTTS_SV.py:129-161, Hyper_Parameters.py:69, Modules.py:215):
    ONE train step at B = 32 x 128 tokens x 800 frames (801 decoder steps), reference widths, every attention row and every key position of
    the persistent kernels busy, fp32 HIP against the **fp64** oracle: forward <= 1e-3, losses <= 1e-4, zero fallbacks, every gradient <= 1e-3 of
    its maximum (measured 3.7e-4).  Round 4 had to loosen this bound to 1e-2 (4.6-5.5e-3 on the location-layer dense kernel and the postnet
    convolution kernels) without knowing why; the three-way table this test records (profiles/r05_depth_parity.jsonl,
    `headline_shape_gradient_error_three_way`) settles it: the L1 terms' gradient is sign(prediction - target) / n, and of the 4 096 000 elements
    that feed them THREE lie within the fp32 forward error of 0, where fp32 and fp64 evaluations take different signs.  One flipped element
    moves every weight gradient upstream by about one row's contribution of 25 632.  With the HIP path's sign pattern handed to the oracle
    (oracle.train.abs_at, the L1 counterpart of the ReLU pattern; it may differ from the oracle's own only inside +-2e-3, asserted) the worst
    gradient error is 3.7e-4; without it 4.8e-3; the oracle's OWN fp32 autograd against its fp64 one: 3.2e-3 plain, 1.1e-4 with its own
    pattern.  Not cancellation in any sum, not the split-K atomics.  The oracle's autograd tapes at this size (one fp64 + two fp64 / one fp32
    for the table, in sequence) need tens of GB of host memory: skipped on a host without it."""
    import psutil
    need = 96 << 30
    if psutil.virtual_memory().available < need:
        pytest.skip("fp64 oracle tape of the full shape needs ~%d GB of host memory" % (need >> 30))
    w = _train_depth_case(dev, 32, 128, 800, "headline_shape", grad_tol=1e-3, three_way=True)
    assert w.persist and w.persist_bwd and w.persist_enc


def _pick_rows_and_bias(raw, B, min_steps, min_distinct=3):
    """raw [P, S]: bias-free stop logits of a POOL of independent rows (the free-running trajectory does not depend on the stop bias:
    finished rows keep computing, impute_finished = False, Modules.py:116).  Choose a bias and B rows of the pool such that every chosen row
    stops (logit >= 0, Modules.py:217), at >= min_distinct different steps, the last one at step >= min_steps - 1, and the smallest |logit|
    of any decision that matters (each row up to and including its own first crossing) is as large as possible."""
    P, S = raw.shape
    best = None
    for beta in np.linspace(-raw.max(), -raw.min(), 1500):
        z = raw + beta
        cross = z >= 0
        has = cross.any(axis=1)
        first = np.where(has, cross.argmax(axis=1), -1)
        upto = np.arange(S)[None, :] <= first[:, None]
        margin = np.where(has, np.where(upto, np.abs(z), np.inf).min(axis=1), -1.0)
        elig = np.flatnonzero(has & (first < S - 1))
        late = [r for r in elig if first[r] >= min_steps - 1]
        if len(elig) < B or not late:
            continue
        anchor = max(late, key=lambda r: margin[r])
        rows, seen = [anchor], {int(first[anchor])}
        order = sorted((r for r in elig if r != anchor), key=lambda r: -margin[r])
        for r in order:                                  # first the best rows that add a NEW stop step, up to min_distinct ...
            if len(seen) >= min_distinct:
                break
            if int(first[r]) not in seen:
                rows.append(r); seen.add(int(first[r]))
        for r in order:                                  # ... then the best of the rest
            if len(rows) >= B:
                break
            if r not in rows:
                rows.append(r)
        if len(rows) < B or len(seen) < min_distinct:
            continue
        score = float(min(margin[r] for r in rows))
        if best is None or score > best[0]:
            best = (score, float(beta), sorted(int(r) for r in rows))
    return best


def _free_running_case(dev, B, Te, max_inf, pool, min_steps, tag, seed):
    from multi_speaker_tts_amd.inference import InferEngine
    from multi_speaker_tts_amd import lib as _lib
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    pd, od = dims_pair(dec_lstm=1024, prenet=256, enc_lstm=256, spk=256, n_mel=80, max_inf=max_inf)
    values = OM.init_params(od, seed)
    g = np.random.default_rng(seed + 1)
    for k in values:
        if k.endswith(("bias", "beta")) and "highway" not in k:
            values[k] = g.normal(0, 0.1, values[k].shape)
    pk, bk = "decoder/decoder/linear_projection/dense/kernel", "decoder/decoder/linear_projection/dense/bias"
    values[pk] = np.array(values[pk]); values[pk][:, -1] *= -12.0       # a livelier stop logit; flipped, so that the start of the sequence
    values[bk] = np.array(values[bk]); values[bk][-1] = -100.0          # (where the random-weight logit peaks) is its minimum
    P = pool
    tok = g.integers(2, od.n_tok, size=(P, Te)).astype(np.int32)
    lengths = g.integers(max(3, Te // 4), Te + 1, P).astype(np.int32)
    lengths[:: 5] = Te                                                   # every fifth row of the pool has the full length
    for b in range(P):
        tok[b, 0] = 0; tok[b, lengths[b] - 1] = 1; tok[b, lengths[b]:] = 1        # Feeder.py:189-204: <S> ... <E>, padded with <E>
    spk = g.normal(0, 1, (P, od.spk)); spk = spk / np.sqrt((spk ** 2).sum()) * np.sqrt(P / B)      # (whole-tensor norm 1 for a B-row batch, Speaker_Embedding/Modules.py:137)
    masks = OT.make_masks(od, P, Te, od.max_inf + 1, False, seed=seed + 2)

    def oracle(rows):
        ob = {"Token": torch.from_numpy(tok[rows]), "Token_Length": torch.from_numpy(lengths[rows]), "Mel": torch.zeros(len(rows), 1, od.n_mel, dtype=torch.float64),
              "Mel_Length": torch.zeros(len(rows), dtype=torch.int32), "Speaker_Embedding": torch.tensor(spk[rows], dtype=torch.float64)}
        mk = {k: v.index_select(OT.mask_batch_axis(k), torch.as_tensor(rows)) for k, v in masks.items()}
        with torch.no_grad():
            return OM.forward(OM.to_torch(values), od, ob, False, mk, with_vocoder=False), mk

    ref, _ = oracle(list(range(P)))
    raw = t2n(ref["Stop_Logit"]) + 100.0                                  # [P, max_inf + 1] bias-free logits
    assert raw.shape == (P, od.max_inf + 1)
    best = _pick_rows_and_bias(raw, B, min_steps)
    assert best is not None, "no stop bias lets %d rows of the pool stop at three or more different steps, the last beyond step %d" % (B, min_steps)
    margin, beta, rows = best
    values[bk][-1] = beta
    ref, mk = oracle(rows)
    z = t2n(ref["Stop_Logit"])
    first = [int(np.argmax(z[b] >= 0)) for b in range(B)]
    S = max(first) + 1
    assert ref["Linear"].shape[1] == S and S >= min_steps and len(set(first)) >= 3, (ref["Linear"].shape, first)
    scale = float(np.abs(z).max())
    eng = InferEngine(pd, device=dev, values=values)
    got = eng.forward({"Token": tok[rows], "Token_Length": lengths[rows], "Speaker_Embedding": spk[rows].astype(np.float32)},
                      masks={k: v.numpy() for k, v in mk.items()}, with_vocoder=False)
    d = eng.d
    assert _lib.load().mstts_persist_infer_supported(B, d.dec_lstm, d.prenet, d.mem, d.att, Te, d.att_k, d.n_mel), "the persistent free-running loop is not available on this device"
    assert eng.persist_infer_launches == 1 and eng.persist_infer_fallbacks == 0, eng.persist_infer_status      # the whole loop was ONE launch
    arrivals, abort, left, finished_rows, steps = eng.persist_infer_status
    assert (abort, left, finished_rows, steps) == (0, 256, B, S), eng.persist_infer_status
    errs = {k: rel_err(got[k], t2n(ref[k])) for k in ("Linear", "Mel", "Stop", "Attention_History")}
    _record(tag, dict(B=B, tokens=Te, token_lengths=lengths[rows].tolist(), persistent_launches=eng.persist_infer_launches, fallbacks=eng.persist_infer_fallbacks,
                      steps=S, first_stop_step=first, threshold_margin=margin, threshold_margin_over_logit_scale=margin / scale, pool=P, errors=errs))
    print("%s: B %d, %d steps in one launch, rows stop at %s (margin %.3g of logit scale %.3g), errors %s" % (tag, B, S, first, margin, scale, errs))
    assert margin >= 1e-2, margin                                         # a decision this far from the threshold cannot flip on fp32 rounding (errors ~1e-6 of the scale)
    assert got["Linear"].shape == (B, S, od.n_mel), (got["Linear"].shape, S)
    for k, e in errs.items():
        assert e < 1e-3, (k, e)
    cut = lambda stop: [int(np.argmax(stop[b] > 0.5)) if (stop[b] > 0.5).any() else stop.shape[1] for b in range(B)]     # MSTTS_SV.py:395
    assert cut(got["Stop"]) == cut(t2n(ref["Stop"])) == first


def test_depth_parity_free_running(dev):
    """>= 100 free-running decoder steps at the reference widths in ONE persistent launch, B = 4 x 64 tokens, rows stopping at >= 3
    DIFFERENT steps (Modules.py:216-219: a row is finished once its stop logit is >= 0; the loop ends when every row is, :395,409;
    MSTTS_SV.py:395: each row is cut at its own first stop), fp64 oracle <= 1e-3, cut points equal.  Rows and stop bias are chosen from a pool
    of independent rows in a first oracle pass so that every decision is >= 1e-2 away from the threshold."""
    _free_running_case(dev, 4, 64, 199, pool=32, min_steps=100, tag="free_running", seed=41)


def test_depth_parity_free_running_config4(dev):
    """The same at BASELINE configs[3]'s shape: batch 16, mixed text lengths up to 128 tokens, >= 400 free-running steps
    (Hyper_Parameters.py:53 allows 1000), one persistent launch, zero fallbacks, per-row cut points equal to the fp64 oracle's."""
    _free_running_case(dev, 16, 128, 479, pool=96, min_steps=400, tag="free_running_config4", seed=43)
