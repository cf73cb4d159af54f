"""Loss, learning-rate schedule, TF-style Adam and one train step (TEST INFRASTRUCTURE).

Restates MSTTS_SV.py:127-192 (loss + optimizer) on top of oracle.model.forward; gradients come
from torch autograd.  Also defines the mask-stream table shared by specification with the HIP
host code (multi_speaker_tts_amd/masks.py keeps an independent copy of the same table).
"""
from __future__ import annotations

import numpy as np
import torch

from . import model as M
from . import rng

# stream ids of the Philox keep-masks (DESIGN.md "Randomness")
STREAMS = {
    "enc_conv_drop_%d": 1, "enc_zc_fw": 10, "enc_zh_fw": 11, "enc_zc_bw": 12, "enc_zh_bw": 13,
    "prenet_drop_%d": 20, "dec_zc_%d": 30, "dec_zh_%d": 31, "post_drop_%d": 40,
    "v_zc_fw": 50, "v_zh_fw": 51, "v_zc_bw": 52, "v_zh_bw": 53, "s_zc_%d": 60, "s_zh_%d": 61,
}


def step_seed(base_seed, step):
    return (int(base_seed) + 1000003 * int(step)) & 0xFFFFFFFFFFFFFFFF


def mask_table(d: M.Dims, B, T_enc, S, training, speaker_windows=0, vocoder=False):
    """[(name, stream, shape, keep_prob)] for one forward."""
    t = []
    for i in range(d.prenet_n):
        t.append(("prenet_drop_%d" % i, STREAMS["prenet_drop_%d"] + i, (S, B, d.prenet), 1 - d.prenet_drop))
    if not training:
        return t
    cin = d.enc_conv_ch
    for i in range(d.enc_conv_n):
        t.append(("enc_conv_drop_%d" % i, STREAMS["enc_conv_drop_%d"] + i, (B, T_enc, cin), 1 - d.conv_drop))
    for dr in ("fw", "bw"):
        t.append(("enc_zc_" + dr, STREAMS["enc_zc_" + dr], (T_enc, B, d.enc_lstm), 1 - d.zoneout))
        t.append(("enc_zh_" + dr, STREAMS["enc_zh_" + dr], (T_enc, B, d.enc_lstm), 1 - d.zoneout))
    for l in range(d.dec_lstm_n):
        t.append(("dec_zc_%d" % l, STREAMS["dec_zc_%d"] + 2 * l, (S, B, d.dec_lstm), 1 - d.zoneout))
        t.append(("dec_zh_%d" % l, STREAMS["dec_zh_%d"] + 2 * l, (S, B, d.dec_lstm), 1 - d.zoneout))
    for i in range(d.post_n):
        cout = d.post_ch if i < d.post_n - 1 else d.n_mel
        t.append(("post_drop_%d" % i, STREAMS["post_drop_%d"] + i, (B, S, cout), 1 - d.conv_drop))
    if vocoder:
        for dr in ("fw", "bw"):
            t.append(("v_zc_" + dr, STREAMS["v_zc_" + dr], (S, B, d.birnn), 1 - d.zoneout))
            t.append(("v_zh_" + dr, STREAMS["v_zh_" + dr], (S, B, d.birnn), 1 - d.zoneout))
    if speaker_windows:
        for i in range(d.spk_lstm_n):
            t.append(("s_zc_%d" % i, STREAMS["s_zc_%d"] + 2 * i, (d.spk_frames, speaker_windows, d.spk_lstm), 1 - d.zoneout))
            t.append(("s_zh_%d" % i, STREAMS["s_zh_%d"] + 2 * i, (d.spk_frames, speaker_windows, d.spk_lstm), 1 - d.zoneout))
    return t


def mask_batch_axis(name):
    """Batch-major masks ([B, T, C]: the conv-block dropouts) have their samples on axis 0, step-major ones ([S, B, C]) on axis 1."""
    return 0 if name.startswith(("enc_conv_drop", "post_drop")) else 1


def make_masks(d, B, T_enc, S, training, seed, rank=0, **kw):
    """Every mask is keyed by (seed, stream, GLOBAL sample index, position inside the sample): rank r of a data-parallel run
    owns the samples r * B ... r * B + B - 1 (for the speaker stack: windows), so 1/2/4/8-rank runs draw identical masks."""
    out = {}
    for name, stream, shape, keep in mask_table(d, B, T_enc, S, training, **kw):
        ax = mask_batch_axis(name)
        out[name] = torch.from_numpy(rng.keep_mask_rows(shape, ax, seed, stream, rank * shape[ax], keep))
    return out


def learning_rate(step, initial=1e-3, minimum=1e-5, decay_start=0, decay_step=10000, decay_rate=0.5):
    """MSTTS_SV.py:163-169: exponential_decay (non-staircase) then clip to [min, initial]."""
    lr = initial * decay_rate ** ((step - decay_start) / decay_step)
    return min(max(lr, minimum), initial)


L1_KINK_BAND = 2e-3   # |prediction - target| below which two correct implementations may disagree on the sign of the L1 term's gradient
L1_INJECTED = {"elements": 0, "differ": 0}


def abs_at(x, sign=None):
    """|x| of the L1 losses (MSTTS_SV.py:138-142: tf.losses.absolute_difference).  `sign` (optional tensor of x's shape, +-1 / 0) fixes
    which side of the kink every element counts on - the L1 counterpart of model.relu_at: the gradient of |x| is sign(x) / n, at the
    headline shape 2 x 2 M elements feed the two L1 terms, the targets are continuous, so a few dozen |prediction - target| land within
    the fp32 forward error of 0, where an fp32 and an fp64 evaluation put the element on different sides and its gradient flips by 2 / n -
    one such element moves every weight gradient upstream by about one row's contribution (measured at B = 32 x 800 frames,
    profiles/r05_depth_parity.jsonl: 3 of 4 096 000 elements differ; HIP vs fp64 4.8e-3 of the maximum plain, 3.7e-4 with the pattern
    injected; this oracle's own fp32 autograd vs its fp64 one 3.2e-3 plain, 1.1e-4 with its own pattern).  The injected pattern may differ from this evaluation's own only
    inside +-L1_KINK_BAND (asserted), so it cannot hide a wrong loss gradient."""
    if sign is None:
        return x.abs()
    sign = torch.as_tensor(sign).to(x.dtype).reshape(x.shape)
    own = torch.sign(x.detach())
    differ = sign != own
    L1_INJECTED["elements"] += int(differ.numel())
    L1_INJECTED["differ"] += int(differ.sum())
    if bool(differ.any()):
        worst = float(x.detach().abs()[differ].max())
        assert worst < L1_KINK_BAND, "injected L1 sign pattern differs outside the kink band: |x| = %g" % worst
    return x * sign


def losses(p, out, batch, wr_rate=1e-6, use_l1=True, l1_signs=None):
    """MSTTS_SV.py:127-161 (quirks Q7, Q8, Q19).  l1_signs: optional {"linear", "post"} sign patterns for abs_at."""
    mel = batch["Mel"]
    L = batch["Mel_Length"].long()
    S = int(L.max()) + 1
    stop_target = (torch.arange(S)[None, :] >= L[:, None]).to(mel.dtype)
    lin, post = out["Linear"][:, :-1], out["Mel"][:, :-1]
    linear_loss = ((lin - mel) ** 2).mean()
    postnet_loss = ((post - mel) ** 2).mean()
    if use_l1:
        linear_loss = linear_loss + abs_at(lin - mel, l1_signs.get("linear") if l1_signs else None).mean()
        postnet_loss = postnet_loss + abs_at(post - mel, l1_signs.get("post") if l1_signs else None).mean()
    z = out["Stop_Logit"]
    stop_loss = (torch.clamp(z, min=0) - z * stop_target + torch.log1p(torch.exp(-z.abs()))).mean()
    wr = wr_rate * sum((p[k] ** 2).sum() / 2 for k in p if M.in_weight_reg(k))
    return {"Loss": linear_loss + postnet_loss + stop_loss + wr, "Linear_Loss": linear_loss,
            "Postnet_Loss": postnet_loss, "Stop_Loss": stop_loss, "Weight_Regularization_Loss": wr}


def adam_tf(param, grad, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-6):
    """tf.train.AdamOptimizer (quirk Q18): epsilon outside the bias correction.  t counts from 1."""
    m = b1 * m + (1 - b1) * grad
    v = b2 * v + (1 - b2) * grad * grad
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return param - lr_t * m / (v.sqrt() + eps), m, v


def train_step(params, opt_state, d, batch, masks, global_step, dtype=torch.float64, update_vocoder_bn=True,
               return_grads=False):
    """One Tacotron2.Train iteration (MSTTS_SV.py:268-273).  params: name->np/torch; opt_state:
    {'m':{},'v':{}} or None.  Returns (new_params, new_opt_state, scalars[, grads, outputs])."""
    p = {k: (v.detach().clone().to(dtype) if torch.is_tensor(v) else torch.tensor(np.asarray(v), dtype=dtype))
         for k, v in params.items()}
    names = [k for k in p if M.is_trainable(k)]
    for k in names:
        p[k].requires_grad_(True)
    bt = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in batch.items()}
    stats = {}
    out = M.forward(p, d, bt, True, masks, stats_out=stats, with_vocoder=False)
    if update_vocoder_bn:      # quirk Q20: the vocoder conv-bank's BN update ops ride along
        with torch.no_grad():
            M.taco1_convbank(p, d, out["Mel"].detach(), True, stats)
    l1 = {k[len("l1_sign_"):]: v for k, v in masks.items() if k.startswith("l1_sign_")} if masks else None
    ls = losses(p, out, bt, l1_signs=l1 or None)
    grads = torch.autograd.grad(ls["Loss"], [p[k] for k in names], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(p[k])) for k, g in zip(names, grads)}
    lr = learning_rate(global_step)
    if opt_state is None:
        opt_state = {"m": {k: torch.zeros_like(p[k]) for k in names}, "v": {k: torch.zeros_like(p[k]) for k in names}}
    new_p, new_m, new_v = {}, {}, {}
    with torch.no_grad():
        for k in p:
            if k in grads:
                new_p[k], new_m[k], new_v[k] = adam_tf(p[k].detach(), grads[k], opt_state["m"][k].to(dtype),
                                                      opt_state["v"][k].to(dtype), global_step + 1, lr)
            elif k in stats:
                new_p[k] = stats[k]
            else:
                new_p[k] = p[k].detach()
    scalars = {k: float(v.detach()) for k, v in ls.items()}
    scalars["Learning_Rate"] = lr
    scalars["Global_Step"] = global_step
    ret = (new_p, {"m": new_m, "v": new_v}, scalars)
    if return_grads:
        ret = ret + (grads, {k: v.detach() for k, v in out.items()})
    return ret


def synthetic_batch(d: M.Dims, B, T_enc, L, seed=1234, rank=0, ragged=False):
    """SURVEY 8(d) synthetic inputs (fixed length unless ragged)."""
    g = np.random.default_rng(seed + rank)
    tok = g.integers(2, d.n_tok, size=(B, T_enc)).astype(np.int32)
    tl = np.full(B, T_enc, np.int32)
    ml = np.full(B, L, np.int32)
    if ragged:
        tl = g.integers(max(2, T_enc // 2), T_enc + 1, size=B).astype(np.int32); tl[0] = T_enc
        ml = g.integers(max(1, L // 2), L + 1, size=B).astype(np.int32); ml[0] = L
    for b in range(B):
        tok[b, 0] = 0
        tok[b, tl[b] - 1] = 1
        tok[b, tl[b]:] = 1
    mel = np.clip(g.normal(0, 1.5, size=(B, L, d.n_mel)), -4, 4).astype(np.float32)
    for b in range(B):
        mel[b, ml[b]:] = 0
    spk = g.normal(0, 1, size=(B, d.spk))
    spk = (spk / np.sqrt((spk ** 2).sum())).astype(np.float32)
    return {"Token": torch.from_numpy(tok), "Token_Length": torch.from_numpy(tl),
            "Mel": torch.from_numpy(mel), "Mel_Length": torch.from_numpy(ml),
            "Speaker_Embedding": torch.from_numpy(spk)}


# ---- Taco1 mel -> spectrogram trainer (Taco1_Mel_to_Spect/Taco1_Mel_to_Spect.py:24-100) ------------------------------------------
def taco1_in_weight_reg(name):
    """:45-53: every trainable variable whose lower-cased name has none of 'bias', 'lstm', 'rnn'."""
    low = name.lower()
    return not any(s in low for s in ("bias", "lstm", "rnn"))


def taco1_train_step(params, opt_state, d, mel, spectrogram, masks, global_step, dtype=torch.float64, wr_rate=1e-6,
                     lr_kw=None, return_grads=False):
    """One Mel_to_Spect.Train iteration: loss = mean|pred - spectrogram| (tf.losses.absolute_difference, :58) + wr_rate * sum
    l2_loss; TF-Adam (eps 1e-6) with the :63-69 learning-rate schedule; BN moving statistics ride along (UPDATE_OPS :80)."""
    p = {k: (v.detach().clone().to(dtype) if torch.is_tensor(v) else torch.tensor(np.asarray(v), dtype=dtype)) for k, v in params.items()}
    names = [k for k in p if k.startswith(M.P_V) and not k.endswith(("moving_mean", "moving_variance"))]
    for k in names:
        p[k].requires_grad_(True)
    stats = {}
    pred = M.taco1_forward(p, d, mel.to(dtype), True, masks, stats)
    l1 = (pred - spectrogram.to(dtype)).abs().mean()
    wr = wr_rate * sum(0.5 * (p[k] ** 2).sum() for k in names if taco1_in_weight_reg(k))
    loss = l1 + wr
    grads = dict(zip(names, torch.autograd.grad(loss, [p[k] for k in names])))
    lr = learning_rate(global_step, **(lr_kw or dict(initial=1e-3, minimum=1e-5, decay_start=50000, decay_step=100, decay_rate=0.5)))
    if opt_state is None:
        opt_state = {"m": {k: torch.zeros_like(p[k]) for k in names}, "v": {k: torch.zeros_like(p[k]) for k in names}}
    new_p, new_m, new_v = {}, {}, {}
    with torch.no_grad():
        for k in p:
            if k in grads:
                new_p[k], new_m[k], new_v[k] = adam_tf(p[k].detach(), grads[k], opt_state["m"][k].to(dtype), opt_state["v"][k].to(dtype), global_step + 1, lr)
            elif k in stats:
                new_p[k] = stats[k]
            else:
                new_p[k] = p[k].detach()
    sc = {"Loss": float(loss.detach()), "L1_Loss": float(l1.detach()), "Weight_Regularization_Loss": float(wr.detach()), "Learning_Rate": lr}
    ret = (new_p, {"m": new_m, "v": new_v}, sc)
    if return_grads:
        ret = ret + (grads, pred.detach())
    return ret


# ---- GE2E speaker-encoder trainer (Speaker_Embedding/Speaker_Embedding.py:27-80, Modules.py:6-98) -----------------------------
def speaker_stack(p, d, mel, training, masks=None):
    """Restructure + Stack_LSTM (Modules.py:6-37): dense 80 -> 256, three zoneout LSTMs, residual wrappers on cells 0 and 1.
    Returns the outputs [N, T, spk]."""
    masks = masks or {}
    x = mel @ p[M.P_S + "dense/kernel"] + p[M.P_S + "dense/bias"]
    for i in range(d.spk_lstm_n):
        pre = M.P_S + "lstm/rnn/multi_rnn_cell/cell_%d/lstmcell_%d/" % (i, i)
        x = M.run_lstm(x, None, p[pre + "kernel"], p[pre + "bias"], d.spk_lstm, masks.get("s_zc_%d" % i), masks.get("s_zh_%d" % i),
                       d.zoneout, training, residual=(i < d.spk_lstm_n - 1))
    return x


def ge2e_loss(x_last, P, w, b):
    """Embedding_Generate + Loss, 'Softmax' method (Modules.py:39-98).  x_last [S*P, D], speaker-major."""
    e = x_last * torch.rsqrt(torch.clamp((x_last * x_last).sum(dim=1, keepdim=True), min=1e-12))
    N, D = e.shape
    S = N // P
    r = e.reshape(S, P, D)
    tot = r.sum(dim=1, keepdim=True)
    cw = (tot - r) / (P - 1)
    cb = r.mean(dim=1)
    cos = lambda a, c: (a * c).sum(-1) / (torch.sqrt((a ** 2).sum(-1)) * torch.sqrt((c ** 2).sum(-1)))
    within = w * cos(r, cw) - b                                                                   # [S,P]
    tx, ty = e[:, None, :], cb[None, :, :]
    cos2 = (ty * tx).sum(2) / (torch.sqrt((ty ** 2).sum(2)) * torch.sqrt((tx ** 2).sum(2)) + 1e-8)   # [N,S]
    between = (w * cos2 - b).reshape(S, P, S)
    keep = ~torch.eye(S, dtype=torch.bool)[:, None, :].expand(S, P, S)
    between = between[keep].reshape(S, P, S - 1)
    logits = torch.cat([within[..., None], between], dim=-1)
    return -(torch.log_softmax(logits, dim=-1)[..., 0]).mean()


def speaker_train_step(params, loss_vars, opt_state, d, mel, P, masks, global_step, dtype=torch.float64, lr_kw=None, return_grads=False):
    """One Speaker_Embedding.Train iteration: plain Adam(eps 1e-8).minimize (the clipped train op of :62-66 is overwritten by
    :69-72), learning rate = max(exponential_decay, Min) (:46-52).  loss_vars = {'loss/weight': 10, 'loss/bias': -5}."""
    p = {k: (v.detach().clone().to(dtype) if torch.is_tensor(v) else torch.tensor(np.asarray(v), dtype=dtype)) for k, v in params.items()}
    lv = {k: torch.tensor(float(v), dtype=dtype, requires_grad=True) for k, v in loss_vars.items()}
    names = [k for k in p if k.startswith(M.P_S)]
    for k in names:
        p[k].requires_grad_(True)
    out = speaker_stack(p, d, mel.to(dtype), True, masks)
    loss = ge2e_loss(out[:, -1, :], P, lv["loss/weight"], lv["loss/bias"])
    allv = [p[k] for k in names] + [lv["loss/weight"], lv["loss/bias"]]
    g = torch.autograd.grad(loss, allv, allow_unused=True)
    grads = {k: (gg if gg is not None else torch.zeros_like(v)) for k, gg, v in zip(names + ["loss/weight", "loss/bias"], g, allv)}
    kw = lr_kw or dict(initial=1e-3, minimum=1e-5, decay_step=10000, decay_rate=0.5)
    lr = max(kw["initial"] * kw["decay_rate"] ** (global_step / kw["decay_step"]), kw["minimum"])
    if opt_state is None:
        opt_state = {"m": {k: torch.zeros_like(v) for k, v in grads.items()}, "v": {k: torch.zeros_like(v) for k, v in grads.items()}}
    new_p, new_lv, new_m, new_v = dict(p), {}, {}, {}
    with torch.no_grad():
        for k in names:
            new_p[k], new_m[k], new_v[k] = adam_tf(p[k].detach(), grads[k], opt_state["m"][k].to(dtype), opt_state["v"][k].to(dtype), global_step + 1, lr, eps=1e-8)
        for k in lv:
            new_lv[k], new_m[k], new_v[k] = adam_tf(lv[k].detach(), grads[k], opt_state["m"][k].to(dtype), opt_state["v"][k].to(dtype), global_step + 1, lr, eps=1e-8)
        new_p = {k: v.detach() for k, v in new_p.items()}
    ret = (new_p, {k: float(v) for k, v in new_lv.items()}, {"m": new_m, "v": new_v}, {"Loss": float(loss.detach()), "Learning_Rate": lr})
    if return_grads:
        ret = ret + (grads, out.detach())
    return ret
