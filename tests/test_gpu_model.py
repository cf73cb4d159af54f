"""Parity of the recurrent kernels, the attention step and the whole train step against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from multi_speaker_tts_amd import lib
from multi_speaker_tts_amd.engine import TrainEngine
from multi_speaker_tts_amd.params import LSA, Dims
from oracle import model as OM, train as OT, audio as OA
from tests.helpers import dims_pair, rel_err, t2n, to_dev

pytestmark = pytest.mark.gpu


def _r(dev, *shape, seed=0, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.tensor(g.normal(0, scale, size=shape), dtype=torch.float32, device=dev)


def test_lstm_point_fwd_bwd(dev):
    """Fused zoneout cell (pointwise part) forward/backward vs torch-fp64 autograd of the oracle cell."""
    B, H = 5, 24
    gates = _r(dev, B, 4 * H, seed=1); bias = _r(dev, 4 * H, seed=2, scale=0.1)
    cp, hp = _r(dev, B, H, seed=3), _r(dev, B, H, seed=4)
    zc = torch.tensor(np.random.default_rng(5).integers(0, 2, (B, H)).astype(np.uint8), device=dev)
    zh = torch.tensor(np.random.default_rng(6).integers(0, 2, (B, H)).astype(np.uint8), device=dev)
    out, cn, hn = torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev)
    acts, craw = torch.zeros(B, 4 * H, device=dev), torch.zeros(B, H, device=dev)
    d = lib.LstmPointFwd()
    d.B, d.H, d.gates_h, d.bias = B, H, lib.ptr(gates), lib.ptr(bias)
    d.c_prev, d.h_prev, d.zc, d.zh, d.zoneout = lib.ptr(cp), lib.ptr(hp), lib.ptr(zc), lib.ptr(zh), 0.1
    d.out, d.out_sb, d.c_next, d.h_next, d.acts_out, d.c_raw = lib.ptr(out), H, lib.ptr(cn), lib.ptr(hn), lib.ptr(acts), lib.ptr(craw)
    lib.call("mstts_lstm_point_fwd", C.byref(d))
    g64 = (gates.double().cpu() + bias.double().cpu()).requires_grad_(True)
    cp64, hp64 = cp.double().cpu().requires_grad_(True), hp.double().cpu().requires_grad_(True)
    # oracle cell with identity kernel: feed pre-activations through x, zero recurrent part
    i, j, f, o = g64.chunk(4, 1)
    c = torch.sigmoid(f + 1.0) * cp64 + torch.sigmoid(i) * torch.tanh(j)
    m = torch.sigmoid(o) * torch.tanh(c)
    c2 = 0.9 * (c - cp64) * zc.cpu().double() + cp64
    h2 = 0.9 * (m - hp64) * zh.cpu().double() + hp64
    assert rel_err(t2n(out), t2n(m)) < 1e-5 and rel_err(t2n(cn), t2n(c2)) < 1e-5 and rel_err(t2n(hn), t2n(h2)) < 1e-5
    dm, dc2, dh2 = _r(dev, B, H, seed=7), _r(dev, B, H, seed=8), _r(dev, B, H, seed=9)
    ((m * dm.double().cpu()).sum() + (c2 * dc2.double().cpu()).sum() + (h2 * dh2.double().cpu()).sum()).backward()
    dg, dcp, dhp = torch.zeros(B, 4 * H, device=dev), torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev)
    b = lib.LstmPointBwd()
    b.B, b.H, b.d_out, b.dout_sb = B, H, lib.ptr(dm), H
    b.d_c_state, b.d_h_state, b.acts, b.c_raw, b.c_prev = lib.ptr(dc2), lib.ptr(dh2), lib.ptr(acts), lib.ptr(craw), lib.ptr(cp)
    b.zc, b.zh, b.zoneout, b.dgates, b.d_c_prev, b.d_h_prev = lib.ptr(zc), lib.ptr(zh), 0.1, lib.ptr(dg), lib.ptr(dcp), lib.ptr(dhp)
    lib.call("mstts_lstm_point_bwd", C.byref(b))
    assert rel_err(t2n(dg), t2n(g64.grad)) < 2e-5
    assert rel_err(t2n(dcp), t2n(cp64.grad)) < 2e-5 and rel_err(t2n(dhp), t2n(hp64.grad)) < 2e-5


@pytest.mark.parametrize("B,T,M,KS", [(3, 37, 48, 31), (2, 128, 768, 31), (1, 16, 16, 5), (4, 300, 768, 31), (33, 128, 100, 7)])
def test_lsa_step_fwd_bwd(dev, B, T, M, KS):
    """One attention step (energy+context kernels, dalign+denergy kernels) vs the oracle's lsa_step."""
    A, CH, Hq = 128, 32, 40
    od = OM.Dims(att_k=KS, dec_lstm=Hq)
    g = np.random.default_rng(3)
    p = {LSA + "query_layer/kernel": g.normal(0, 0.2, (Hq, A)),
         LSA + "attention_convolution_dense_layer/conv1d/kernel": g.normal(0, 0.3, (KS, 1, CH)),
         LSA + "attention_convolution_dense_layer/conv1d/bias": g.normal(0, 0.1, (CH,)),
         LSA + "attention_convolution_dense_layer/dense/kernel": g.normal(0, 0.3, (CH, A)),
         LSA + "score_layer/weight_w": g.normal(0, 0.5, (1, 1, A)), LSA + "score_layer/bias_b": g.normal(0, 0.1, (1, 1, A))}
    lengths = np.array([T] + list(g.integers(max(1, T // 2), T + 1, B - 1)), np.int32)
    mask = np.arange(T)[None, :] < lengths[:, None]
    keys = g.normal(0, 1, (B, T, A)) * mask[:, :, None]; values = g.normal(0, 1, (B, T, M)) * mask[:, :, None]
    query = g.normal(0, 1, (B, Hq)); cum = np.abs(g.normal(0, 0.5, (B, T))) * mask
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    kt, vt = torch.tensor(keys, requires_grad=True), torch.tensor(values, requires_grad=True)
    qt, ct = torch.tensor(query, requires_grad=True), torch.tensor(cum, requires_grad=True)
    align, cum_next, ctx = OM.lsa_step(pt, od, kt, vt, torch.tensor(mask), qt, ct)
    f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev).contiguous()
    dk, dv, dl = f32(keys), f32(values), torch.tensor(lengths, device=dev)
    dp = {k: f32(v) for k, v in p.items()}
    c = lib.LsaConst()
    c.B, c.T, c.A, c.M, c.KS, c.CH = B, T, A, M, KS, CH
    c.keys, c.values, c.lengths = lib.ptr(dk), lib.ptr(dv), lib.ptr(dl)
    c.conv_k, c.conv_b = lib.ptr(dp[LSA + "attention_convolution_dense_layer/conv1d/kernel"]), lib.ptr(dp[LSA + "attention_convolution_dense_layer/conv1d/bias"])
    c.dense_k, c.score_w, c.score_b = lib.ptr(dp[LSA + "attention_convolution_dense_layer/dense/kernel"]), lib.ptr(dp[LSA + "score_layer/weight_w"]), lib.ptr(dp[LSA + "score_layer/bias_b"])
    loc_k, loc_b = torch.zeros(KS, A, device=dev), torch.zeros(A, device=dev)
    lib.call("mstts_lsa_fold_location", c.conv_k, c.conv_b, c.dense_k, lib.ptr(loc_k), lib.ptr(loc_b), KS, CH, A)
    c.loc_k, c.loc_b = lib.ptr(loc_k), lib.ptr(loc_b)
    loc_kt = torch.full((A, 36), 7.0, device=dev)              # the by-unit copy the single-launch forward step prefers
    lib.call("mstts_lsa_filter_by_unit", lib.ptr(loc_k), lib.ptr(loc_kt), KS, A)
    assert torch.equal(loc_kt[:, :KS], loc_k.t()) and float(loc_kt[:, KS:].abs().max()) == 0.0
    c.loc_kt = lib.ptr(loc_kt)
    wc = p[LSA + "attention_convolution_dense_layer/conv1d/kernel"][:, 0, :]
    assert rel_err(t2n(loc_k), wc @ p[LSA + "attention_convolution_dense_layer/dense/kernel"]) < 2e-6
    q = f32(query @ p[LSA + "query_layer/kernel"]); dcum = f32(cum)
    en, al, cn, cx = torch.zeros(B, T, device=dev), torch.zeros(B, T, device=dev), torch.zeros(B, T, device=dev), torch.zeros(B, M, device=dev)
    lib.call("mstts_lsa_energy_fwd", C.byref(c), lib.ptr(q), 1, 0, None, lib.ptr(dcum), lib.ptr(en))
    lib.call("mstts_lsa_context_fwd", C.byref(c), lib.ptr(en), lib.ptr(dcum), lib.ptr(al), lib.ptr(cn), lib.ptr(cx), M, None, 0)
    assert rel_err(t2n(al), t2n(align)) < 2e-5 and rel_err(t2n(cn), t2n(cum_next)) < 2e-5 and rel_err(t2n(cx), t2n(ctx)) < 2e-5
    # single-launch form (in-launch energy exchange): same outputs, written straight into strided rows
    gran = torch.zeros(int(lib.load().mstts_lsa_step_ws_bytes(B, T)) // 8, dtype=torch.int64, device=dev)
    al2, cn2, cx2, cx3 = torch.zeros(B, T, device=dev), torch.zeros(B, T, device=dev), torch.zeros(B, M + 4, device=dev), torch.zeros(B, M, device=dev)
    qs = torch.zeros(B, A, device=dev)
    q3 = torch.stack([0.25 * q, 0.5 * q, 0.25 * q]).contiguous()            # the query arrives as split-K partial slabs
    lib.call("mstts_lsa_step_fwd", C.byref(c), lib.ptr(q3), 3, B * A, lib.ptr(qs), lib.ptr(dcum), lib.ptr(al2), lib.ptr(cn2),
             lib.ptr(cx2), M + 4, lib.ptr(cx3), M, None, lib.ptr(gran), 7)
    assert rel_err(t2n(al2), t2n(align)) < 2e-5 and rel_err(t2n(cn2), t2n(cum_next)) < 2e-5 and rel_err(t2n(cx2[:, :M]), t2n(ctx)) < 2e-5
    assert torch.equal(cx2[:, :M], cx3) and float(cx2[:, M:].abs().max()) == 0.0 and rel_err(t2n(qs), t2n(q)) < 1e-6
    assert int(gran[-1]) == 0 and bool(((gran[:B * T] >> 32) == 7).all())      # no time-outs; every granule carries this epoch
    # backward: upstream grads on ctx and on the next cumulative state.  The latter arrives as G_next plus the
    # filter-transpose of the next step's h (G[t] = G_next[t] + sum_j h_next[t+pad-j][j]).
    pad = (KS - 1) // 2
    d_ctx = g.normal(0, 1, (B, M)); G_next = g.normal(0, 1, (B, T)); h_next = g.normal(0, 1, (B, T, 32))
    G_ref = G_next.copy()
    for j in range(KS):
        for t_ in range(T):
            tau = t_ + pad - j
            if 0 <= tau < T:
                G_ref[:, t_] += h_next[:, tau, j]
    obj = (ctx * torch.tensor(d_ctx)).sum() + (cum_next * torch.tensor(G_ref)).sum()
    obj.backward()
    G, da = torch.zeros(B, T, device=dev), torch.zeros(B, T, device=dev)
    d_ctx_d, G_next_d, h_next_d = f32(d_ctx), f32(G_next), f32(h_next)     # keep alive: the calls are asynchronous
    lib.call("mstts_lsa_dalign_bwd", C.byref(c), lib.ptr(d_ctx_d), M, None, 0, 0, 0, lib.ptr(G_next_d), lib.ptr(h_next_d), lib.ptr(G), lib.ptr(da))
    assert rel_err(t2n(G), G_ref) < 2e-5
    de, dq, hh = torch.zeros(B, T, device=dev), torch.zeros(B, A, device=dev), torch.zeros(B, T, 32, device=dev)
    lib.call("mstts_lsa_denergy_bwd", C.byref(c), lib.ptr(al), lib.ptr(da), lib.ptr(q), lib.ptr(dcum), lib.ptr(de), lib.ptr(dq), lib.ptr(hh))
    # single-launch form of the two calls above (row-wide dot(a, d_a) exchanged inside the launch)
    G2, de2, dq2, hh2 = torch.zeros(B, T, device=dev), torch.zeros(B, T, device=dev), torch.zeros(B, A, device=dev), torch.zeros(B, T, 32, device=dev)
    ctx_f = f32(t2n(ctx))                                   # this step's forward context: dot(a, d_a) = dot(a, G) + ctx . d_ctx inside the kernel
    lib.call("mstts_lsa_step_bwd", C.byref(c), lib.ptr(d_ctx_d), M, None, 0, 0, 0, lib.ptr(G_next_d), lib.ptr(h_next_d), lib.ptr(G2),
             lib.ptr(al), lib.ptr(q), lib.ptr(dcum), lib.ptr(ctx_f), M, lib.ptr(de2), lib.ptr(dq2), lib.ptr(hh2))
    assert rel_err(t2n(G2), t2n(G)) < 1e-6
    assert rel_err(t2n(de2), t2n(de)) < 1e-5 and rel_err(t2n(dq2), t2n(dq)) < 1e-5 and rel_err(t2n(hh2), t2n(hh)) < 1e-5
    if (B, T, M) == (2, 128, 768):
        # time-out path of the forward kernel (never taken on a healthy chip): remove one slice's workgroups; the rest of each row waits
        # out its bounded spin, recomputes the missing energies serially and must still produce the same results; the counter reads > 0
        skip = 3
        al3, cn3, cx3b = torch.zeros(B, T, device=dev), torch.zeros(B, T, device=dev), torch.zeros(B, M, device=dev)
        gran3 = torch.zeros_like(gran)
        lib.call("mstts_lsa_step_fwd_selftest", C.byref(c), lib.ptr(q3), 3, B * A, None, lib.ptr(dcum), lib.ptr(al3), lib.ptr(cn3), lib.ptr(cx3b), M,
                 lib.ptr(gran3), 9, skip)
        keep_t = np.ones(T, bool); keep_t[16 * skip:16 * skip + 16] = False
        keep_m = np.ones(M, bool); keep_m[96 * skip:96 * skip + 96] = False
        assert int(gran3[-1]) > 0
        assert rel_err(t2n(al3)[:, keep_t], t2n(align)[:, keep_t]) < 2e-5 and rel_err(t2n(cx3b)[:, keep_m], t2n(ctx)[:, keep_m]) < 2e-5
        assert float(al3[:, ~torch.tensor(keep_t)].abs().max()) == 0.0             # the removed slice wrote nothing
    dquery = t2n(dq).astype(np.float64) @ p[LSA + "query_layer/kernel"].T
    assert rel_err(dquery, t2n(qt.grad)) < 5e-5
    # grad wrt cum = G (carried) + filter-transpose of h
    hn = t2n(hh).astype(np.float64)
    dcum_ref = t2n(G).astype(np.float64).copy()
    for j in range(KS):
        for t_ in range(T):
            tau = t_ + pad - j
            if 0 <= tau < T:
                dcum_ref[:, t_] += hn[:, tau, j]
    assert rel_err(dcum_ref, t2n(ct.grad)) < 5e-5
    # parameter gradients + d_keys through the post-loop kernel with S = 1, then unfolded to the reference variables
    dkeys = torch.zeros(B, T, A, device=dev)
    dlk, gw, gsb = torch.zeros(KS, A, device=dev), torch.zeros(A, device=dev), torch.zeros(A, device=dev)
    ws = torch.empty(int(lib.load().mstts_lsa_param_bwd_ws_floats(B, T, 1)) + 2, device=dev) if B % 2 else None      # both accumulation forms: fp64 block reduction / fp32 atomics
    lib.call("mstts_lsa_param_bwd", C.byref(c), 1, lib.ptr(q), lib.ptr(dcum), lib.ptr(de), lib.ptr(dkeys), lib.ptr(dlk), lib.ptr(gw), lib.ptr(gsb), lib.ptr(ws))
    gk, gb, gd = torch.zeros(KS, CH, device=dev), torch.zeros(CH, device=dev), torch.zeros(CH, A, device=dev)
    lib.call("mstts_lsa_unfold_location_grad", c.conv_k, c.conv_b, c.dense_k, lib.ptr(dlk), lib.ptr(gsb), lib.ptr(gk), lib.ptr(gb), lib.ptr(gd), KS, CH, A)
    assert rel_err(t2n(dkeys), t2n(kt.grad)) < 5e-5
    assert rel_err(t2n(gd), t2n(pt[LSA + "attention_convolution_dense_layer/dense/kernel"].grad)) < 5e-5
    assert rel_err(t2n(gw), t2n(pt[LSA + "score_layer/weight_w"].grad).reshape(-1)) < 5e-5
    assert rel_err(t2n(gsb), t2n(pt[LSA + "score_layer/bias_b"].grad).reshape(-1)) < 5e-5
    assert rel_err(t2n(gk), t2n(pt[LSA + "attention_convolution_dense_layer/conv1d/kernel"].grad)[:, 0, :]) < 1e-4
    assert rel_err(t2n(gb), t2n(pt[LSA + "attention_convolution_dense_layer/conv1d/bias"].grad)) < 1e-4
    # d_values from the context: outer(align, d_ctx)
    assert rel_err(t2n(al)[:, :, None] * d_ctx[:, None, :], t2n(vt.grad)) < 5e-5


@pytest.mark.parametrize("B,T,bf16", [(5, 128, 0), (32, 113, 0), (3, 40, 1), (4, 150, 0)])
def test_lsa_step_fwd_q(dev, B, T, bf16):
    """mstts_lsa_step_fwd_q: the query projection q = m1 . Wq computed and exchanged inside the attention launch.  Against the plain
    single-launch step fed with the same query (fp64 product, or bf16-rounded operands in the config-3 form), over several steps
    (distinct epochs on one granule buffer), and once in the self-test form where one slice is missing and the others must recompute
    its query units and energies."""
    A, CH, KS, M, H = 128, 32, 31, 768, 1024
    L = lib.load()
    assert L.mstts_lsa_step_q_supported(T, M, H) == 1 and L.mstts_lsa_step_q_supported(64, 96, H) == 0 and L.mstts_lsa_step_q_supported(T, M, 512) == 0
    g = np.random.default_rng(17)
    f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev).contiguous()
    lengths = np.array([T] + list(g.integers(max(1, T // 2), T + 1, B - 1)), np.int32)
    mask = np.arange(T)[None, :] < lengths[:, None]
    dk, dv = f32(g.normal(0, 1, (B, T, A)) * mask[:, :, None]), f32(g.normal(0, 1, (B, T, M)) * mask[:, :, None])
    conv_k, conv_b, dense_k = f32(g.normal(0, 0.3, (KS, 1, CH))), f32(g.normal(0, 0.1, (CH,))), f32(g.normal(0, 0.3, (CH, A)))
    score_w, score_b = f32(g.normal(0, 0.5, (A,))), f32(g.normal(0, 0.1, (A,)))
    dl = torch.tensor(lengths, device=dev)
    c = lib.LsaConst()
    c.B, c.T, c.A, c.M, c.KS, c.CH = B, T, A, M, KS, CH
    c.keys, c.values, c.lengths = lib.ptr(dk), lib.ptr(dv), lib.ptr(dl)
    c.conv_k, c.conv_b, c.dense_k, c.score_w, c.score_b = lib.ptr(conv_k), lib.ptr(conv_b), lib.ptr(dense_k), lib.ptr(score_w), lib.ptr(score_b)
    loc_k, loc_b, loc_kt = torch.zeros(KS, A, device=dev), torch.zeros(A, device=dev), torch.zeros(A, 36, device=dev)
    lib.call("mstts_lsa_fold_location", c.conv_k, c.conv_b, c.dense_k, lib.ptr(loc_k), lib.ptr(loc_b), KS, CH, A)
    lib.call("mstts_lsa_filter_by_unit", lib.ptr(loc_k), lib.ptr(loc_kt), KS, A)
    c.loc_k, c.loc_b, c.loc_kt = lib.ptr(loc_k), lib.ptr(loc_b), lib.ptr(loc_kt)
    wq = f32(g.normal(0, 0.05, (H, A)))
    WP = H + M                                                   # m1 sits in rows [m1 | ctx] like the decoder's projection input
    gran_q = torch.zeros(int(L.mstts_lsa_step_q_ws_bytes(B, T)) // 8, dtype=torch.int64, device=dev)
    gran = torch.zeros(int(L.mstts_lsa_step_ws_bytes(B, T)) // 8, dtype=torch.int64, device=dev)
    for step in range(1, 4):
        pj = f32(g.normal(0, 1, (B, WP)))
        cum = f32(np.abs(g.normal(0, 0.5, (B, T))) * mask)
        if bf16:
            qref = (pj[:, :H].to(torch.bfloat16).double() @ wq.to(torch.bfloat16).double()).float()
        else:
            qref = (pj[:, :H].double() @ wq.double()).float()
        al, cn, cx, qs = (torch.zeros(B, T, device=dev), torch.zeros(B, T, device=dev), torch.zeros(B, M, device=dev), torch.zeros(B, A, device=dev))
        lib.call("mstts_lsa_step_fwd", C.byref(c), lib.ptr(qref.contiguous()), 1, 0, None, lib.ptr(cum), lib.ptr(al), lib.ptr(cn), lib.ptr(cx), M, None, 0,
                 None, lib.ptr(gran), step)
        al2, cn2, cx2 = torch.zeros(B, T, device=dev), torch.zeros(B, T, device=dev), torch.zeros(B, M, device=dev)
        lib.call("mstts_lsa_step_fwd_q", C.byref(c), lib.ptr(pj), WP, lib.ptr(wq), H, bf16, lib.ptr(qs), lib.ptr(cum), lib.ptr(al2), lib.ptr(cn2),
                 lib.ptr(cx2), M, None, 0, None, lib.ptr(gran_q), step, -1)
        torch.cuda.synchronize()
        assert rel_err(t2n(qs), t2n(qref)) < 3e-6
        assert rel_err(t2n(al2), t2n(al)) < 1e-5 and rel_err(t2n(cn2), t2n(cn)) < 1e-5 and rel_err(t2n(cx2), t2n(cx)) < 1e-5
        assert int(gran_q[B * T]) == 0                                                  # no time-outs
        assert bool(((gran_q[B * T + 1:] >> 32) == step).all())                         # every query granule carries this epoch
    # self-test: slice 3 never runs; every other workgroup times out on its 16 query units and its energies and recomputes them
    al3, cn3, cx3 = torch.zeros(B, T, device=dev), torch.zeros(B, T, device=dev), torch.zeros(B, M, device=dev)
    lib.call("mstts_lsa_step_fwd_q", C.byref(c), lib.ptr(pj), WP, lib.ptr(wq), H, bf16, None, lib.ptr(cum), lib.ptr(al3), lib.ptr(cn3),
             lib.ptr(cx3), M, None, 0, None, lib.ptr(gran_q), 9, 3)
    torch.cuda.synchronize()
    assert int(gran_q[B * T]) > 0
    ncs = max(-(-T // 16), -(-M // 96))
    tsl, dsl = -(-T // ncs), -(-(-(-M // ncs)) // 4) * 4
    own = np.ones(T, bool); own[3 * tsl:4 * tsl] = False                                # the missing slice's own outputs are not written
    ownc = np.ones(M, bool); ownc[3 * dsl:4 * dsl] = False
    assert rel_err(t2n(al3)[:, own], t2n(al)[:, own]) < 1e-5 and rel_err(t2n(cx3)[:, ownc], t2n(cx)[:, ownc]) < 1e-5
    if bf16:
        return
    # mstts_lsa_step_fwd_qp: the output projection [m1 | ctx] . Wp + bias out of the same launch (free-running decoder) - the context part
    # from the projected values vp = values . Wp[H:, :], the m1 part on the query projection through the by-owner packed kernel rows;
    # against the fp64 product over the context the plain step produced
    NM, NP = 80, 84
    assert L.mstts_lsa_step_qp_supported(T, M, H, NP) == 1 and L.mstts_lsa_step_qp_supported(T, M, H, 92) == 0
    wp, bias = f32(g.normal(0, 0.05, (WP, NP))), f32(g.normal(0, 0.1, (NM + 1,)))
    wp_own = torch.zeros(int(L.mstts_lsa_proj_pack_floats()), device=dev)
    lib.call("mstts_lsa_proj_pack", lib.ptr(wp), NP, H, NP, lib.ptr(wp_own))
    vp = (dv.double().reshape(B * T, M) @ wp[H:].double()).float().contiguous()
    gran_p = torch.zeros(int(L.mstts_lsa_step_qp_ws_bytes(B, T)) // 8, dtype=torch.int64, device=dev)
    # ... and the next step's prenet on that frame (two dense layers, relu, dropout masks as multipliers) in the same launch
    PN, keep_p = 256, 0.5
    assert L.mstts_lsa_step_prenet_supported(PN, NM) == 1 and L.mstts_lsa_step_prenet_supported(128, NM) == 0
    pw0, pb0 = f32(g.normal(0, 0.2, (NM, PN))), f32(g.normal(0, 0.1, (PN,)))
    pw1, pb1 = f32(g.normal(0, 0.1, (PN, PN))), f32(g.normal(0, 0.1, (PN,)))
    pm0 = torch.tensor(g.integers(0, 2, (B, PN)).astype(np.uint8), device=dev)
    pm1 = torch.tensor(g.integers(0, 2, (B, PN)).astype(np.uint8), device=dev)
    pre_ld = PN + 44
    pn = lib.LsaPrenet()
    pn.w0, pn.b0, pn.w1, pn.b1, pn.m0, pn.m1 = lib.ptr(pw0), lib.ptr(pb0), lib.ptr(pw1), lib.ptr(pb1), lib.ptr(pm0), lib.ptr(pm1)
    pn.inv_keep, pn.P, pn.out_ld = 1.0 / keep_p, PN, pre_ld

    def prenet_ref(frame):
        h = torch.relu(frame.double() @ pw0.double() + pb0.double()) * pm0.double() / keep_p
        return torch.relu(h @ pw1.double() + pb1.double()) * pm1.double() / keep_p
    for step in range(1, 4):
        pj = f32(g.normal(0, 1, (B, WP)))
        cum = f32(np.abs(g.normal(0, 0.5, (B, T))) * mask)
        qref = (pj[:, :H].double() @ wq.double()).float()
        al, cn, cx = torch.zeros(B, T, device=dev), torch.zeros(B, T, device=dev), torch.zeros(B, M, device=dev)
        lib.call("mstts_lsa_step_fwd", C.byref(c), lib.ptr(qref.contiguous()), 1, 0, None, lib.ptr(cum), lib.ptr(al), lib.ptr(cn), lib.ptr(cx), M, None, 0,
                 None, lib.ptr(gran), 10 + step)
        ref = torch.cat([pj[:, :H], cx], 1).double() @ wp.double()
        ref[:, :NM + 1] += bias.double()
        al2, cn2, cx2 = torch.zeros(B, T, device=dev), torch.zeros(B, T, device=dev), torch.zeros(B, M, device=dev)
        lin, stop = torch.full((B, NM), 7.0, device=dev), torch.full((B,), 7.0, device=dev)
        pre_out = torch.full((B, pre_ld), 7.0, device=dev)
        pn.out = lib.ptr(pre_out)
        lib.call("mstts_lsa_step_fwd_qp", C.byref(c), lib.ptr(pj), WP, lib.ptr(wq), H, lib.ptr(wp_own), lib.ptr(vp), lib.ptr(bias), NP, NM,
                 lib.ptr(lin), lib.ptr(stop), lib.ptr(cum), lib.ptr(al2), lib.ptr(cn2), lib.ptr(cx2), M, lib.ptr(pj[:, H:]), WP, None,
                 C.byref(pn) if step != 2 else None, lib.ptr(gran_p), step, -1)
        torch.cuda.synchronize()
        if step != 2:
            assert rel_err(t2n(pre_out[:, :PN]), t2n(prenet_ref(ref[:, :NM]))) < 2e-5
            assert float((pre_out[:, PN:] - 7.0).abs().max()) == 0.0
        else:
            assert float((pre_out - 7.0).abs().max()) == 0.0                           # no prenet requested: nothing written
        assert int(gran_p[B * T]) == 0
        assert rel_err(t2n(al2), t2n(al)) < 1e-5 and rel_err(t2n(cn2), t2n(cn)) < 1e-5 and rel_err(t2n(cx2), t2n(cx)) < 1e-5
        assert rel_err(t2n(pj[:, H:]), t2n(cx)) < 1e-5                                  # second context destination (the projection input rows)
        assert rel_err(t2n(lin), t2n(ref[:, :NM])) < 1e-5 and rel_err(t2n(stop), t2n(ref[:, NM])) < 1e-5
    # self-test form: slice 3 missing -> the other owners still produce their outputs (from recomputed energies), outputs 33..43 stay unwritten
    # (the other owners then lack the missing slice's m1 . Wp + b values for their copy of the frame: recomputed, lsa_pm_serial)
    lin3, stop3 = torch.full((B, NM), 7.0, device=dev), torch.full((B,), 7.0, device=dev)
    pre3 = torch.full((B, pre_ld), 7.0, device=dev)
    pn.out = lib.ptr(pre3)
    lib.call("mstts_lsa_step_fwd_qp", C.byref(c), lib.ptr(pj), WP, lib.ptr(wq), H, lib.ptr(wp_own), lib.ptr(vp), lib.ptr(bias), NP, NM,
             lib.ptr(lin3), lib.ptr(stop3), lib.ptr(cum), lib.ptr(al2), lib.ptr(cn2), lib.ptr(cx2), M, None, 0, None, C.byref(pn), lib.ptr(gran_p), 9, 3)
    torch.cuda.synchronize()
    assert int(gran_p[B * T]) > 0
    keep = np.ones(NM, bool); keep[33:44] = False
    assert rel_err(t2n(lin3)[:, keep], t2n(ref[:, :NM])[:, keep]) < 1e-5 and rel_err(t2n(stop3), t2n(ref[:, NM])) < 1e-5
    assert float((lin3[:, 33:44] - 7.0).abs().max()) == 0.0
    keepc = np.ones(PN, bool); keepc[96:128] = False                                    # the missing slice's 32 prenet columns stay unwritten
    assert rel_err(t2n(pre3[:, :PN])[:, keepc], t2n(prenet_ref(ref[:, :NM]))[:, keepc]) < 2e-5
    assert float((pre3[:, 96:128] - 7.0).abs().max()) == 0.0


def test_lsa_step_exchange_under_load(dev):
    """The in-launch energy exchange of mstts_lsa_step_fwd over 300 consecutive epochs with a different query each
    epoch and a weight-streaming kernel interleaved (uneven load, warm L1/L2): every epoch must equal the two-launch
    form on the same inputs (a stale or torn granule would show up as the previous epoch's energies) and no
    workgroup may time out."""
    B, T, M, A, CH, KS, N = 32, 128, 768, 128, 32, 31, 300
    g = torch.Generator(device="cpu").manual_seed(11)
    rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(dev).contiguous()
    keys, values = rn(B, T, A), rn(B, T, M)
    conv_k, conv_b, dense_k, sw, sb_ = rn(KS, 1, CH, sc=0.3), rn(CH, sc=0.1), rn(CH, A, sc=0.3), rn(A, sc=0.5), rn(A, sc=0.1)
    loc_k, loc_b = torch.zeros(KS, A, device=dev), torch.zeros(A, device=dev)
    c = lib.LsaConst()
    c.B, c.T, c.A, c.M, c.KS, c.CH = B, T, A, M, KS, CH
    c.keys, c.values, c.lengths = lib.ptr(keys), lib.ptr(values), None
    c.conv_k, c.conv_b, c.dense_k, c.score_w, c.score_b = lib.ptr(conv_k), lib.ptr(conv_b), lib.ptr(dense_k), lib.ptr(sw), lib.ptr(sb_)
    lib.call("mstts_lsa_fold_location", c.conv_k, c.conv_b, c.dense_k, lib.ptr(loc_k), lib.ptr(loc_b), KS, CH, A)
    c.loc_k, c.loc_b = lib.ptr(loc_k), lib.ptr(loc_b)
    qs = rn(N, B, A)
    cum = torch.zeros(N + 1, B, T, device=dev); cum_r = torch.zeros(N + 1, B, T, device=dev)
    al, al_r = torch.zeros(N, B, T, device=dev), torch.zeros(N, B, T, device=dev)
    cx, cx_r = torch.zeros(N, B, M, device=dev), torch.zeros(N, B, M, device=dev)
    en = torch.zeros(B, T, device=dev)
    gran = torch.zeros(B * T + 1, dtype=torch.int64, device=dev)
    X, W, Pw = rn(32, 1024, sc=0.1), rn(1024, 4096, sc=0.05), torch.zeros(16 * 32 * 4096, device=dev)
    for e in range(N):
        lib.call("mstts_lsa_step_fwd", C.byref(c), lib.ptr(qs[e]), 1, 0, None, lib.ptr(cum[e]), lib.ptr(al[e]), lib.ptr(cum[e + 1]),
                 lib.ptr(cx[e]), M, None, 0, None, lib.ptr(gran), e + 1)
        if e % 3 != 2:      # uneven load between the steps
            lib.call("mstts_skinny_fwd", lib.ptr(X), 1024, lib.ptr(W), 4096, lib.ptr(Pw), 0, 32, 4096, 1024, 4)
    for e in range(N):
        # same inputs per epoch (the recurrence through cum would amplify rounding differences over 300 steps)
        lib.call("mstts_lsa_energy_fwd", C.byref(c), lib.ptr(qs[e]), 1, 0, None, lib.ptr(cum[e]), lib.ptr(en))
        lib.call("mstts_lsa_context_fwd", C.byref(c), lib.ptr(en), lib.ptr(cum[e]), lib.ptr(al_r[e]), lib.ptr(cum_r[e + 1]), lib.ptr(cx_r[e]), M, None, 0)
    torch.cuda.synchronize()
    assert int(gran[-1]) == 0
    worst = max(rel_err(t2n(al[e]), t2n(al_r[e])) for e in range(N))
    assert worst < 1e-4, worst
    assert rel_err(t2n(cx), t2n(cx_r)) < 1e-4 and rel_err(t2n(cum[1:]), t2n(cum_r[1:])) < 1e-4
    # backward kernel, same regime: a different upstream gradient every epoch
    dctx = rn(N, B, M); Gn, hn = rn(B, T), rn(B, T, 32)
    G1, G2 = torch.zeros(B, T, device=dev), torch.zeros(N, B, T, device=dev)
    da = torch.zeros(B, T, device=dev)
    de1, de2 = torch.zeros(N, B, T, device=dev), torch.zeros(N, B, T, device=dev)
    dq1, dq2 = torch.zeros(N, B, A, device=dev), torch.zeros(N, B, A, device=dev)
    h1, h2 = torch.zeros(N, B, T, 32, device=dev), torch.zeros(N, B, T, 32, device=dev)
    for e in range(N):              # (no exchange in the backward kernel: the forward context of the same epoch closes the row-wide dot)
        lib.call("mstts_lsa_step_bwd", C.byref(c), lib.ptr(dctx[e]), M, None, 0, 0, 0, lib.ptr(Gn), lib.ptr(hn), lib.ptr(G2[e]),
                 lib.ptr(al[e]), lib.ptr(qs[e]), lib.ptr(cum[e]), lib.ptr(cx[e]), M, lib.ptr(de2[e]), lib.ptr(dq2[e]), lib.ptr(h2[e]))
        if e % 3 != 2:
            lib.call("mstts_skinny_fwd", lib.ptr(X), 1024, lib.ptr(W), 4096, lib.ptr(Pw), 0, 32, 4096, 1024, 4)
    for e in range(N):
        lib.call("mstts_lsa_dalign_bwd", C.byref(c), lib.ptr(dctx[e]), M, None, 0, 0, 0, lib.ptr(Gn), lib.ptr(hn), lib.ptr(G1), lib.ptr(da))
        lib.call("mstts_lsa_denergy_bwd", C.byref(c), lib.ptr(al[e]), lib.ptr(da), lib.ptr(qs[e]), lib.ptr(cum[e]), lib.ptr(de1[e]), lib.ptr(dq1[e]), lib.ptr(h1[e]))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(de2).all())
    worst = max(rel_err(t2n(de2[e]), t2n(de1[e])) for e in range(N))
    assert worst < 1e-4, worst
    assert rel_err(t2n(dq2), t2n(dq1)) < 1e-4 and rel_err(t2n(h2), t2n(h1)) < 1e-4


def test_stft_mel(dev):
    from multi_speaker_tts_amd import Audio
    g = np.random.default_rng(0)
    t = np.arange(16000 * 2) / 16000.0
    y = (0.3 * np.sin(2 * np.pi * 220 * t) + 0.1 * np.sin(2 * np.pi * 3000 * t) + 0.02 * g.normal(size=t.shape)).astype(np.float32)
    ref = OA.melspectrogram(y)
    for use_fft in (True, False):                  # one-launch FFT-in-LDS path / DFT-as-GEMM path
        got = Audio.melspectrogram(y, 1025, 12.5, 50, 80, 16000, max_abs_value=4, use_fft=use_fft)
        assert got.shape == ref.shape == (80, 1 + len(y) // 200)
        assert np.abs(got - ref).max() < 2e-3, use_fft     # normalised range is [-4, 4]: 5e-4 relative


def test_stft_fft_batch_spectrogram_and_sizes(dev):
    """mstts_stft_fft: several waveforms of different lengths in one launch (each equal to its own single call and to the oracle,
    including the reflect-padded edge frames and a waveform barely longer than the padding), the linear spectrogram output
    (Audio.py:19-22,34-40), and the other power-of-two transform sizes."""
    from multi_speaker_tts_amd import Audio
    g = np.random.default_rng(3)
    wavs = []
    for n in (16000, 1025, 4321, 200 * 37):
        t = np.arange(n) / 16000.0
        wavs.append((0.25 * np.sin(2 * np.pi * (150 + 40 * len(wavs)) * t) + 0.05 * g.normal(size=n)).astype(np.float32))
    outs = Audio.stft_features(wavs, 1025, 12.5, 50, 16000, num_mels=80, max_abs_value=4, want_mel=True, want_spec=True, device=dev)
    for y, (mel, spec) in zip(wavs, outs):
        rs, rm = OA.spectrogram_and_mel(y)
        assert mel.shape == (1 + len(y) // 200, 80) and spec.shape == (1 + len(y) // 200, 1025)
        assert np.abs(t2n(mel).T - rm).max() < 2e-3
        assert np.abs(t2n(spec).T - rs).max() < 1e-3           # [0, 1] range; bins near the -100 dB floor carry the fp32 FFT's noise
        one = Audio.melspectrogram(y, 1025, 12.5, 50, 80, 16000, max_abs_value=4, device=dev)
        assert np.array_equal(one, t2n(mel).T)                  # batched launch == single launch, bit for bit
    s1, m1 = Audio.spectrogram_and_mel(wavs[0], 1025, 12.5, 50, 16000, num_mels=80, max_abs_mels=4, device=dev)
    assert np.array_equal(s1, Audio.spectrogram(wavs[0], 1025, 12.5, 50, 16000, device=dev)) and np.array_equal(m1, t2n(outs[0][0]).T)
    for num_freq, fl in ((257, 25), (513, 50), (2049, 50)):    # n_fft 512 / 1024 / 4096
        got = Audio.melspectrogram(wavs[0], num_freq, 12.5, fl, 40, 16000, max_abs_value=4, device=dev)
        ref = OA.melspectrogram(wavs[0], num_freq, 12.5, fl, 40, 16000, 4)
        assert np.abs(got - ref).max() < 2e-3, num_freq
    with pytest.raises(ValueError):
        Audio.melspectrogram(wavs[0][:1024], 1025, 12.5, 50, 80, 16000, max_abs_value=4, device=dev)
    # the two options the reference's signature carries (Audio.py:29-32,45-46): [0, 1] normalisation and spectral subtraction
    got = Audio.melspectrogram(wavs[2], 1025, 12.5, 50, 80, 16000, max_abs_value=None, device=dev)
    assert np.abs(got - OA.melspectrogram(wavs[2], max_abs_value=None)).max() < 5e-4 and got.min() >= 0.0 and got.max() <= 1.0
    got = Audio.melspectrogram(wavs[2], 1025, 12.5, 50, 80, 16000, max_abs_value=4, spectral_subtract=True, device=dev)
    assert np.abs(got - OA.melspectrogram(wavs[2], spectral_subtract=True)).max() < 2e-3
    sgot = Audio.spectrogram(wavs[2], 1025, 12.5, 50, 16000, spectral_subtract=True, device=dev)
    sref = OA.spectrogram(wavs[2], spectral_subtract=True)
    assert np.mean(np.abs(sgot - sref)) < 1e-4 and np.abs(sgot - sref).max() < 2e-2      # a bin clipped to the floor on one side only moves by a few dB


def _engine_vs_oracle(dev, B, Te, L, ragged, seed, recurrent_dtype=None, l1_inject=True, **dims_kw):
    pd, od = dims_pair(**dims_kw)
    values = OM.init_params(od, seed)
    # make BN / biases non-trivial so every gradient path is exercised
    g = np.random.default_rng(seed + 1)
    for k in values:
        if k.endswith(("bias", "beta", "bias_b")):
            values[k] = g.normal(0, 0.1, values[k].shape)
        if k.endswith("gamma"):
            values[k] = 1.0 + g.normal(0, 0.1, values[k].shape)
    batch = OT.synthetic_batch(od, B, Te, L, seed=seed, ragged=ragged)
    S = L + 1
    masks = OT.make_masks(od, B, Te, S, True, seed=OT.step_seed(1234, 0))
    eng = TrainEngine(pd, device=dev, values=values, recurrent_dtype=recurrent_dtype)
    w = eng.plan(B, Te, L)
    eng.forward(to_dev(batch, dev), w, seed=OT.step_seed(1234, 0))     # masks drawn on the device by Philox
    for name, m in masks.items():                                       # ... must equal the oracle's
        assert np.array_equal(t2n(w.masks[name]), m.numpy()), name
    eng.loss_and_backward(w)
    torch.cuda.synchronize()
    # ReLU pattern of the encoder convolutions, injected into the oracle like the dropout masks (oracle.model.relu_at): with
    # B*T*C ~ 2M pre-activations per layer a handful sit within fp32 rounding of 0 and the gradient is discontinuous there -
    # one such element moved a conv-bias gradient by 12 % of its maximum at B = 32, T = 128.  The oracle asserts that the
    # injected pattern differs from its own only inside +-1e-4.
    omasks = dict(masks)
    for i in range(od.enc_conv_n):
        omasks["relu_enc_%d" % i] = (w.enc_a[i] > 0).reshape(B, Te, od.enc_conv_ch).cpu()
    # ... and the sign pattern of the two L1 terms (MSTTS_SV.py:138-142), the same kind of kink (oracle.train.abs_at): d|x|/dx = sign(x) / n
    # flips for the few |prediction - target| that lie within the fp32 forward error of 0; the oracle asserts the injected pattern differs
    # from its own only inside +-2e-3.  (The HIP loss kernel forms lin - mel from the stored fp32 tensors: one exact fp32 subtraction.)
    if l1_inject and eng.use_l1:
        mel_t = batch["Mel"].to(dev)
        omasks["l1_sign_linear"] = torch.sign(w.linear[:, :-1] - mel_t).cpu()
        omasks["l1_sign_post"] = torch.sign(w.mel_out[:, :-1] - mel_t).cpu()
    w.oracle_masks = omasks
    new_p, opt, sc, grads, out = OT.train_step(values, None, od, batch, omasks, 0, return_grads=True)
    return eng, w, od, values, batch, sc, grads, out, new_p


MID = dict(dec_lstm=64, enc_lstm=32, spk=64, prenet=32)      # shapes on which the skinny K-split kernels are active


# the reference's own layer widths (Hyper_Parameters.py:4-62): the skinny<14>/<16> instantiations, the 84-column padded projection,
# the T=128 single-launch attention geometry and the 512-channel convolutions all meet the oracle end to end in fp32 here; only the
# frozen vocoder / speaker stacks (not on the gradient path) stay reduced
REF = dict(emb=512, enc_conv_ch=512, enc_lstm=256, spk=256, prenet=256, dec_lstm=1024, n_mel=80, post_ch=512)


@pytest.mark.parametrize("B,Te,L,ragged,kw", [(3, 9, 6, False, {}), (4, 21, 13, True, {}), (5, 18, 9, True, MID), (16, 12, 7, True, MID),
                                                (1, 2, 1, False, {}), (2, 140, 3, True, MID),       # smallest batch/lengths; more tokens than one attention pass
                                                (32, 128, 4, False, REF), (8, 40, 12, True, REF),   # fp32 at the reference widths (config-2 geometry / ragged)
                                                (6, 160, 10, True, REF)])                           # ... and a batch padded to 160 tokens (256-position persistent kernels)
def test_train_step_parity(dev, B, Te, L, ragged, kw):
    eng, w, od, values, batch, sc, grads, out, new_p = _engine_vs_oracle(dev, B, Te, L, ragged, seed=11, **kw)
    if kw is REF and eng.persist:     # the reference-width cases run the persistent launches (<= 128 tokens and the 256-position instantiation)
        assert w.persist and w.persist_bwd and eng.persist_fallbacks == 0 and eng.persist_bwd_fallbacks == 0 and eng.non_persistent_plans == 0
    tol = 1e-3    # north_star: within 1e-3 relative on fp32 mels
    assert rel_err(t2n(w.linear), t2n(out["Linear"])) < tol
    assert rel_err(t2n(w.mel_out), t2n(out["Mel"])) < tol
    assert rel_err(t2n(w.stop), t2n(out["Stop_Logit"])) < tol
    assert rel_err(t2n(w.align_hist).transpose(1, 2, 0), t2n(out["Attention_History"])) < tol
    got = eng.scalars(w)
    for k in ("Linear_Loss", "Postnet_Loss", "Stop_Loss", "Weight_Regularization_Loss", "Loss"):
        assert abs(got[k] - sc[k]) <= 1e-4 * max(1.0, abs(sc[k])), (k, got[k], sc[k])
    # gradients (the engine adds the regulariser's gradient inside Adam; add it here for comparison)
    ggot = eng.params.export(grads=True)
    worst = {}
    for k, gr in grads.items():
        ref = t2n(gr).astype(np.float64)
        mine = ggot[k].astype(np.float64)
        if OM.in_weight_reg(k):
            mine = mine + 1e-6 * np.asarray(values[k])
        worst[k] = np.abs(mine - ref).max() / (np.abs(ref).max() + 1e-9)
    bad = {k: v for k, v in worst.items() if v > 5e-3}
    assert not bad, bad
    # optimizer + BN moving statistics
    eng.adam_step()
    torch.cuda.synchronize()
    pgot = eng.params.export()
    bad = {}
    for k, ref in new_p.items():
        if k.startswith("speaker_embedding"):
            continue
        e = rel_err(pgot[k], t2n(ref))
        if e > 2e-3:
            bad[k] = e
    assert not bad, bad


@pytest.mark.parametrize("B,Te,L,kw", [(4, 21, 13, {}), (5, 18, 9, MID)])
def test_train_two_steps_parity(dev, B, Te, L, kw):
    """Two consecutive optimizer steps (TF-Adam slots, bias correction at t = 2, LR schedule, BN moving statistics, derived kernel
    copies refreshed between the steps) against two oracle steps - MSTTS_SV.py:268-273 run twice."""
    pd, od = dims_pair(**kw)
    values = OM.init_params(od, 5)
    g = np.random.default_rng(6)
    for k in values:
        if k.endswith(("bias", "beta", "bias_b")):
            values[k] = g.normal(0, 0.1, values[k].shape)
    eng = TrainEngine(pd, device=dev, values=values, seed=1234)
    p_ref, opt = values, None
    for step in range(2):
        batch = OT.synthetic_batch(od, B, Te, L, seed=20 + step, ragged=True)
        masks = OT.make_masks(od, B, Te, L + 1, True, seed=OT.step_seed(1234, step))
        assert eng.global_step == step
        w = eng.train_step(to_dev(batch, dev))                         # masks drawn from (engine seed, global_step)
        torch.cuda.synchronize()
        omasks = dict(masks)
        for i in range(od.enc_conv_n):
            omasks["relu_enc_%d" % i] = (w.enc_a[i] > 0).reshape(B, Te, od.enc_conv_ch).cpu()
        p_ref, opt, sc = OT.train_step(p_ref, opt, od, batch, omasks, step)
        got = eng.scalars(w)
        for k in ("Linear_Loss", "Postnet_Loss", "Stop_Loss", "Weight_Regularization_Loss", "Loss"):
            assert abs(got[k] - sc[k]) <= 2e-4 * max(1.0, abs(sc[k])), (step, k, got[k], sc[k])
    pgot = eng.params.export()
    bad = {k: rel_err(pgot[k], t2n(ref)) for k, ref in p_ref.items() if not k.startswith("speaker_embedding")}
    bad = {k: v for k, v in bad.items() if v > 2e-3}
    assert not bad, bad


def test_golden_fixture_hip(dev):
    """The HIP path against the committed fixture tests/golden/tiny_train_step.npz (inputs AND expected outputs are data in the
    file: variables, batch, keep-masks -> mel / linear / stop / alignments / loss scalars / every gradient).  Nothing under
    oracle/ is called here; tests/test_kats.py::test_golden_fixtures_match_oracle keeps the file and the oracle in step."""
    import json
    import os
    from multi_speaker_tts_amd.params import Dims
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_train_step.npz"))
    cfg = json.loads(str(g["cfg"]))
    values = {k[2:]: g[k] for k in g.files if k.startswith("p/")}
    batch = {k[2:]: torch.from_numpy(g[k]).to(dev).contiguous() for k in g.files if k.startswith("b/")}
    batch = {k: (v.float() if v.is_floating_point() else v) for k, v in batch.items()}
    masks = {k[2:]: g[k] for k in g.files if k.startswith("m/")}
    B, Te, L = int(g["B"]), int(g["Te"]), int(g["L"])
    eng = TrainEngine(Dims(**cfg), device=dev, values=values)
    w = eng.plan(B, Te, L)
    eng.forward(batch, w, masks=masks)
    eng.loss_and_backward(w)
    torch.cuda.synchronize()
    assert rel_err(t2n(w.mel_out), g["mel"]) < 1e-3 and rel_err(t2n(w.linear), g["linear"]) < 1e-3
    assert rel_err(t2n(w.stop), g["stop"]) < 1e-3 and rel_err(t2n(w.align_hist).transpose(1, 2, 0), g["align"]) < 1e-3
    sc, got = json.loads(str(g["scalars"])), eng.scalars(w)
    for k in ("Linear_Loss", "Postnet_Loss", "Stop_Loss", "Weight_Regularization_Loss", "Loss"):
        assert abs(got[k] - sc[k]) <= 1e-4 * max(1.0, abs(sc[k])), (k, got[k], sc[k])
    ggot = eng.params.export(grads=True)
    bad = {}
    for k in g.files:
        if not k.startswith("g/"):
            continue
        n = k[2:]
        mine = ggot[n].astype(np.float64) + (1e-6 * np.asarray(values[n]) if OM.in_weight_reg(n) else 0.0)
        e = np.abs(mine - g[k]).max() / (np.abs(g[k]).max() + 1e-9)
        if e > 5e-3:
            bad[n] = e
    assert not bad, bad


def test_full_size_config2_properties(dev):
    """BASELINE config 2 at its full size (batch 32 x 128 tokens x 800 frames, reference widths, fp32) - too large for the oracle,
    so size-independent properties: every output / gradient / updated variable finite, the in-launch exchange of the attention
    kernels never timed out (forward and backward counters 0), the loss goes down when one batch is repeated, the workspace
    cache stays bounded, and a second engine fed the same inputs reproduces the forward (up to the summation order of the
    batch-norm statistics)."""
    import bench
    from multi_speaker_tts_amd import engine as E
    from multi_speaker_tts_amd.params import Dims
    d = Dims()
    eng = TrainEngine(d, device=dev, seed=1)
    batch = bench.synthetic_batch(d, 32, 128, 800, 1, 0, dev)
    losses = []
    for i in range(4):
        w = eng.train_step(batch)
        torch.cuda.synchronize()
        s = eng.scalars(w)
        losses.append(s["Loss"])
        assert all(np.isfinite(v) for v in s.values()), s
        assert bool(torch.isfinite(eng.params.grad).all()) and bool(torch.isfinite(eng.params.train).all())
        assert bool(torch.isfinite(w.mel_out).all()) and bool(torch.isfinite(w.align_hist).all())
        assert eng.exchange_timeouts(w) == 0
        assert len(eng._plans) <= E.MAX_PLANS
    assert w.mel_out.shape == (32, 801, 80) and losses[-1] < losses[0], losses
    a = t2n(w.align_hist)                                              # every alignment row is a distribution over the 128 tokens
    assert np.abs(a.sum(-1) - 1.0).max() < 1e-4 and a.min() >= 0.0
    eng2 = TrainEngine(d, device=dev, seed=1)
    w2 = eng2.plan(32, 128, 800)
    eng2.forward(batch, w2, seed=7)
    lin_a = w2.linear.clone()
    eng2.forward(batch, w2, seed=7)
    torch.cuda.synchronize()
    # same inputs, same masks: the forward repeats up to the summation order of the batch-norm statistics (atomic adds), 801 steps deep
    assert rel_err(t2n(w2.linear), t2n(lin_a)) < 1e-4


def test_full_size_backward_is_linear_in_the_loss_scale(dev):
    """Size-independent property of the whole backward pass at BASELINE config 2's full size: BPTT, the attention backward, every
    hoisted weight-gradient product and the batch-norm backward are linear in the incoming loss gradient, so scaling the loss by a
    power of two (exact in fp32) scales every one of the 30 M gradient elements by it - up to the order of the atomic split-K sums."""
    import bench
    from multi_speaker_tts_amd.params import Dims
    d = Dims()
    eng = TrainEngine(d, device=dev, seed=5)
    batch = bench.synthetic_batch(d, 32, 128, 800, 5, 0, dev)
    w = eng.plan(32, 128, 800)
    eng.forward(batch, w, seed=11)
    eng.loss_and_backward(w, grad_scale=1.0)
    g1 = eng.params.grad.clone()
    eng.loss_and_backward(w, grad_scale=0.25)
    torch.cuda.synchronize()
    g2 = eng.params.grad
    assert bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0
    # per variable: |g2 - g1 / 4| against that variable's own largest gradient
    ps = eng.params
    for name, shape, _ in ps.table:
        if not ps.trainable[name]:
            continue
        off, n = ps.offset[name], int(np.prod(shape))
        a, b = g1[off:off + n], g2[off:off + n]
        scale = float(a.abs().max())
        assert scale > 0 or "moving_" in name, name                 # every trainable variable receives a gradient
        # 1-D variables (biases, BN offsets) are sums over all 25 632 rows of terms that largely cancel, accumulated by atomic adds in
        # varying order: their rounding residue is relative to the sum of magnitudes, not to the (much smaller) sum
        tol = 1e-3 if len(shape) == 1 else 2e-5
        assert float((b - 0.25 * a).abs().max()) <= tol * scale, name
    assert float((g2 - 0.25 * g1).abs().max()) <= 2e-5 * float(g1.abs().max())
    assert eng.exchange_timeouts(w) == 0


def test_full_size_config3_properties(dev):
    """BASELINE config 3's per-GPU workload at its full size (batch 32 x 128 tokens x 800 frames, bf16 operands with fp32 accumulation,
    fp32 master weights / gradients / Adam): the same size-independent properties as config 2, plus closeness of its forward to the
    fp32 engine's on the same inputs and masks (the north_star tolerance is quoted for fp32; bf16 operands 801 steps deep stay
    within a few per cent of the mel range)."""
    import bench
    from multi_speaker_tts_amd import engine as E
    from multi_speaker_tts_amd.params import Dims
    d = Dims()
    eng = TrainEngine(d, device=dev, seed=1, recurrent_dtype="bf16", gemm_dtype="bf16")
    assert eng.bf is not None
    batch = bench.synthetic_batch(d, 32, 128, 800, 1, 0, dev)
    losses = []
    for i in range(4):
        w = eng.train_step(batch)
        torch.cuda.synchronize()
        s = eng.scalars(w)
        losses.append(s["Loss"])
        assert all(np.isfinite(v) for v in s.values()), s
        assert bool(torch.isfinite(eng.params.grad).all()) and bool(torch.isfinite(eng.params.train).all())
        assert bool(torch.isfinite(w.mel_out).all()) and bool(torch.isfinite(w.align_hist).all())
        assert eng.exchange_timeouts(w) == 0
        assert len(eng._plans) <= E.MAX_PLANS
    assert w.mel_out.shape == (32, 801, 80) and min(losses[1:]) < losses[0], losses      # Adam's first steps at lr 1e-3 overshoot now and then
    a = t2n(w.align_hist)
    assert np.abs(a.sum(-1) - 1.0).max() < 1e-4 and a.min() >= 0.0
    # the master weights stay fp32: the bf16 engine and an fp32 engine started from the same seed hold identical variables before a step
    e16 = TrainEngine(d, device=dev, seed=3, recurrent_dtype="bf16", gemm_dtype="bf16")
    e32 = TrainEngine(d, device=dev, seed=3)
    assert torch.equal(e16.params.train, e32.params.train)
    w16, w32 = e16.plan(32, 128, 800), e32.plan(32, 128, 800)
    e16.forward(batch, w16, seed=7)
    lin16 = w16.linear.clone()
    e32.forward(batch, w32, seed=7)
    torch.cuda.synchronize()
    assert rel_err(t2n(lin16), t2n(w32.linear)) < 5e-2


@pytest.mark.parametrize("B,Te,L,kw", [(5, 18, 9, MID), (32, 24, 6, dict(dec_lstm=1024, enc_lstm=256, spk=256, prenet=256, n_mel=80))])
def test_train_step_parity_bf16_recurrent(dev, monkeypatch, B, Te, L, kw):
    """BASELINE config 3 ("bf16 with fp32 master"): the decoder's recurrent products run on packed bf16 copies of the fp32
    master weights.  The oracle emulates exactly that (bf(X).bf(W), data gradients bf(dY).bf(W)^T, weight gradients from the
    unrounded operands); an input that sits on a bf16 rounding boundary may round differently in fp32 and fp64, so the bounds
    are a little wider than in fp32 mode."""
    monkeypatch.setattr(OM, "RECURRENT_BF16", True)
    eng, w, od, values, batch, sc, grads, out, new_p = _engine_vs_oracle(dev, B, Te, L, True, seed=13, recurrent_dtype="bf16", **kw)
    assert eng.bf is not None
    assert rel_err(t2n(w.linear), t2n(out["Linear"])) < 2e-3 and rel_err(t2n(w.mel_out), t2n(out["Mel"])) < 2e-3
    assert rel_err(t2n(w.align_hist).transpose(1, 2, 0), t2n(out["Attention_History"])) < 2e-3
    got = eng.scalars(w)
    assert abs(got["Loss"] - sc["Loss"]) <= 5e-4 * max(1.0, abs(sc["Loss"]))
    ggot = eng.params.export(grads=True)
    worst = {}
    for k, gr in grads.items():
        ref = t2n(gr).astype(np.float64)
        mine = ggot[k].astype(np.float64) + (1e-6 * np.asarray(values[k]) if OM.in_weight_reg(k) else 0.0)
        worst[k] = np.abs(mine - ref).max() / (np.abs(ref).max() + 1e-9)
    bad = {k: v for k, v in worst.items() if v > 2e-2}
    assert not bad, bad
    # the mode really differs from fp32: against the un-emulated oracle the mel is off by more than the fp32 tolerance
    monkeypatch.setattr(OM, "RECURRENT_BF16", False)
    ref32 = OM.forward(OM.to_torch(values), od, {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}, True,
                       OT.make_masks(od, B, Te, L + 1, True, seed=OT.step_seed(1234, 0)), with_vocoder=False)
    assert rel_err(t2n(w.mel_out), t2n(ref32["Mel"])) > 1e-4


def _l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / ((b ** 2).sum() + 1e-30)))


@pytest.mark.parametrize("B,Te,L,kw,recurrent", [(5, 18, 9, MID, "bf16"), (32, 24, 6, dict(dec_lstm=1024, enc_lstm=256, spk=256, prenet=256, n_mel=80, emb=128, enc_conv_ch=128, post_ch=128), "bf16"),
                                                   (32, 24, 6, dict(dec_lstm=1024, enc_lstm=256, spk=256, prenet=256, n_mel=80, emb=128, enc_conv_ch=128, post_ch=128), "f32")])
def test_train_step_parity_bf16_full(dev, monkeypatch, B, Te, L, kw, recurrent):
    """Complete BASELINE config-3 arithmetic: every dense / conv contraction of the step (forward, data gradients, weight gradients,
    the hoisted recurrent weight gradients) AND the decoder's recurrent products multiply bf16-rounded operands with fp32
    accumulation; master weights, activations, gradients, BN, losses and Adam stay fp32.  The oracle emulates exactly that
    (oracle.model.GEMM_BF16 + RECURRENT_BF16); the kernels themselves are pinned to 2e-5 by the op tests
    (test_gpu_ops.py::test_gemm_bf16_*).  A whole step cannot be compared that tightly in this mode: rounding to 8 mantissa bits is
    discontinuous, so two evaluations whose intermediate values differ by 1e-6 round a fraction of the operands differently, each
    flip worth 2^-8 of that operand, and five BN layers / BPTT amplify it - the oracle's OWN gradients move by 2-4 % (relative L2) under
    a 1e-6 perturbation of the variables (tests/test_cpu_oracle.py::test_bf16_emulation_sensitivity).  Hence: relative-L2 bounds at
    that noise level, a tight bound where the chain is short (the decoder outputs), and the emulation must be clearly closer to the
    HIP path than exact arithmetic is.  recurrent = "f32": what `bench.py --config3` runs where the persistent decoder loops exist -
    bf16 operands in every hoisted contraction, exact fp32 inside the two loops."""
    pd, od = dims_pair(**kw)
    values = OM.init_params(od, 13)
    g = np.random.default_rng(14)
    for k in values:
        if k.endswith(("bias", "beta", "bias_b")):
            values[k] = g.normal(0, 0.1, values[k].shape)
        if k.endswith("gamma"):
            values[k] = 1.0 + g.normal(0, 0.1, values[k].shape)
    batch = OT.synthetic_batch(od, B, Te, L, seed=13, ragged=True)
    masks = OT.make_masks(od, B, Te, L + 1, True, seed=OT.step_seed(1234, 0))
    eng = TrainEngine(pd, device=dev, values=values, recurrent_dtype=recurrent, gemm_dtype="bf16")
    w = eng.plan(B, Te, L)
    eng.forward(to_dev(batch, dev), w, seed=OT.step_seed(1234, 0))
    eng.loss_and_backward(w)
    torch.cuda.synchronize()
    if recurrent == "bf16" and od.dec_lstm == 1024 and eng.persist:
        # round 5: at the reference widths the all-bf16 step runs the BF16 instantiations of the persistent decoder launches
        assert eng.persist_bf16 and w.persist and w.persist_bwd and w.pdesc.recurrent_bf16 == 1 and w.pdesc_b.recurrent_bf16 == 1
        assert eng.persist_fallbacks == 0 and eng.persist_bwd_fallbacks == 0
    omasks = dict(masks)
    for i in range(od.enc_conv_n):
        omasks["relu_enc_%d" % i] = (w.enc_a[i] > 0).reshape(B, Te, od.enc_conv_ch).cpu()
    monkeypatch.setattr(OM, "KINK_BAND", 5e-2)          # bf16 products: the two evaluations differ by up to ~1e-2 near the ReLU kink
    ggot = eng.params.export(grads=True)
    res = {}
    for mode in ("emulated", "exact"):
        monkeypatch.setattr(OM, "RECURRENT_BF16", mode == "emulated" and recurrent == "bf16")
        monkeypatch.setattr(OM, "GEMM_BF16", mode == "emulated")
        _, _, sc, grads, out = OT.train_step(values, None, od, batch, omasks, 0, return_grads=True)
        gl2 = {k: _l2(ggot[k].astype(np.float64) + (1e-6 * np.asarray(values[k]) if OM.in_weight_reg(k) else 0.0), t2n(gr)) for k, gr in grads.items()}
        res[mode] = dict(linear=_l2(t2n(w.linear), t2n(out["Linear"])), mel=_l2(t2n(w.mel_out), t2n(out["Mel"])),
                         align=_l2(t2n(w.align_hist).transpose(1, 2, 0), t2n(out["Attention_History"])), loss=sc["Loss"], grads=gl2)
    em, ex = res["emulated"], res["exact"]
    assert em["linear"] < 2e-3 and em["align"] < 2e-3 and em["mel"] < 2e-2, em
    assert em["linear"] < ex["linear"] / 3 and em["mel"] < ex["mel"], (em["linear"], ex["linear"], em["mel"], ex["mel"])
    assert abs(eng.scalars(w)["Loss"] - em["loss"]) <= 1e-3 * max(1.0, abs(em["loss"]))
    bad = {k: v for k, v in em["grads"].items() if v > 0.15}
    assert not bad and float(np.median(list(em["grads"].values()))) < 6e-2, (bad, float(np.median(list(em["grads"].values()))))


@pytest.mark.parametrize("stop_bias,max_inf", [(-6.0, 9), (6.0, 9), (0.0, 14)])
def test_inference_forward_parity(dev, stop_bias, max_inf):
    """Free-running forward (speaker encoder -> encoder -> decoder with stop gating -> postnet -> Taco1)
    vs the oracle with identical injected prenet masks."""
    from multi_speaker_tts_amd.inference import InferEngine
    pd, od = dims_pair(max_inf=max_inf)
    values = OM.init_params(od, 21)
    g = np.random.default_rng(5)
    for k in values:                       # non-trivial BN statistics / biases
        if k.endswith("moving_mean"):
            values[k] = g.normal(0, 0.2, values[k].shape)
        if k.endswith("moving_variance"):
            values[k] = 0.5 + np.abs(g.normal(0, 0.5, values[k].shape))
        if k.endswith(("bias", "beta")) and "highway" not in k:
            values[k] = g.normal(0, 0.1, values[k].shape)
    values["decoder/decoder/linear_projection/dense/bias"][-1] = stop_bias
    B, Te = 3, 11
    batch = OT.synthetic_batch(od, B, Te, 4, seed=9, ragged=True)
    spk_mel = np.clip(g.normal(0, 1.5, (B * od.spk_samples, od.spk_frames, od.n_mel)), -4, 4).astype(np.float32)
    masks = OT.make_masks(od, B, Te, max_inf + 1, False, seed=77)
    ob = {"Token": batch["Token"], "Token_Length": batch["Token_Length"], "Mel": torch.zeros(B, 1, od.n_mel, dtype=torch.float64),
          "Mel_Length": torch.zeros(B, dtype=torch.int32), "Speaker_Embedding_Mel": torch.tensor(spk_mel, dtype=torch.float64)}
    ref = OM.forward(OM.to_torch(values), od, ob, False, masks, with_vocoder=True)
    eng = InferEngine(pd, device=dev, values=values)
    got = eng.forward({"Token": batch["Token"].numpy(), "Token_Length": batch["Token_Length"].numpy(), "Speaker_Embedding_Mel": spk_mel},
                      masks={k: v.numpy() for k, v in masks.items()})
    S = ref["Linear"].shape[1]
    assert got["Linear"].shape == (B, S, od.n_mel), (got["Linear"].shape, S)
    if stop_bias > 0:
        assert S == 1
    if stop_bias < -1:
        assert S == max_inf + 1
    assert rel_err(got["Linear"], t2n(ref["Linear"])) < 1e-3
    assert rel_err(got["Mel"], t2n(ref["Mel"])) < 1e-3
    assert rel_err(got["Stop"], t2n(ref["Stop"])) < 1e-3
    assert rel_err(got["Attention_History"], t2n(ref["Attention_History"])) < 1e-3
    assert rel_err(got["Spectrogram"], t2n(ref["Spectrogram"])) < 1e-3


def test_inference_decode_full_width(dev):
    """Free-running decoder at the reference's layer widths (weight-streaming path: fused prenet, stacked cell-0 kernel,
    single-launch attention, padded projection) vs the oracle, ragged token lengths, injected prenet masks."""
    from multi_speaker_tts_amd.inference import InferEngine
    pd, od = dims_pair(dec_lstm=1024, prenet=256, enc_lstm=256, spk=256, n_mel=80, max_inf=7)
    assert od.mem == 768 and od.att == 128
    B, Te = 5, 23
    assert lib.load().mstts_decoder_infer_fast(B, pd.dec_lstm, pd.prenet, pd.mem, pd.att, pd.n_mel) == 1
    values = OM.init_params(od, 33)
    g = np.random.default_rng(6)
    for k in values:
        if k.endswith(("bias", "beta")) and "highway" not in k:
            values[k] = g.normal(0, 0.1, values[k].shape)
    values["decoder/decoder/linear_projection/dense/bias"][-1] = -8.0
    batch = OT.synthetic_batch(od, B, Te, 4, seed=10, ragged=True)
    spk = g.normal(0, 1, (B, od.spk)); spk = spk / np.sqrt((spk ** 2).sum())
    masks = OT.make_masks(od, B, Te, od.max_inf + 1, False, seed=78)
    ob = {"Token": batch["Token"], "Token_Length": batch["Token_Length"], "Mel": torch.zeros(B, 1, od.n_mel, dtype=torch.float64),
          "Mel_Length": torch.zeros(B, dtype=torch.int32), "Speaker_Embedding": torch.tensor(spk, dtype=torch.float64)}
    ref = OM.forward(OM.to_torch(values), od, ob, False, masks, with_vocoder=False)
    eng = InferEngine(pd, device=dev, values=values)
    got = eng.forward({"Token": batch["Token"].numpy(), "Token_Length": batch["Token_Length"].numpy(), "Speaker_Embedding": spk.astype(np.float32)},
                      masks={k: v.numpy() for k, v in masks.items()}, with_vocoder=False)
    S = ref["Linear"].shape[1]
    assert S == od.max_inf + 1 and got["Linear"].shape == (B, S, od.n_mel)
    assert rel_err(got["Linear"], t2n(ref["Linear"])) < 1e-3
    assert rel_err(got["Mel"], t2n(ref["Mel"])) < 1e-3
    assert rel_err(got["Stop"], t2n(ref["Stop"])) < 1e-3
    assert rel_err(got["Attention_History"], t2n(ref["Attention_History"])) < 1e-3


def test_inference_rows_are_independent_of_their_batch(dev):
    """Size-independent property of the free-running forward (BASELINE config 4's shape: 16 utterances, texts up to 128 tokens, reference
    widths): a row decoded alone gives what it gives inside the batch - memory masking (Modules.py:87-93), the -inf score mask, the
    length-reversed encoder direction and the batch-norm inference path leave no trace of the other rows.  (The padded token columns
    are kept: the encoder convolutions are not length-aware in the reference, Modules.py:29-36, so a row's result does depend on its
    own padding.)  The two runs take different launch geometries (batch 16 vs batch 1)."""
    from multi_speaker_tts_amd.inference import InferEngine
    pd, od = dims_pair(dec_lstm=1024, prenet=256, enc_lstm=256, spk=256, n_mel=80, max_inf=39)
    B, Te = 16, 128
    values = OM.init_params(od, 21)
    g = np.random.default_rng(8)
    for k in values:
        if k.endswith(("bias", "beta")) and "highway" not in k:
            values[k] = g.normal(0, 0.1, values[k].shape)
    values["decoder/decoder/linear_projection/dense/bias"][-1] = -8.0           # nobody stops early: 40 steps for every row
    tok = g.integers(2, od.n_tok, size=(B, Te)).astype(np.int32)
    lengths = np.concatenate([[Te, 2], g.integers(3, Te, B - 2)]).astype(np.int32)
    for b in range(B):
        tok[b, 0] = 0; tok[b, lengths[b] - 1] = 1; tok[b, lengths[b]:] = 1    # <S> ... <E>, padded with <E> like the feeder
    spk = g.normal(0, 1, (B, od.spk)); spk = (spk / np.sqrt((spk ** 2).sum())).astype(np.float32)
    masks = {k: v.numpy() for k, v in OT.make_masks(od, B, Te, od.max_inf + 1, False, seed=5).items()}
    eng = InferEngine(pd, device=dev, values=values)
    full = eng.forward({"Token": tok, "Token_Length": lengths, "Speaker_Embedding": spk}, masks=masks, with_vocoder=False)
    assert full["Linear"].shape == (B, od.max_inf + 1, od.n_mel) and np.isfinite(full["Mel"]).all()
    for b in (0, 1, 5, 15):
        n = int(lengths[b])
        one = eng.forward({"Token": tok[b:b + 1].copy(), "Token_Length": lengths[b:b + 1].copy(), "Speaker_Embedding": spk[b:b + 1].copy()},
                          masks={k: np.ascontiguousarray(np.take(v, [b], axis=OT.mask_batch_axis(k))) for k, v in masks.items()}, with_vocoder=False)
        assert rel_err(one["Linear"][0], full["Linear"][b]) < 1e-3, b
        assert rel_err(one["Mel"][0], full["Mel"][b]) < 1e-3, b
        assert full["Attention_History"].shape == (B, Te, od.max_inf + 1)                    # [row, token, step]
        assert np.abs(one["Attention_History"][0] - full["Attention_History"][b]).max() < 1e-3, b
        assert np.abs(full["Attention_History"][b][n:, :]).max(initial=0.0) == 0.0, b      # no weight on the padding
        assert np.abs(full["Attention_History"][b].sum(0) - 1.0).max() < 1e-4, b


def test_speaker_encoder_training_zoneout(dev):
    """The frozen speaker encoder inside a TRAIN step runs with stochastic zoneout (the reference feeds Is_Training into it,
    MSTTS_SV.py:49-56): HIP forward with Philox masks drawn on the device vs the oracle's training-mode speaker encoder."""
    from multi_speaker_tts_amd.inference import InferEngine
    from multi_speaker_tts_amd.masks import MaskSet
    pd, od = dims_pair(spk=32, spk_lstm=32)
    values = OM.init_params(od, 3)
    g = np.random.default_rng(4)
    B = 3
    NB = B * od.spk_samples
    mel = np.clip(g.normal(0, 1.5, (NB, od.spk_frames, od.n_mel)), -4, 4).astype(np.float32)
    masks = OT.make_masks(od, 1, 1, 1, True, seed=99, speaker_windows=NB)
    ref = OM.speaker_encoder(OM.to_torch(values), od, torch.tensor(mel, dtype=torch.float64), True, masks)
    ref_inf = OM.speaker_encoder(OM.to_torch(values), od, torch.tensor(mel, dtype=torch.float64), False, masks)
    eng = InferEngine(pd, device=dev, values=values)
    ms = MaskSet(pd, 1, 1, 1, True, dev, speaker_windows=NB)
    ms.draw(99)
    for k in ("s_zc_0", "s_zh_2"):
        assert np.array_equal(t2n(ms[k]), masks[k].numpy())
    got = eng.speaker_embedding(torch.tensor(mel, device=dev), masks=ms)
    got_inf = eng.speaker_embedding(torch.tensor(mel, device=dev))
    torch.cuda.synchronize()
    assert rel_err(t2n(got), t2n(ref)) < 1e-4 and rel_err(t2n(got_inf), t2n(ref_inf)) < 1e-4
    assert rel_err(t2n(got), t2n(ref_inf)) > 1e-3             # the two modes really differ


def test_tacotron2_surface(dev, tmp_path, monkeypatch):
    """Drop-in surface: Tacotron2(is_Training).Train_Step / Inference / Save / Restore round trip."""
    from multi_speaker_tts_amd import Hyper_Parameters as hp
    from multi_speaker_tts_amd.MSTTS_SV import Tacotron2, TRAIN_KEYS
    from multi_speaker_tts_amd.params import Dims
    monkeypatch.setattr(hp, "Checkpoint_Path", str(tmp_path / "ckpt"))
    monkeypatch.setattr(hp, "Inference_Path", str(tmp_path / "inf"))
    dims = Dims(emb=32, enc_conv_ch=32, enc_lstm=16, spk=256, prenet=16, dec_lstm=32, post_ch=16, bank_ch=8, proj1_ch=16, birnn=8, n_spec=20,
                spk_lstm=256, max_inf=6)
    monkeypatch.setattr(hp.Speaker_Embedding, "Checkpoint_Path", str(tmp_path / "no_spk"))
    monkeypatch.setattr(hp.Taco1_Mel_to_Spect, "Checkpoint_Path", str(tmp_path / "no_voc"))
    with pytest.raises(ValueError):            # MSTTS_SV.py:226-227,239-240: missing sub-model checkpoints are an error
        Tacotron2(is_Training=True, device=dev, dims=dims)
    t = Tacotron2(is_Training=True, device=dev, dims=dims, allow_random_init=True)
    pat = t.feeder.Get_Train_Pattern(batch_Size=2, token_Length=9, mel_Length=200)
    r0 = t.Train_Step(pat)
    assert set(TRAIN_KEYS) <= set(r0) and r0["Global_Step"] == 0 and np.isfinite(r0["Loss"])
    r1 = t.Train_Step(pat)
    assert r1["Global_Step"] == 1 and r1["Loss"] < r0["Loss"] + 1.0
    t.Save()
    before = t.params.export()
    t.params.train.zero_()
    t.Restore()
    after = t.params.export()
    assert t.global_step == 2 and all(np.array_equal(before[k], after[k]) for k in before if k.startswith(("encoder", "decoder", "attention")))
    # the reference's own checkpoint format (TF V2 bundle): export, then Restore() from a directory holding only that
    import shutil
    tf_dir = tmp_path / "tfckpt"
    prefix = t.Export_TF_Checkpoint(str(tf_dir))
    assert (tf_dir / "checkpoint").exists() and prefix.endswith("CHECKPOINT-2")
    m_before, v_before = t.params.adam_m.clone(), t.params.adam_v.clone()
    t.params.train.zero_(); t.params.adam_m.zero_(); t.params.adam_v.zero_(); t.train_engine.global_step = 0
    monkeypatch.setattr(hp, "Checkpoint_Path", str(tf_dir))
    t.Restore()
    again = t.params.export()
    assert t.global_step == 2 and all(np.array_equal(before[k], again[k]) for k in before if k.startswith(("encoder", "decoder", "attention")))
    assert torch.equal(t.params.adam_m, m_before) and torch.equal(t.params.adam_v, v_before)
    monkeypatch.setattr(hp, "Checkpoint_Path", str(tmp_path / "ckpt"))
    mels = [np.clip(np.random.default_rng(i).normal(0, 1.5, (230, 80)), -4, 4).astype(np.float32) for i in range(2)]
    res = t.Inference(None, ["Please call Stella.", "Who knows?"], speaker_Mel_List=mels)
    S = res["Linear"].shape[1]
    assert res["Linear"].shape == (2, S, 80) and res["Spectrogram"].shape == (2, S, 20) and res["Attention_History"].shape[:2] == (2, 21)
    assert len(res["Cut"]) == 2 and res["Cut"][1]["Attention_History"].shape[0] == len("Who knows?") + 2
    with pytest.raises(KeyError):
        t.Inference(None, ["ünknown"], speaker_Mel_List=mels[:1])


def test_inference_from_wav_paths(dev, tmp_path, monkeypatch):
    """Tacotron2.Inference(path_List=[wav ...], text_List) end to end from real files (MSTTS_SV.py:295-299, Feeder.py:186-233): two
    48 kHz wavs with silence around the signal -> load at 16 kHz, trim (top_db 15) x 0.99 -> mel on the GPU -> five 64-frame speaker
    windows -> speaker encoder -> free-running decoder -> Taco1 -> cut at the stop token -> NPZ + Griffin-Lim WAV files."""
    from scipy.io import wavfile
    from multi_speaker_tts_amd import Hyper_Parameters as hp, Audio, Feeder as F
    from multi_speaker_tts_amd.MSTTS_SV import Tacotron2
    from multi_speaker_tts_amd.params import Dims
    from oracle import audio as OA, feeder as OF
    monkeypatch.setattr(hp, "Checkpoint_Path", str(tmp_path / "ckpt"))
    monkeypatch.setattr(hp, "Inference_Path", str(tmp_path / "inf"))
    dims = Dims(emb=32, enc_conv_ch=32, enc_lstm=16, spk=256, prenet=16, dec_lstm=32, post_ch=16, bank_ch=8, proj1_ch=16, birnn=8,
                spk_lstm=256, max_inf=6)
    assert dims.n_spec == hp.Sound.Spectrogram_Dim
    g = np.random.default_rng(3)
    paths = []
    for i, seconds in enumerate((3.1, 1.2)):           # the second one is shorter than the 192 frames five windows need
        n = int(48000 * seconds)
        t = np.arange(n) / 48000.0
        y = 0.4 * np.sin(2 * np.pi * (180 + 60 * i) * t) * (0.6 + 0.4 * np.sin(2 * np.pi * 3 * t)) + 0.02 * g.normal(size=n)
        y = np.concatenate([1e-4 * g.normal(size=9600), y, 1e-4 * g.normal(size=14400)])          # 0.2 s / 0.3 s of near silence
        p = str(tmp_path / ("spk%d.wav" % i))
        wavfile.write(p, 48000, (y * 32767).astype(np.int16))
        paths.append(p)
    texts = ["Please call Stella.", "Who knows?"]
    t = Tacotron2(is_Training=False, device=dev, dims=dims, allow_random_init=True)
    # the feeder's leg, piece by piece
    sigs = [F.load_wav(p) for p in paths]
    assert all(s.dtype == np.float32 and 0.3 < np.abs(s).max() <= 0.99 for s in sigs)                 # [-1, 1] samples x 0.99 (Feeder.py:215)
    assert 2.9 * 16000 < sigs[0].shape[0] < 3.3 * 16000 and 1.0 * 16000 < sigs[1].shape[0] < 1.4 * 16000     # resampled, silence trimmed
    mels = [np.transpose(Audio.melspectrogram(y=s, num_freq=hp.Sound.Spectrogram_Dim, frame_shift_ms=hp.Sound.Frame_Shift,
                                              frame_length_ms=hp.Sound.Frame_Length, num_mels=hp.Sound.Mel_Dim, sample_rate=hp.Sound.Sample_Rate,
                                              max_abs_value=hp.Sound.Max_Abs_Mel, device=dev)).astype(np.float32) for s in sigs]
    for s, m in zip(sigs, mels):
        assert m.shape == (1 + s.shape[0] // 200, 80) and np.abs(m.T - OA.melspectrogram(s)).max() < 2e-3
    pattern = t.feeder.Get_Inference_Pattern(paths, texts)
    assert mels[0].shape[0] >= 192 > mels[1].shape[0]
    assert np.array_equal(pattern["Speaker_Embedding_Mel"], OF.speaker_windows(mels))                 # [2 * 5, 64, 80], Feeder.py:62-87
    assert np.array_equal(pattern["Token"][0, :21], [0, 29, 25, 18, 14, 32, 18, 2, 16, 14, 25, 25, 2, 32, 33, 18, 25, 25, 14, 10, 1])
    # the whole call, from paths and from the same mels
    od = OM.Dims(**{f: getattr(dims, f) for f in ("emb", "enc_conv_ch", "enc_lstm", "spk", "prenet", "dec_lstm", "post_ch", "bank_ch", "proj1_ch",
                                                 "birnn", "spk_lstm", "max_inf")})
    masks = {k: v.numpy() for k, v in OT.make_masks(od, 2, pattern["Token"].shape[1], od.max_inf + 1, False, seed=31).items()}
    res = t.Inference(paths, texts, file_Prefix="pl", masks=masks)
    again = t.Inference(None, texts, speaker_Mel_List=mels, masks=masks, export=False)
    for k in ("Linear", "Mel", "Stop", "Spectrogram"):
        assert np.array_equal(res[k], again[k]), k
    S = res["Linear"].shape[1]
    assert res["Spectrogram"].shape == (2, S, hp.Sound.Spectrogram_Dim) and np.isfinite(res["Spectrogram"]).all()
    for i in range(2):
        npz = np.load(str(tmp_path / "inf" / "NPZ" / ("pl.IDX_%d.npz" % i)))
        assert np.array_equal(npz["Mel"], res["Cut"][i]["Mel"]) and npz["Attention_History"].shape[0] == len(texts[i]) + 2
        wav_path = tmp_path / "inf" / "WAV" / ("pl.IDX_%d.WAV" % i)
        if res["Cut"][i]["Spectrogram"].shape[0] > 1:          # the reference refuses one-frame spectrograms (MSTTS_SV.py:403-406)
            rate, wav = wavfile.read(str(wav_path))
            assert rate == hp.Sound.Sample_Rate and wav.shape[0] > 0 and np.isfinite(wav).all()
    with pytest.raises(ValueError):
        t.Inference(paths[:1], texts)


def test_pattern_generate_cli_to_training(dev, tmp_path, monkeypatch):
    """SURVEY 8(f).2 end to end: `Pattern_Generate -lj <corpus>` (corpus walker -> GPU mel extraction -> pattern pickles ->
    METADATA.PICKLE), then Tacotron2.Train reads those patterns through the feeder's producer thread and takes two steps."""
    from scipy.io import wavfile
    from multi_speaker_tts_amd import Hyper_Parameters as hp
    from multi_speaker_tts_amd import Pattern_Generate as PG
    from multi_speaker_tts_amd.MSTTS_SV import Tacotron2
    from multi_speaker_tts_amd.params import Dims
    lj = tmp_path / "LJ"
    (lj / "wavs").mkdir(parents=True)
    rows = []
    sentences = ["Please call Stella.", "Who knows much believes the less.", "His voice is tested now.", "Things are always at their best."]
    for i, text in enumerate(sentences):
        t = np.arange(int((0.7 + 0.1 * i) * 16000)) / 16000.0
        y = 0.4 * np.sin(2 * np.pi * (180 + 40 * i) * t) * (0.6 + 0.4 * np.sin(2 * np.pi * 3 * t))
        wavfile.write(str(lj / "wavs" / ("LJ001-%04d.wav" % i)), 16000, (y * 32767).astype(np.int16))
        rows.append("LJ001-%04d|%s|%s" % (i, text, text))
    (lj / "metadata.csv").write_text("\n".join(rows) + "\n", encoding="utf-8")
    monkeypatch.setattr(hp.Train, "Pattern_Path", str(tmp_path / "patterns"))
    monkeypatch.setattr(hp.Train, "Main_Train_Dataset_List", ["LJ"])
    monkeypatch.setattr(hp.Train, "Batch_Size", 2)
    monkeypatch.setattr(hp, "Checkpoint_Path", str(tmp_path / "ckpt"))
    assert PG.main(["-lj", str(lj)], device=dev) == 4
    names = sorted(os.listdir(tmp_path / "patterns"))
    assert "METADATA.PICKLE" in names and "LJ.LJ001-0002.PICKLE" in names and len(names) == 5
    import pickle
    with open(tmp_path / "patterns" / "LJ.LJ001-0000.PICKLE", "rb") as f:
        pat = pickle.load(f)
    assert pat["Text"] == "PLEASE CALL STELLA." and pat["Dataset"] == "LJ" and pat["Mel"].shape[1] == 80 and 40 <= pat["Mel"].shape[0] <= 57       # 0.7 s at 200-sample hops, minus what the silence trim removed
    assert list(pat["Token"]) == [29, 25, 18, 14, 32, 18, 2, 16, 14, 25, 25, 2, 32, 33, 18, 25, 25, 14, 10]         # SURVEY A.3 without <S>/<E>
    dims = Dims(emb=32, enc_conv_ch=32, enc_lstm=16, spk=256, prenet=16, dec_lstm=32, post_ch=16, bank_ch=8, proj1_ch=16, birnn=8, n_spec=20, spk_lstm=256)
    t = Tacotron2(is_Training=True, device=dev, dims=dims, allow_random_init=True)
    try:
        t.Train(max_steps=2, run_inference=False)
    finally:
        t.feeder.close()
    assert t.global_step == 2


def test_variable_length_training_and_convergence(dev):
    """Bucketed real data gives a different (tokens, frames) shape almost every step (Feeder.py:111-124,136-175).  Every workspace set is a
    view of ONE arena sized for the largest shape to come (arena_hint), so from the second step on a NEW shape allocates nothing on the
    device - the caching allocator's allocation count stays flat across train_step - the sets of different shapes alias each
    other, every shape trains, and repeating one batch drives the loss down (the whole step - forward, BPTT, TF-Adam - pulls one way).
    The suite runs with MSTTS_ARENA_POISON=1 (tests/conftest.py): whenever a set is activated its whole extent is filled with NaN first, so a
    kernel that relied on zero-initialised or stale workspace memory would put NaN into the loss here."""
    from multi_speaker_tts_amd import engine as E
    pd, od = dims_pair(**MID)
    shapes = [(4, 9, 6), (4, 12, 8), (3, 7, 11), (4, 9, 6), (2, 15, 5), (4, 10, 9), (4, 15, 11), (1, 5, 3)]
    eng = TrainEngine(pd, device=dev, seed=3, arena_hint=(4, 15, 11))
    assert eng.arena_poison
    counts, extents = [], []
    for i, (B, Te, L) in enumerate(shapes):
        batch = to_dev(OT.synthetic_batch(od, B, Te, L, seed=20 + i, ragged=True), dev)
        torch.cuda.synchronize()
        n0 = torch.cuda.memory_stats(dev)["allocation.all.allocated"]
        w = eng.train_step(batch)
        torch.cuda.synchronize()
        counts.append(torch.cuda.memory_stats(dev)["allocation.all.allocated"] - n0)
        extents.append(w.extent_bytes)
        assert np.isfinite(eng.scalars(w)["Loss"]) and len(eng._plans) <= E.MAX_PLANS
    assert counts[0] > 0 and all(c == 0 for c in counts[1:]), counts           # the first step builds the arena (and the engine's one-off buffers); no later one allocates
    assert eng._arena.generation == 1 and max(extents) <= eng._arena.cap         # sized once, never grown
    w_a, w_b = eng.plan(4, 9, 6), eng.plan(4, 12, 8)
    base = eng._arena.buf.data_ptr()
    assert w_a is eng.plan(4, 9, 6)                                              # cached views
    for w_ in (w_a, w_b):                                                        # every set starts at the arena's first byte: different shapes alias the same memory
        first = min(t.data_ptr() for t in w_.masks.buf.values())
        assert first == base and base <= w_.emb.data_ptr() < base + w_.extent_bytes <= base + eng._arena.cap
    # a shape beyond the hint grows the arena once (and only then)
    big = to_dev(OT.synthetic_batch(od, 5, 17, 13, seed=77, ragged=True), dev)
    w = eng.train_step(big)
    assert eng._arena.generation == 2 and np.isfinite(eng.scalars(w)["Loss"])
    batch = to_dev(OT.synthetic_batch(od, 4, 9, 6, seed=99), dev)
    losses = []
    for _ in range(40):
        w = eng.train_step(batch)
        losses.append(eng.scalars(w)["Loss"])
    assert np.mean(losses[-5:]) < 0.95 * np.mean(losses[:5]), (losses[:5], losses[-5:])      # noise targets + dropout .5: slow but steady
    assert eng.global_step == len(shapes) + 1 + 40


def test_gradient_ready_ranges(dev):
    """The three points at which the train step hands gradient ranges to the all-reduce (postnet -> decoder/attention ->
    encoder) partition the slab, and a range is final when announced (it does not change afterwards)."""
    pd, od = dims_pair()
    eng = TrainEngine(pd, device=dev, seed=5)
    batch = to_dev(OT.synthetic_batch(od, 3, 9, 6, seed=4), dev)
    w = eng.plan(3, 9, 6)
    eng.forward(batch, w)
    seen, snaps = [], []

    def on_ready(lo, hi):
        seen.append((lo, hi))
        snaps.append(eng.params.grad[lo:hi].clone())
    eng.loss_and_backward(w, on_ready=on_ready)
    torch.cuda.synchronize()
    assert len(seen) == 3 and sorted(seen)[0][0] == 0 and sorted(seen)[-1][1] == eng.params.n_train
    s = sorted(seen)
    assert all(s[i][1] == s[i + 1][0] for i in range(2))                       # contiguous, no overlap
    assert seen[0][0] > seen[1][0] > seen[2][0]                                # postnet (end of slab) first, encoder last
    for (lo, hi), snap in zip(seen, snaps):
        assert torch.equal(eng.params.grad[lo:hi], snap) and float(snap.abs().max()) > 0


def test_inference_results_own_their_host_blocks(dev):
    """inference.to_host (the results of InferEngine.forward): bit-equal to `.cpu()`, one page-locked block per array - a later call neither aliases nor
    overwrites an earlier call's arrays - and non-contiguous / integer tensors go through as they are."""
    from multi_speaker_tts_amd.inference import to_host
    g = torch.Generator(device="cpu").manual_seed(3)
    a = torch.randn(37, 129, generator=g).to(dev)
    b = torch.randint(0, 100, (5, 7), generator=g, dtype=torch.int32).to(dev)
    r1 = to_host({"a": a, "b": b, "t": a.t()})
    assert np.array_equal(r1["a"], a.cpu().numpy()) and np.array_equal(r1["b"], b.cpu().numpy()) and np.array_equal(r1["t"], a.t().cpu().numpy())
    keep = r1["a"].copy()
    a.mul_(2.0)
    r2 = to_host({"a": a})
    assert np.array_equal(r1["a"], keep) and np.array_equal(r2["a"], 2.0 * keep)
    assert not np.shares_memory(r1["a"], r2["a"])
    del r2
    r3 = to_host({"a": a})                       # (may reuse r2's block - never r1's, which is still held)
    assert np.array_equal(r1["a"], keep) and not np.shares_memory(r1["a"], r3["a"])


def test_mel_to_spectrogram_reference_widths(dev):
    """VERDICT r5 missing #3 / weak #3: the mel -> spectrogram network (Taco1_Mel_to_Spect/Modules.py:8-105 as wired at MSTTS_SV.py:100-115)
    and the speaker encoder (Speaker_Embedding/Modules.py:6-37,127-137) had oracle checks at reduced widths only; at the reference's
    widths the only tests compared the engine with itself.  Here: EVERY width is the reference's (conv bank 8 x 128 -> max-pool -> 1024 -> 256
    -> 80, residual, highway 4 x 80, BiRNN 2 x 128, dense 1025; speaker stack dense 256 + 3 x LSTM 256 with residual wrappers on cells 0 / 1),
    at BASELINE configs[3]'s size - batch 16 x 401 frames and 5 x 16 speaker windows of 64 frames - against the fp64 oracle, tolerance 1e-3
    (north_star), persistent recurrences asserted to have run (Q13: no length masking in the BiRNN; Q14: whole-tensor l2 normalisation)."""
    from multi_speaker_tts_amd.inference import InferEngine
    pd, od = Dims(), OM.Dims()
    assert (od.bank_k, od.bank_ch, od.proj1_ch, od.n_mel, od.highway_n, od.birnn, od.n_spec, od.spk_lstm, od.spk_lstm_n, od.spk) == (8, 128, 256, 80, 4, 128, 1025, 256, 3, 256)
    values = OM.init_params(od, 41)
    g = np.random.default_rng(17)
    for k in values:                       # trained-like statistics: moving moments away from (0, 1), non-zero biases
        if not k.startswith(("mel_to_spectrogram", "speaker_embedding")):
            continue
        if k.endswith("moving_mean"):
            values[k] = g.normal(0, 0.2, values[k].shape)
        if k.endswith("moving_variance"):
            values[k] = 0.5 + np.abs(g.normal(0, 0.5, values[k].shape))
        if k.endswith(("bias", "beta")) and "highway" not in k:
            values[k] = g.normal(0, 0.1, values[k].shape)
    B, S = 16, 401
    mel = np.clip(g.normal(0, 1.5, (B, S, od.n_mel)), -4, 4).astype(np.float32)
    spk_mel = np.clip(g.normal(0, 1.5, (B * od.spk_samples, od.spk_frames, od.n_mel)), -4, 4).astype(np.float32)
    p64 = OM.to_torch({k: v for k, v in values.items() if k.startswith(("mel_to_spectrogram", "speaker_embedding"))})
    with torch.no_grad():
        ref_spec = OM.taco1_forward(p64, od, torch.tensor(mel, dtype=torch.float64), False).numpy()
        ref_emb = OM.speaker_encoder(p64, od, torch.tensor(spk_mel, dtype=torch.float64), False).numpy()
    eng = InferEngine(pd, device=dev, values=values)
    n0 = eng.persist_lstm_launches
    spec = t2n(eng.mel_to_spectrogram(torch.tensor(mel, device=dev).view(B * S, od.n_mel), B, S))
    emb = t2n(eng.speaker_embedding(torch.tensor(spk_mel, device=dev)))
    assert spec.shape == (B, S, od.n_spec) and emb.shape == (B, od.spk)
    e_spec, e_emb = rel_err(spec, ref_spec), rel_err(emb, ref_emb)
    # per-frame check as well: the BiRNN's backward direction starts at the LAST frame, the forward one at the first - both ends must hold
    e_ends = max(rel_err(spec[:, :3], ref_spec[:, :3]), rel_err(spec[:, -3:], ref_spec[:, -3:]))
    print("taco1 @ reference widths: spectrogram %.2e (ends %.2e), speaker embedding %.2e, persistent LSTM launches %d, fallbacks %d"
          % (e_spec, e_ends, e_emb, eng.persist_lstm_launches - n0, eng.persist_lstm_fallbacks))
    assert e_spec < 1e-3 and e_ends < 1e-3, (e_spec, e_ends)
    assert e_emb < 1e-3, e_emb
    assert abs(float(np.sqrt((emb.astype(np.float64) ** 2).sum())) - 1.0) < 1e-5          # Q14: the WHOLE [B, 256] tensor has unit norm
    if lib.load().mstts_persist_lstm_fwd_supported_n(B, pd.birnn, 2):
        assert eng.persist_lstm_launches - n0 == 1 + pd.spk_lstm_n and eng.persist_lstm_fallbacks == 0


def test_batch_uploader_first_upload_survives_a_busy_device(dev):
    """Round-6 race, found as non-finite parameters in the first steps of 2 of 6 Tacotron2.Train_Step runs inside a busy process: the upload
    blocks were zero-filled on the CALLER's stream and written on the copy stream, unordered - the fill could land behind the first upload.
    Here: a fresh uploader per round, the caller's stream kept busy with fills in front of the first stage(); what arrives on the device must
    be the pattern, every round, for both slots."""
    from multi_speaker_tts_amd.MSTTS_SV import _BatchUploader
    g = np.random.default_rng(0)
    junk = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    for rnd in range(12):
        pats = [{"Token": g.integers(2, 40, size=(32, 100 + rnd)).astype(np.int32), "Token_Length": np.full(32, 100 + rnd, np.int32),
                 "Mel": g.normal(0, 1, size=(32, 300 + 7 * rnd, 80)).astype(np.float32), "Mel_Length": np.full(32, 300, np.int32),
                 "Speaker_Embedding_Mel": g.normal(0, 1, size=(160, 64, 80)).astype(np.float32)} for _ in range(3)]
        for _ in range(6):
            junk.fill_(float(rnd))                                     # ~0.1 ms each of queued work on the caller's stream
        up = _BatchUploader(dev, presize={"Token": 32 * 256, "Mel": 32 * 721 * 80, "Speaker_Embedding_Mel": 160 * 64 * 80})
        for p in pats:
            b = up.stage(p)
            torch.cuda.current_stream().wait_event(b["_uploaded"])
            got = {k: b[k].clone() for k in p}
            up.release(b)
            torch.cuda.synchronize()
            for k in p:
                assert np.array_equal(t2n(got[k]), p[k]), (rnd, k)


@pytest.mark.parametrize("cfg", ["reference widths, persistent launches"])
def test_deterministic_training_is_bit_reproducible(dev, cfg):
    """TrainEngine(deterministic=True) fixes every summation order of a train step - no reduction cut chosen by
    the engine or the library (split_k = 1, mstts_gemm_deterministic), the column sums behind batch norm / bias gradients, the embedding
    scatter and the attention layer's d_keys in their one-add-per-element forms - so two engines started from the same variables, fed the same
    batches, end BIT-IDENTICAL after three optimizer steps: every variable, both Adam slots, the batch-norm moving statistics, the gradient
    slab of the last step.  (The default mode's atomics leave the last bit or two open: recorded beside it, not asserted.)
    Scope: the path the reference widths take - the persistent launches.  The launch-per-step fallback loops (other widths, or a launch that
    gave up) cut their skinny products along K with atomics of their own and are NOT covered: measured 8.6e-7 between two such runs."""
    from tests.helpers import dims_pair
    ref = cfg.startswith("reference")
    kw = dict(emb=64, enc_conv_ch=64, enc_lstm=256, spk=256, prenet=256, dec_lstm=1024, n_mel=80, post_ch=64) if ref else dict(MID)
    pd, od = dims_pair(**kw)
    values = OM.init_params(od, 5)
    shapes = [(4, 33, 21), (3, 40, 12), (4, 33, 21)]
    batches = [to_dev(OT.synthetic_batch(od, B, Te, L, seed=30 + i, ragged=True), dev) for i, (B, Te, L) in enumerate(shapes)]

    def run(det):
        eng = TrainEngine(pd, device=dev, values=values, seed=11, deterministic=det)
        assert eng.deterministic == det
        for b in batches:
            w = eng.train_step(b)
        torch.cuda.synchronize()
        if ref:
            assert w.persist and eng.persist_fallbacks == 0 and eng.persist_bwd_fallbacks == 0 and eng.persist_enc_fallbacks == 0
        ps = eng.params
        return {"train": ps.train.clone(), "m": ps.adam_m.clone(), "v": ps.adam_v.clone(), "frozen": ps.frozen.clone(), "grad": ps.grad.clone()}, eng.scalars(w)["Loss"]
    a, la = run(True)
    b, lb = run(True)
    for k in a:
        assert bool(torch.isfinite(a[k]).all()), k
        assert torch.equal(a[k], b[k]), "%s differs between two deterministic runs: max |d| %g" % (k, float((a[k] - b[k]).abs().max()))
    assert abs(la - lb) <= 1e-5 * max(1.0, abs(la))                      # (the loss words themselves stay atomic sums: they feed nothing)
    c, lc = run(False)
    d = {k: float((a[k] - c[k]).abs().max() / (a[k].abs().max() + 1e-30)) for k in a}
    print("deterministic vs default schedule after 3 steps (max |difference| / max |value|):", d)
    assert all(v < 2e-3 for v in d.values()), d                         # same arithmetic, other summation orders


def test_side_chains_change_nothing(dev, monkeypatch):
    """Round 6: in a train step three runs of work leave the main stream for the encoder's stream (the vocoder conv-bank's statistics side effect under
    the loss and the postnet's backward pass, the postnet's weight-gradient products behind it, the encoder's backward pass behind its BPTT launch
    beside the decoder's weight-gradient products; engine.VOC_OVERLAP / POSTNET_WGRAD_OVERLAP / ENC_TAIL_OVERLAP).  They reorder launches across
    streams, not arithmetic: in the deterministic mode three optimizer steps with the chains and without them end BIT-IDENTICAL - variables, Adam
    slots, moving statistics (the vocoder's included), the gradient slab."""
    from tests.helpers import dims_pair
    from multi_speaker_tts_amd import engine as E
    pd, od = dims_pair(emb=64, enc_conv_ch=64, enc_lstm=256, spk=256, prenet=256, dec_lstm=1024, n_mel=80, post_ch=64)
    values = OM.init_params(od, 6)
    shapes = [(4, 33, 21), (3, 40, 12), (4, 33, 21)]
    batches = [to_dev(OT.synthetic_batch(od, B, Te, L, seed=40 + i, ragged=True), dev) for i, (B, Te, L) in enumerate(shapes)]

    def run(on):
        for name in ("VOC_OVERLAP", "POSTNET_WGRAD_OVERLAP", "ENC_TAIL_OVERLAP"):
            monkeypatch.setattr(E, name, on)
        eng = TrainEngine(pd, device=dev, values=values, seed=12, deterministic=True)
        assert eng.update_vocoder_bn
        seen, inner = [], eng._vocoder_bn_update

        def spy(w):
            seen.append(torch.cuda.current_stream() == eng._enc_stream)     # where the chain is enqueued
            return inner(w)
        eng._vocoder_bn_update = spy
        for b in batches:
            w = eng.train_step(b)
            assert w.voc_done is None and not w.side_wgrads              # joined in front of the BPTT launch
        torch.cuda.synchronize()
        assert seen == [on] * 3, seen
        assert eng.persist_fallbacks == 0 and eng.persist_bwd_fallbacks == 0 and eng.persist_enc_fallbacks == 0
        ps = eng.params
        return {"train": ps.train.clone(), "m": ps.adam_m.clone(), "v": ps.adam_v.clone(), "frozen": ps.frozen.clone(), "grad": ps.grad.clone()}
    a, b = run(True), run(False)
    for k in a:
        assert bool(torch.isfinite(a[k]).all()), k
        assert torch.equal(a[k], b[k]), "%s differs with / without the side chains: max |d| %g" % (k, float((a[k] - b[k]).abs().max()))
