"""Persistent BiLSTM launches (csrc/persist_lstm.hip): all T steps of both directions in one launch, forward and BPTT, against the
launch-per-step pair drivers (mstts_lstm_seq_fwd_pair / mstts_lstm_seq_bwd_pair, themselves checked against the oracle in
test_gpu_ops.py / test_gpu_model.py) on the same buffers: ragged lengths, reversed direction, zoneout masks, fewer than 32 rows."""
import ctypes as C

import numpy as np
import pytest
import torch

from multi_speaker_tts_amd import lib
from oracle import model as OM, train as OT
from tests.helpers import dims_pair, rel_err, t2n, to_dev

pytestmark = pytest.mark.gpu
H = 256


def _setup(dev, B, T, seed, zoneout=0.1):
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
    lens = torch.randint(max(1, T // 3), T + 1, (B,), generator=g, dtype=torch.int32)
    lens[0] = T
    if B > 1:
        lens[1] = 1
    st = {"B": B, "T": T, "lens": lens.to(dev), "zoneout": zoneout}
    for dr in ("fw", "bw"):
        st["wh_" + dr] = r(H, 4 * H, scale=0.06)
        st["xw_" + dr] = r(B, T, 4 * H, scale=0.8)
        st["zc_" + dr] = (torch.rand(T, B, H, generator=g) > zoneout).to(torch.uint8).to(dev)
        st["zh_" + dr] = (torch.rand(T, B, H, generator=g) > zoneout).to(torch.uint8).to(dev)
        st["dout_" + dr] = r(B, T, H, scale=0.5)
    return st


def _fwd_bufs(dev, B, T):
    z = lambda *s: torch.full(s, float("nan"), device=dev)
    o = {"out": torch.zeros(B, T, 2 * H, device=dev)}
    for dr in ("fw", "bw"):
        o["c_" + dr], o["h_" + dr] = z(T + 1, B, H), z(T + 1, B, H)
        o["acts_" + dr], o["craw_" + dr] = z(T, B, 4 * H), z(T, B, H)
    return o


def _fwd_descs(st, o, extra):
    B, T = st["B"], st["T"]
    qs = []
    for di, dr in enumerate(("fw", "bw")):
        q = lib.LstmSeqFwd()
        q.B, q.T, q.H = B, T, H
        q.xw = lib.ptr(st["xw_" + dr]); q.wh = lib.ptr(st["wh_" + dr]); q.wh_ld = 4 * H
        q.lengths = lib.ptr(st["lens"]); q.reverse = di; q.zoneout = st["zoneout"]
        q.zc = lib.ptr(st["zc_" + dr]); q.zh = lib.ptr(st["zh_" + dr])
        q.out = lib.ptr(o["out"], di * H); q.out_sb = T * 2 * H; q.out_st = 2 * H
        q.c_hist = lib.ptr(o["c_" + dr]); q.h_hist = lib.ptr(o["h_" + dr]); q.acts = lib.ptr(o["acts_" + dr]); q.c_raw = lib.ptr(o["craw_" + dr])
        if extra is not None:
            q.gates_ws = lib.ptr(extra["gates_" + dr])
        qs.append(q)
    return qs


def _run_fwd(dev, st, persistent):
    B, T = st["B"], st["T"]
    L = lib.load()
    o = _fwd_bufs(dev, B, T)
    if persistent:
        assert L.mstts_persist_lstm_supported(B, H)
        n = L.mstts_persist_lstm_pack_floats()
        pk = {}
        for dr in ("fw", "bw"):
            pk[dr], pk["t" + dr] = torch.empty(n, device=dev), torch.empty(n, device=dev)
            lib.call("mstts_persist_lstm_pack", lib.ptr(st["wh_" + dr]), 4 * H, lib.ptr(pk[dr]), lib.ptr(pk["t" + dr]))
        xch = torch.empty(L.mstts_persist_lstm_ws_bytes() // 4, device=dev)
        ctrl = torch.zeros(16, dtype=torch.int32, device=dev)
        hist = torch.empty(L.mstts_persist_lstm_hist_floats(T), device=dev)
        qs = _fwd_descs(st, o, None)
        lib.call("mstts_lstm_seq_fwd_pair_persistent", C.byref(qs[0]), C.byref(qs[1]), lib.ptr(pk["fw"]), lib.ptr(pk["bw"]), lib.ptr(xch), lib.ptr(ctrl), lib.ptr(hist))
        torch.cuda.synchronize()
        c = ctrl.cpu().numpy()
        assert c[1] == 0 and c[2] == 64, c[:4]
        o["pk"], o["xch"], o["ctrl"], o["hist"] = pk, xch, ctrl, hist
    else:
        extra = {"gates_" + dr: torch.empty(L.mstts_lstm_seq_ws_floats(B, H, 0), device=dev) for dr in ("fw", "bw")}
        qs = _fwd_descs(st, o, extra)
        lib.call("mstts_lstm_seq_fwd_pair", C.byref(qs[0]), C.byref(qs[1]))
        torch.cuda.synchronize()
    return o


def _run_bwd(dev, st, o, persistent):
    B, T = st["B"], st["T"]
    L = lib.load()
    r = {}
    qs = []
    for di, dr in enumerate(("fw", "bw")):
        r["dgs_" + dr] = torch.full((T, B, 4 * H), float("nan"), device=dev)
        r["dgp_" + dr] = torch.full((B, T, 4 * H), float("nan"), device=dev)
        q = lib.LstmSeqBwd()
        q.B, q.T, q.H = B, T, H
        q.wh = lib.ptr(st["wh_" + dr]); q.wh_ld = 4 * H
        q.lengths = lib.ptr(st["lens"]); q.reverse = di; q.zoneout = st["zoneout"]
        q.zc = lib.ptr(st["zc_" + dr]); q.zh = lib.ptr(st["zh_" + dr])
        q.d_out = lib.ptr(st["dout_" + dr]); q.dout_sb = T * H; q.dout_st = H
        q.c_hist = lib.ptr(o["c_" + dr]); q.acts = lib.ptr(o["acts_" + dr]); q.c_raw = lib.ptr(o["craw_" + dr])
        q.dgates_step = lib.ptr(r["dgs_" + dr]); q.dgates_pos = lib.ptr(r["dgp_" + dr])
        r["ws_" + dr] = torch.empty(L.mstts_lstm_seq_ws_floats(B, H, 1), device=dev)
        q.ws = lib.ptr(r["ws_" + dr])
        qs.append(q)
    if persistent:
        pk, xch, ctrl = o["pk"], o["xch"], o["ctrl"]
        r["bws"] = torch.empty(L.mstts_persist_lstm_bwd_floats(T), device=dev)
        lib.call("mstts_lstm_seq_bwd_pair_persistent", C.byref(qs[0]), C.byref(qs[1]), lib.ptr(pk["tfw"]), lib.ptr(pk["tbw"]), lib.ptr(xch), lib.ptr(ctrl),
                 lib.ptr(o["hist"]), lib.ptr(r["bws"]))
        torch.cuda.synchronize()
        c = ctrl.cpu().numpy()
        assert c[1] == 0 and c[2] == 32, c[:4]
    else:
        lib.call("mstts_lstm_seq_bwd_pair", C.byref(qs[0]), C.byref(qs[1]))
        torch.cuda.synchronize()
    return r


@pytest.mark.parametrize("B,T", [(32, 128), (5, 9), (17, 40), (1, 1)])
def test_persistent_bilstm_equals_launch_per_step(dev, B, T):
    st = _setup(dev, B, T, seed=100 + B)
    ref = _run_fwd(dev, st, False)
    got = _run_fwd(dev, st, True)
    for k in ("out", "c_fw", "h_fw", "acts_fw", "craw_fw", "c_bw", "h_bw", "acts_bw", "craw_bw"):
        a, b = t2n(got[k]), t2n(ref[k])
        assert np.isfinite(a).all(), k
        assert rel_err(a, b) < 2e-5, (k, rel_err(a, b))
    # BPTT: the launch-per-step pair on its own forward's row-major histories, the persistent launch on the packed history of its forward
    rb = _run_bwd(dev, st, ref, False)
    gb = _run_bwd(dev, st, got, True)
    for k in ("dgs_fw", "dgp_fw", "dgs_bw", "dgp_bw"):
        a, b = t2n(gb[k]), t2n(rb[k])
        if k.startswith("dgp"):                 # positions past a row's length are written by neither path in the reversed direction
            m = np.isfinite(b)
            assert np.array_equal(m, np.isfinite(a)), k
            a, b = np.where(m, a, 0.0), np.where(m, b, 0.0)
        assert np.isfinite(a).all(), k
        assert rel_err(a, b) < 5e-5, (k, rel_err(a, b))


def test_persistent_bilstm_is_deterministic_and_reusable(dev):
    """Two runs give identical bits; a second sequence through the same ring / control buffers (stale generations) is right as well."""
    st = _setup(dev, 32, 64, seed=7)
    a = _run_fwd(dev, st, True)
    b = _run_fwd(dev, st, True)
    assert np.array_equal(t2n(a["out"]), t2n(b["out"])) and np.array_equal(t2n(a["c_bw"]), t2n(b["c_bw"]))
    st2 = _setup(dev, 32, 64, seed=8)
    ref2 = _run_fwd(dev, st2, False)
    L = lib.load()
    o = _fwd_bufs(dev, 32, 64)
    qs = _fwd_descs(st2, o, None)
    for dr in ("fw", "bw"):
        lib.call("mstts_persist_lstm_pack", lib.ptr(st2["wh_" + dr]), 4 * H, lib.ptr(a["pk"][dr]), lib.ptr(a["pk"]["t" + dr]))
    lib.call("mstts_lstm_seq_fwd_pair_persistent", C.byref(qs[0]), C.byref(qs[1]), lib.ptr(a["pk"]["fw"]), lib.ptr(a["pk"]["bw"]), lib.ptr(a["xch"]), lib.ptr(a["ctrl"]),
             lib.ptr(a["hist"]))
    torch.cuda.synchronize()
    assert rel_err(t2n(o["out"]), t2n(ref2["out"])) < 2e-5


def test_train_step_with_and_without_the_persistent_encoder(dev, monkeypatch):
    """A whole train step at the loop's reference widths three ways: encoder recurrence persistent both ways, persistent forward with the
    BPTT falling back to the launch-per-step pair (which then reads the row-major history the unpack kernel wrote), and launch per step both
    ways: every gradient agrees."""
    from tests.test_gpu_persist import _engine
    from tests.helpers import to_dev
    from oracle import train as OT
    eng, od = _engine(dev)
    if not eng.persist_enc:
        pytest.skip("persistent BiLSTM launches not available on this device")
    B, Te, L = 8, 40, 6
    batch = to_dev(OT.synthetic_batch(od, B, Te, L, seed=21, ragged=True), dev)
    seed = OT.step_seed(1234, 0)
    w = eng.plan(B, Te, L)
    assert w.persist_enc

    def grads(fail_bptt=False):
        eng.forward(batch, w, seed=seed)
        if fail_bptt:
            eng.persist_enc_selftest = 1                  # the encoder BPTT launch's status reads "gave up": the backward pass re-runs launch by launch
        eng.loss_and_backward(w)
        torch.cuda.synchronize()
        return t2n(eng.params.grad).copy()

    a = grads()
    assert w.enc_hist_valid and eng.persist_enc_fallbacks == 0
    b = grads(fail_bptt=True)                             # persistent forward, launch-per-step BPTT (reads the row-major history the unpack kernel wrote)
    assert eng.persist_enc_fallbacks == 1
    w.persist_enc = False
    c = grads()                                           # launch per step both ways
    assert not w.enc_hist_valid
    scale = np.abs(c).max()
    assert np.abs(a - c).max() < 2e-4 * scale, np.abs(a - c).max() / scale
    assert np.abs(b - c).max() < 2e-4 * scale, np.abs(b - c).max() / scale
    # the persistent launches in the serial order (not on their own stream under the decoder-side products): the same gradients
    import multi_speaker_tts_amd.engine as E
    assert E.ENC_OVERLAP
    w.persist_enc = True
    monkeypatch.setattr(E, "ENC_OVERLAP", False)
    e = grads()
    assert w.enc_hist_valid and eng.persist_enc_fallbacks == 1
    assert np.abs(e - a).max() < 2e-5 * scale, np.abs(e - a).max() / scale


@pytest.mark.parametrize("B,T", [(70, 21), (320, 12), (33, 5)])
def test_persistent_single_sequence_row_groups(dev, B, T):
    """More than 32 rows of ONE unidirectional sequence (the speaker-encoder trainer's layers): ceil(B / 32) independent row groups in one
    launch, forward and BPTT, against mstts_lstm_seq_fwd / mstts_lstm_seq_bwd."""
    L = lib.load()
    if not L.mstts_persist_lstm_supported_n(B, H, 1):
        pytest.skip("not supported on this device")
    st = _setup(dev, B, T, seed=300 + B)
    out = {}
    for persistent in (False, True):
        o = {"out": torch.zeros(B, T, H, device=dev)}
        z = lambda *s: torch.full(s, float("nan"), device=dev)
        o["c"], o["h"], o["acts"], o["craw"] = z(T + 1, B, H), z(T + 1, B, H), z(T, B, 4 * H), z(T, B, H)
        q = lib.LstmSeqFwd()
        q.B, q.T, q.H = B, T, H
        q.xw = lib.ptr(st["xw_fw"]); q.wh = lib.ptr(st["wh_fw"]); q.wh_ld = 4 * H
        q.lengths = lib.ptr(st["lens"]); q.reverse = 0; q.zoneout = st["zoneout"]
        q.zc = lib.ptr(st["zc_fw"]); q.zh = lib.ptr(st["zh_fw"])
        q.out = lib.ptr(o["out"]); q.out_sb = T * H; q.out_st = H
        q.c_hist = lib.ptr(o["c"]); q.h_hist = lib.ptr(o["h"]); q.acts = lib.ptr(o["acts"]); q.c_raw = lib.ptr(o["craw"])
        o["dgs"], o["dgp"] = z(T, B, 4 * H), z(B, T, 4 * H)
        qb = lib.LstmSeqBwd()
        qb.B, qb.T, qb.H = B, T, H
        qb.wh = lib.ptr(st["wh_fw"]); qb.wh_ld = 4 * H
        qb.lengths = lib.ptr(st["lens"]); qb.reverse = 0; qb.zoneout = st["zoneout"]
        qb.zc = lib.ptr(st["zc_fw"]); qb.zh = lib.ptr(st["zh_fw"])
        qb.d_out = lib.ptr(st["dout_fw"]); qb.dout_sb = T * H; qb.dout_st = H
        qb.c_hist = lib.ptr(o["c"]); qb.acts = lib.ptr(o["acts"]); qb.c_raw = lib.ptr(o["craw"])
        qb.dgates_step = lib.ptr(o["dgs"]); qb.dgates_pos = lib.ptr(o["dgp"])
        if persistent:
            n = L.mstts_persist_lstm_pack_floats()
            pk, pkt = torch.empty(n, device=dev), torch.empty(n, device=dev)
            lib.call("mstts_persist_lstm_pack", lib.ptr(st["wh_fw"]), 4 * H, lib.ptr(pk), lib.ptr(pkt))
            xch = torch.empty(L.mstts_persist_lstm_ws_bytes_n(B, 1) // 4, device=dev)
            ctrl = torch.zeros(16, dtype=torch.int32, device=dev)
            hist = torch.empty(L.mstts_persist_lstm_hist_floats_n(T, B, 1), device=dev)
            bws = torch.empty(L.mstts_persist_lstm_bwd_floats_n(T, B, 1), device=dev)
            lib.call("mstts_lstm_seq_fwd_persistent", C.byref(q), lib.ptr(pk), lib.ptr(xch), lib.ptr(ctrl), lib.ptr(hist))
            torch.cuda.synchronize()
            groups = (B + 31) // 32
            c = ctrl.cpu().numpy()
            assert c[1] == 0 and c[2] == 32 * groups, c[:4]
            lib.call("mstts_lstm_seq_bwd_persistent", C.byref(qb), lib.ptr(pkt), lib.ptr(xch), lib.ptr(ctrl), lib.ptr(hist), lib.ptr(bws))
            torch.cuda.synchronize()
            c = ctrl.cpu().numpy()
            assert c[1] == 0 and c[2] == 16 * groups, c[:4]
            o["keep"] = (pk, pkt, xch, ctrl, hist, bws)
        else:
            o["gates"] = torch.empty(L.mstts_lstm_seq_ws_floats(B, H, 0), device=dev)
            o["ws"] = torch.empty(L.mstts_lstm_seq_ws_floats(B, H, 1), device=dev)
            q.gates_ws = lib.ptr(o["gates"]); qb.ws = lib.ptr(o["ws"])
            lib.call("mstts_lstm_seq_fwd", C.byref(q))
            lib.call("mstts_lstm_seq_bwd", C.byref(qb))
            torch.cuda.synchronize()
        out[persistent] = o
    for k in ("out", "c", "h", "acts", "craw", "dgs", "dgp"):
        a, b = t2n(out[True][k]), t2n(out[False][k])
        assert np.isfinite(a).all(), k
        assert rel_err(a, b) < 5e-5, (k, rel_err(a, b))


@pytest.mark.parametrize("B,T,HH,ndir", [(16, 50, 128, 2), (5, 9, 128, 2), (1, 1, 128, 2), (16, 401, 128, 2), (40, 12, 128, 1), (80, 64, 256, 1)])
def test_forward_only_launch_h128_and_single_direction(dev, B, T, HH, ndir):
    """The forward launch at H = 128 (the Taco1 vocoder's BiRNN, Taco1_Mel_to_Spect/Modules.py:75-99) and as a single direction over row groups
    (the speaker-encoder stack, Speaker_Embedding/Modules.py:15-37), inference arithmetic (no keep-masks: state' = 0.9 new + 0.1 old),
    against the launch-per-step drivers on the same inputs."""
    L = lib.load()
    if not L.mstts_persist_lstm_fwd_supported_n(B, HH, ndir):
        pytest.skip("persistent LSTM launches not available on this device")
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + T)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
    lens = torch.randint(max(1, T // 3), T + 1, (B,), generator=g, dtype=torch.int32)
    lens[0] = T
    lens = lens.to(dev)
    wh = [r(HH, 4 * HH, scale=0.08) for _ in range(ndir)]
    xw = [r(B, T, 4 * HH, scale=0.8) for _ in range(ndir)]

    def run(persistent):
        out = torch.zeros(B, T, ndir * HH, device=dev)
        keep, qs = [], []
        for di in range(ndir):
            q = lib.LstmSeqFwd()
            q.B, q.T, q.H = B, T, HH
            q.xw = lib.ptr(xw[di]); q.wh = lib.ptr(wh[di]); q.wh_ld = 4 * HH
            q.lengths = lib.ptr(lens); q.reverse = di; q.zoneout = 0.1
            q.out = lib.ptr(out, di * HH); q.out_sb = T * ndir * HH; q.out_st = ndir * HH
            ch, hh = torch.full((T + 1, B, HH), float("nan"), device=dev), torch.full((T + 1, B, HH), float("nan"), device=dev)
            q.c_hist, q.h_hist = lib.ptr(ch), lib.ptr(hh)
            gw = torch.empty(L.mstts_lstm_seq_ws_floats(B, HH, 0), device=dev)
            q.gates_ws = lib.ptr(gw)
            keep += [ch, hh, gw]
            qs.append(q)
        if persistent:
            pk = [torch.empty(HH * 4 * HH, device=dev) for _ in range(ndir)]
            for di in range(ndir):
                lib.call("mstts_persist_lstm_pack_fwd", lib.ptr(wh[di]), 4 * HH, HH, lib.ptr(pk[di]))
            xch = torch.empty(L.mstts_persist_lstm_ws_bytes_n(B, ndir) // 4, device=dev)
            ctrl = torch.zeros(16, dtype=torch.int32, device=dev)
            hist = torch.empty(L.mstts_persist_lstm_hist_floats_n(T, B, ndir), device=dev)
            if ndir == 2:
                lib.call("mstts_lstm_seq_fwd_pair_persistent", C.byref(qs[0]), C.byref(qs[1]), lib.ptr(pk[0]), lib.ptr(pk[1]), lib.ptr(xch), lib.ptr(ctrl), lib.ptr(hist))
            else:
                lib.call("mstts_lstm_seq_fwd_persistent", C.byref(qs[0]), lib.ptr(pk[0]), lib.ptr(xch), lib.ptr(ctrl), lib.ptr(hist))
            torch.cuda.synchronize()
            c = ctrl.cpu().numpy()
            assert c[1] == 0 and c[2] == ndir * ((B + 31) // 32) * (HH // 8), c[:4]
        else:
            if ndir == 2:
                lib.call("mstts_lstm_seq_fwd_pair", C.byref(qs[0]), C.byref(qs[1]))
            else:
                lib.call("mstts_lstm_seq_fwd", C.byref(qs[0]))
            torch.cuda.synchronize()
        return t2n(out), [t2n(keep[3 * di]) for di in range(ndir)], [t2n(keep[3 * di + 1]) for di in range(ndir)]

    oa, ca, ha = run(True)
    ob, cb, hb = run(False)
    tol = 2e-4 if T > 100 else 2e-5
    assert np.isfinite(oa).all() and rel_err(oa, ob) < tol, rel_err(oa, ob)
    for di in range(ndir):
        assert rel_err(ca[di][1:], cb[di][1:]) < tol and rel_err(ha[di][1:], hb[di][1:]) < tol, di


# ---------------------------------------------------------------------------------------------------------------- host-side contracts (round 5)
REFW_SPK = dict(spk=256, spk_lstm=256, n_mel=80)


def _spk_setup(dev, B=3, seed=3):
    from multi_speaker_tts_amd.inference import InferEngine
    from multi_speaker_tts_amd import lib
    pd, od = dims_pair(**REFW_SPK)
    values = OM.init_params(od, seed)
    g = np.random.default_rng(seed + 1)
    NB = B * od.spk_samples
    mel = torch.tensor(np.clip(g.normal(0, 1.5, (NB, od.spk_frames, od.n_mel)), -4, 4).astype(np.float32), device=dev)
    eng = InferEngine(pd, device=dev, values=values)
    if not lib.load().mstts_persist_lstm_fwd_supported_n(NB, pd.spk_lstm, 1):
        pytest.skip("persistent LSTM launches not available on this device")
    return eng, pd, od, values, mel


def test_guarded_subgraph_retries_launch_by_launch(dev):
    """InferEngine._guarded: a persistent LSTM launch of a sub-graph called on its own that gave up (self-test knob) is re-run launch by
    launch, with one warning, and gives the launch-per-step result; the next call is healthy again."""
    import warnings
    eng, pd, od, values, mel = _spk_setup(dev)
    ref = t2n(eng.speaker_embedding(mel)).copy()
    n0 = eng.persist_lstm_launches
    assert n0 == pd.spk_lstm_n and eng.persist_lstm_fallbacks == 0
    eng.persist_lstm = False
    plain = t2n(eng.speaker_embedding(mel)).copy()
    eng.persist_lstm = True
    assert rel_err(ref, plain) < 1e-4
    eng.persist_lstm_selftest = 1
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        got = t2n(eng.speaker_embedding(mel)).copy()
    assert eng.persist_lstm_fallbacks == 1 and sum("re-running launch by launch" in str(r.message) for r in rec) == 1
    assert np.array_equal(got, plain)                                 # the retry IS the launch-per-step path
    again = t2n(eng.speaker_embedding(mel)).copy()
    assert eng.persist_lstm_fallbacks == 1 and np.array_equal(again, ref)


def test_deferred_speaker_ticket_is_redeemed_by_the_train_forward(dev):
    """MSTTS_SV.py:49-56,211: the frozen speaker stack runs in front of every train step.  With defer=True its persistent launches are not
    waited for; TrainEngine.forward redeems the ticket at its own sync point.  A failed launch (self-test knob) makes the forward recompute
    the embedding launch by launch and run its pass again: same outputs as the healthy step, one redo counted, BN moving statistics updated once."""
    from multi_speaker_tts_amd.engine import TrainEngine
    eng, pd, od, values, mel = _spk_setup(dev, B=4)
    tr = TrainEngine(pd, device=dev, values=values)
    B, Te, L = 4, 9, 6
    batch = to_dev(OT.synthetic_batch(od, B, Te, L, seed=5, ragged=True), dev)
    w = tr.plan(B, Te, L)

    def step(fail):
        tr.params.load(values)
        tr._derived_stale = True
        emb, ticket = eng.speaker_embedding(mel, defer=True)
        assert ticket is not None
        b = dict(batch)
        b["Speaker_Embedding"] = emb.clone()
        b["_speaker_ticket"] = ticket
        if fail:
            eng.persist_lstm_selftest = 1
            b["Speaker_Embedding"].fill_(7.0)                          # what a launch that gave up leaves behind: junk
        tr.forward(b, w, seed=11)
        torch.cuda.synchronize()
        assert "_speaker_ticket" not in b
        return t2n(w.mel_out).copy(), t2n(tr.params.frozen[:tr.params.n_moving]).copy(), t2n(b["Speaker_Embedding"]).copy()

    mel0, mov0, e0 = step(False)
    assert tr.speaker_ticket_redos == 0
    mel1, mov1, e1 = step(True)
    assert tr.speaker_ticket_redos == 1 and eng.persist_lstm_fallbacks == 1 and eng.persist_lstm_selftest == 0
    assert rel_err(e1, e0) < 1e-4 and rel_err(mel1, mel0) < 1e-4
    assert rel_err(mov1, mov0) < 1e-5                                  # one moving-statistics update per step, not two


def test_inference_after_a_trainer_step_sees_the_new_kernels(dev):
    """Speaker_Embedding.Inference() keeps one InferEngine on the trainer's ParamStore; its packed recurrent kernels are cached per
    ParamStore.version, so EVERY writer of the variables has to bump it (ParamStore.touch): inference -> trainer step -> inference must
    equal the launch-per-step forward on the updated variables."""
    from multi_speaker_tts_amd.speaker_trainer import SpeakerTrainEngine
    from multi_speaker_tts_amd.inference import InferEngine
    eng, pd, od, values, mel = _spk_setup(dev)
    tr = SpeakerTrainEngine(pd, device=dev, values=values)
    inf = InferEngine(pd, device=dev, params=tr.params)
    before = t2n(inf.speaker_embedding(mel)).copy()
    g = np.random.default_rng(1)
    N, T = 12, 9
    x = torch.tensor(np.clip(g.normal(0, 1.5, (N, T, od.n_mel)), -4, 4).astype(np.float32), device=dev)
    v0 = tr.params.version
    for _ in range(3):
        tr.train_step(x, 3)
    assert tr.params.version == v0 + 3
    after = t2n(inf.speaker_embedding(mel)).copy()
    inf.persist_lstm = False
    plain = t2n(inf.speaker_embedding(mel)).copy()
    assert rel_err(after, plain) < 1e-4, rel_err(after, plain)
    assert rel_err(after, before) > 1e-4                               # the step really moved the variables


def test_two_deferred_tickets_outstanding_together(dev):
    """every deferred ticket used to share ONE page-locked read-back block and slot numbering restarts at 0 per call, so a second
    speaker_embedding(defer=True) before the first ticket was redeemed (batch prefetch) overwrote the words the first ticket reads.  Now a
    ticket owns its block: the first call's launch is made to fail ON THE DEVICE (its control words overwritten behind it), the second is
    healthy, and each ticket reports its own launches."""
    eng, pd, od, values, mel = _spk_setup(dev, B=4)
    emb1, t1 = eng.speaker_embedding(mel, defer=True)
    assert t1 is not None and len(t1.pending) >= 1
    torch.cuda.synchronize()
    # call 1's read-back has landed in ITS block; now damage that block the way a launch that gave up would have: abort code 3
    t1.host[t1.pending[0][0]][1] = 3
    emb2, t2 = eng.speaker_embedding(mel, defer=True)                  # reuses device slots 0.., reads back into ANOTHER block
    assert t2 is not None and t2.host is not t1.host
    blocks = (t1.host, t2.host)
    assert t2.ok() is True                                             # the healthy call is not blamed for the first one's words
    assert t1.ok() is False                                            # ... and the failed one is not absolved by the second one's
    with pytest.raises(RuntimeError):
        t1.ok()                                                        # a ticket is redeemed once
    assert len(eng._deferred_host_pool) == 2
    emb3, t3 = eng.speaker_embedding(mel, defer=True)                  # blocks are pooled, not re-allocated
    assert t3.host is blocks[0] or t3.host is blocks[1]
    assert t3.ok() and np.array_equal(t2n(emb3), t2n(emb2))
