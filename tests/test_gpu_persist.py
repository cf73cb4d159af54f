"""The persistent decoder loop (csrc/persist.hip, mstts_decoder_train_fwd_persistent: all S steps of Modules.py:397-443 in one
launch) against the launch-per-step loop it replaces (mstts_decoder_train_fwd) on the same engine, inputs and keep-masks: every
tensor the BPTT reads must agree.  Parity with the oracle at these widths is test_gpu_model.py::test_train_step_parity (its two
reference-width cases take the persistent path) and tests/test_gpu_depth.py::test_depth_parity_train."""
import numpy as np
import pytest
import torch

from helpers import dims_pair, rel_err, t2n, to_dev
from oracle import model as OM, train as OT
from multi_speaker_tts_amd.engine import TrainEngine

pytestmark = pytest.mark.gpu

# reference widths where the loop sees them (decoder cells 1024, memory 768 = 2 x 256 + 256, attention 128 / 31 taps); the rest reduced
WIDE = dict(emb=64, enc_conv_ch=64, enc_lstm=256, spk=256, prenet=256, dec_lstm=1024, n_mel=16, post_ch=32)      # (prenet 256: the launch forms the prenet rows' product itself)
HIST = ("in0", "in1", "pj", "c0", "c1", "acts0", "acts1", "craw0", "craw1", "q_hist", "align_hist", "cum_hist")


def _engine(dev, seed=3):
    pd, od = dims_pair(**WIDE)
    values = OM.init_params(od, seed)
    g = np.random.default_rng(seed + 1)
    for k in values:
        if k.endswith(("bias", "bias_b")):
            values[k] = g.normal(0, 0.1, values[k].shape)
    eng = TrainEngine(pd, device=dev, values=values)
    if not eng.persist:
        pytest.skip("persistent loop not available on this device (needs 256 CUs and one workgroup per CU)")
    return eng, od


def _snapshot(w, eng=None):
    if eng is not None:
        eng.unpack_history(w)            # the persistent forward keeps the cell operands packed by owner; the comparison reads the histories
    out = {k: t2n(getattr(w, k)).copy() for k in HIST + ("linear", "mel_out", "stop")}
    out["c0"], out["c1"] = out["c0"][:-1], out["c1"][:-1]      # (slot S, the state behind the last step, is no BPTT operand: the packed form does not carry it)
    # in1[s] = [m0_s | h1_{s-1}]: slot S holds h1 of the last step and NO cell-0 output (there is no step S) - that half is never written by
    # either loop and never read; the workspace arena hands out memory that is not zero (NaN under the tests), so it is not compared either
    H = out["in1"].shape[-1] // 2
    out["in1_slot_S_h1"] = out["in1"][-1][:, H:].copy()
    out["in1"] = out["in1"][:-1]
    return out


@pytest.mark.parametrize("B,Te,L,ragged", [(32, 128, 9, False), (8, 40, 12, True), (5, 128, 3, True), (1, 7, 2, False), (32, 100, 60, True),
                                                (32, 129, 5, True), (8, 160, 12, True), (32, 200, 9, False), (32, 256, 6, True), (3, 131, 40, True)])       # > 128 tokens: the 256-position instantiation (Feeder.py:148-152 pads a batch to its longest text)
def test_persistent_equals_launch_per_step(dev, B, Te, L, ragged):
    eng, od = _engine(dev)
    batch = to_dev(OT.synthetic_batch(od, B, Te, L, seed=5, ragged=ragged), dev)
    seed = OT.step_seed(1234, 0)
    w = eng.plan(B, Te, L)
    assert w.persist
    eng.forward(batch, w, seed=seed)
    torch.cuda.synchronize()
    st = w.pctrl.cpu().numpy()
    assert eng.persist_fallbacks == 0 and st[1] == 0 and st[2] == 256, st[:3]
    a = _snapshot(w, eng)
    for k in HIST:                                       # the second pass must really write them again
        getattr(w, k).zero_()
    w.persist = False
    eng.forward(batch, w, seed=seed)
    torch.cuda.synchronize()
    b = _snapshot(w, eng)
    bad = {}
    for k in a:
        assert np.isfinite(a[k]).all() and np.isfinite(b[k]).all(), k
        e = rel_err(a[k], b[k])
        if e > (5e-4 if L > 20 else 5e-5):               # same fp32 products, different summation order (and atomics upstream of the loop: not bit-reproducible run to run)
            bad[k] = e
    assert not bad, bad


def test_persistent_is_deterministic(dev):
    """The launch itself, twice on identical inputs (the decoder descriptors of the plan, called directly - upstream of the loop the
    forward pass contains reductions with atomics): bit-identical histories, i.e. no summation order depends on arrival order."""
    import ctypes as C
    from multi_speaker_tts_amd import lib
    eng, od = _engine(dev)
    batch = to_dev(OT.synthetic_batch(od, 32, 128, 7, seed=9, ragged=True), dev)
    w = eng.plan(32, 128, 7)
    for k in HIST:                       # (histories the launch does not write when it packs the cell operands instead - acts, craw, c - hold whatever the
        getattr(w, k).zero_()            #  arena held: give the first pass the zeros the later passes get)
    eng.forward(batch, w, seed=77)
    torch.cuda.synchronize()
    keys = HIST + (("opk",) if w.opk_valid else ())
    a = {k: t2n(getattr(w, k)).copy() for k in keys}
    for _ in range(3):
        for k in keys:
            getattr(w, k).zero_()
        lib.call("mstts_decoder_train_fwd_persistent", C.byref(w.dec), C.byref(w.pdesc))
        torch.cuda.synchronize()
        st = w.pctrl.cpu().numpy()
        assert st[1] == 0 and st[2] == 256, st[:3]
        for k in keys:
            assert np.array_equal(a[k], t2n(getattr(w, k))), k
    assert eng.persist_fallbacks == 0


def test_persistent_abort_falls_back(dev):
    """Self-test knob: workgroup 0 raises the abort word at step 3 and leaves; every other workgroup finds the word while it waits for
    that workgroup's data, the launch ends early, and the engine re-runs the sequence on the launch-per-step path."""
    eng, od = _engine(dev)
    B, Te, L = 8, 40, 10
    batch = to_dev(OT.synthetic_batch(od, B, Te, L, seed=5, ragged=True), dev)
    w = eng.plan(B, Te, L)
    eng.persist_selftest = 4
    mov0 = eng.params.frozen[:eng.params.n_moving].clone()
    assert eng.params.n_moving > 0
    eng.forward(batch, w, seed=11)
    torch.cuda.synchronize()
    assert eng.persist_fallbacks == 1 and eng.persist_last_status[1] == 3 and eng.persist_last_status[2] < 256
    a = _snapshot(w, eng)
    mov_a = eng.params.frozen[:eng.params.n_moving].clone()
    eng.persist_selftest = 0
    w.persist = False
    eng.params.frozen[:eng.params.n_moving].copy_(mov0)
    eng.forward(batch, w, seed=11)
    torch.cuda.synchronize()
    b = _snapshot(w, eng)
    # the BN moving statistics took ONE update (Modules.py:37-40: one update op per train step), not one from the aborted launch's junk
    # outputs and a second from the re-run (postnet layers and the vocoder conv bank sit behind the decoder loop)
    mov_b = eng.params.frozen[:eng.params.n_moving]
    assert not torch.equal(mov_b, mov0)
    assert torch.allclose(mov_a, mov_b, rtol=1e-5, atol=1e-7), float((mov_a - mov_b).abs().max())
    for k in a:                                          # the fallback IS the launch-per-step path (what is upstream of the loop
        assert rel_err(a[k], b[k]) < 5e-5, k             # holds atomic reductions, so equal to rounding, not to the bit)
    w.persist = True
    eng.forward(batch, w, seed=11)                       # and the next launch is healthy again
    torch.cuda.synchronize()
    assert eng.persist_fallbacks == 1
    c = _snapshot(w, eng)
    assert rel_err(c["pj"], b["pj"]) < 2e-5


def test_encoder_launch_failure_reruns_the_pass(dev):
    """The encoder's persistent launches are checked together with the decoder's at the END of a pass (one host sync per pass).  A failed encoder
    launch (test knob: the next check reads a give-up code) therefore means the whole pass ran on junk: it is run again launch by launch - same
    results as an undisturbed pass, the BN moving statistics updated once, one fallback counted; the same for the backward pass."""
    eng, od = _engine(dev)
    if not eng.persist_enc:
        pytest.skip("persistent encoder launches not available on this device")
    B, Te, L = 8, 40, 6
    batch = to_dev(OT.synthetic_batch(od, B, Te, L, seed=5, ragged=True), dev)
    w = eng.plan(B, Te, L)
    mov0 = eng.params.frozen[:eng.params.n_moving].clone()
    eng.forward(batch, w, seed=11); eng.loss_and_backward(w)
    torch.cuda.synchronize()
    ref, gref, mov_ref = _snapshot(w, eng), eng.params.grad.clone(), eng.params.frozen[:eng.params.n_moving].clone()
    assert eng.persist_enc_fallbacks == 0 and w.enc_hist_valid
    # forward
    eng.params.frozen[:eng.params.n_moving].copy_(mov0)
    eng.persist_enc_selftest = 1
    eng.forward(batch, w, seed=11)
    assert eng.persist_enc_fallbacks == 1 and not w.enc_hist_valid and not w.persist_now        # re-run without the persistent launches
    eng.loss_and_backward(w)
    torch.cuda.synchronize()
    got = _snapshot(w, eng)
    for k in ("pj", "align_hist", "linear", "mel_out"):
        assert rel_err(got[k], ref[k]) < 5e-5, k
    assert rel_err(t2n(eng.params.grad), t2n(gref)) < 2e-4
    assert torch.allclose(eng.params.frozen[:eng.params.n_moving], mov_ref, rtol=1e-5, atol=1e-7)
    # backward: a healthy forward, then the encoder BPTT launch "fails"
    eng.forward(batch, w, seed=11)
    assert w.enc_hist_valid and w.persist_now
    eng.persist_enc_selftest = 1
    eng.loss_and_backward(w)
    torch.cuda.synchronize()
    assert eng.persist_enc_fallbacks == 2 and eng.persist_bwd_fallbacks == 0
    assert rel_err(t2n(eng.params.grad), t2n(gref)) < 2e-4


def test_adaptive_fallback_cooldown(dev, monkeypatch):
    """Policy (engine._persist_begin_step): two consecutive steps whose persistent launch gave up switch the persistent plans off for a
    cool-down (one warning), the steps in between run the launch-per-step loops without paying a rendezvous, then the launches are
    probed again."""
    import warnings
    from multi_speaker_tts_amd import engine as E
    monkeypatch.setattr(E, "PERSIST_COOLDOWN", 3)
    eng, od = _engine(dev)
    B, Te, L = 8, 40, 6
    batch = to_dev(OT.synthetic_batch(od, B, Te, L, seed=5, ragged=True), dev)
    w = eng.plan(B, Te, L)
    eng.forward(batch, w, seed=11)
    ref = _snapshot(w, eng)
    eng.persist_selftest = 2
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        eng.forward(batch, w, seed=11)                   # strike 1
        eng.forward(batch, w, seed=11)                   # strike 2
        assert eng.persist_fallbacks == 2 and eng.persist_disabled_steps == 0
        for i in range(3):                               # cool-down: not even attempted (the self-test knob would fail them)
            eng.forward(batch, w, seed=11)
            assert not w.persist_now and eng.persist_fallbacks == 2 and eng.persist_disabled_steps == i + 1
        assert rel_err(_snapshot(w, eng)["pj"], ref["pj"]) < 5e-5
        eng.persist_selftest = 0
        eng.forward(batch, w, seed=11)                   # probe: healthy again
        torch.cuda.synchronize()
        assert w.persist_now and eng.persist_fallbacks == 2 and eng.persist_disabled_steps == 3
        assert rel_err(_snapshot(w, eng)["pj"], ref["pj"]) < 5e-5
    assert sum("consecutive steps fell back" in str(r.message) for r in rec) == 1


@pytest.mark.parametrize("park_us", [400, 30000])
def test_intruder_on_the_compute_units(dev, park_us):
    """Another tenant holds 16 CUs (mstts_debug_park_cus on a side stream: 96 KB of LDS per workgroup, nothing of the persistent
    launches fits beside it) while a train step's persistent launches want all 256.  Short stay (0.4 ms, inside the 2 ms rendezvous
    bound): the launch waits and runs.  Long stay (30 ms): the rendezvous gives up, the step runs the launch-per-step loop.  Either way
    the step finishes with the result of the undisturbed step; wall times are recorded."""
    import json, os, time
    from multi_speaker_tts_amd import lib
    eng, od = _engine(dev)
    B, Te, L = 16, 64, 30
    batch = to_dev(OT.synthetic_batch(od, B, Te, L, seed=5, ragged=True), dev)
    w = eng.plan(B, Te, L)
    eng.forward(batch, w, seed=11); eng.loss_and_backward(w)
    torch.cuda.synchronize()
    ref = _snapshot(w, eng)
    gref = eng.params.grad.clone()
    t0 = time.perf_counter()
    eng.forward(batch, w, seed=11); eng.loss_and_backward(w)
    torch.cuda.synchronize()
    quiet_ms = (time.perf_counter() - t0) * 1e3
    assert eng.persist_fallbacks == 0 and eng.persist_bwd_fallbacks == 0
    side = torch.cuda.Stream(device=dev, priority=-1)            # (a high-priority stream gets a hardware queue of its own where the runtime has one left)
    done = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        lib.call("mstts_debug_park_cus", 16, park_us, lib.ptr(done))
    if park_us >= 2500:
        time.sleep(0.003)            # the long-stay tenant must be ON its CUs before the step is enqueued (streams give no ordering between the two)
    eng.forward(batch, w, seed=11)
    fwd_fb = eng.persist_fallbacks
    eng.loss_and_backward(w)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    print("intruder %d us on 16 CUs: quiet step %.2f ms, disturbed %.2f ms, forward fallbacks %d, plan persistent %r / used %r, last status %r"
          % (park_us, quiet_ms, ms, fwd_fb, w.persist, w.persist_now, getattr(eng, "persist_last_status", None)))
    assert int(done.item()) == 16
    got = _snapshot(w, eng)
    for k in ("pj", "align_hist", "linear", "mel_out"):
        assert rel_err(got[k], ref[k]) < 5e-5, k
    assert rel_err(t2n(eng.params.grad), t2n(gref)) < 2e-4
    if park_us >= 2500:
        if fwd_fb == 0 and ms >= park_us * 1e-3:
            # HIP multiplexes streams onto a few hardware queues; late in a long test session the side stream can share the compute stream's
            # queue, and then the tenant simply ran BEFORE the step (serialised, nothing to test).  Seen in the full-suite run, never alone.
            pytest.skip("the side stream shares a hardware queue with the compute stream here: the tenant ran before the step, not beside it")
        assert fwd_fb == 1, "a tenant that outstays the rendezvous bound must end in the launch-per-step loop"
    else:
        assert fwd_fb == 0, "a tenant that leaves inside the rendezvous bound must only delay the launch"
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(root):
        with open(os.path.join(root, "intruder.jsonl"), "a") as f:
            f.write(json.dumps(dict(park_us=park_us, quiet_step_ms=quiet_ms, disturbed_step_ms=ms, forward_fallbacks=fwd_fb,
                                    bptt_fallbacks=eng.persist_bwd_fallbacks, enc_fallbacks=eng.persist_enc_fallbacks)) + "\n")
    print("intruder %d us on 16 CUs: quiet step %.2f ms, disturbed %.2f ms, forward fallbacks %d" % (park_us, quiet_ms, ms, fwd_fb))


def test_persistent_train_steps(dev):
    """Three optimizer steps through the persistent forward (kernels re-packed after every Adam update): loss scalars equal those of
    an engine that never uses it."""
    pd, od = dims_pair(**WIDE)
    values = OM.init_params(od, 3)
    batch = to_dev(OT.synthetic_batch(od, 16, 64, 20, seed=5, ragged=True), dev)
    out = []
    for persist in (True, False):
        eng = TrainEngine(pd, device=dev, values=values)
        if persist and not eng.persist:
            pytest.skip("persistent loop not available on this device")
        eng.persist = eng.persist and persist
        losses = []
        for _ in range(3):
            w = eng.train_step(batch)
            losses.append(eng.scalars(w)["Loss"])
        out.append((losses, eng.persist_fallbacks))
    assert out[0][1] == 0
    assert np.allclose(out[0][0], out[1][0], rtol=2e-4), out


BWD = ("dg0", "dg1", "dq_hist", "de_hist")


def _bwd_snapshot(eng, w):
    M = eng.d.mem
    out = {k: t2n(getattr(w, k)).copy() for k in BWD}
    parts = 1 if w.persist_bwd else w.d_in0_parts
    out["d_ctx"] = t2n(w.d_in0[:parts, 1:, :, :M].sum(0)).copy()          # gradient of the context rows from cell 0 of the next step
    out["grad"] = t2n(eng.params.grad).copy()
    return out


@pytest.mark.parametrize("B,Te,L,ragged", [(32, 128, 9, False), (8, 40, 12, True), (5, 128, 3, True), (1, 7, 2, False), (32, 100, 60, True),
                                                (32, 129, 5, True), (8, 160, 12, True), (32, 200, 9, False), (32, 256, 6, True), (3, 131, 40, True)])       # > 128 tokens: the 256-position instantiation (Feeder.py:148-152 pads a batch to its longest text)
def test_persistent_bptt_equals_launch_per_step(dev, B, Te, L, ragged):
    """mstts_decoder_train_bwd_persistent (the whole BPTT in one launch) against mstts_decoder_train_bwd on the same forward state:
    gate gradients, query / energy gradients, the context gradient and every parameter gradient of the step."""
    eng, od = _engine(dev)
    if not eng.persist_bwd:
        pytest.skip("persistent BPTT not available on this device")
    batch = to_dev(OT.synthetic_batch(od, B, Te, L, seed=5, ragged=ragged), dev)
    seed = OT.step_seed(1234, 0)
    w = eng.plan(B, Te, L)
    assert w.persist_bwd
    eng.forward(batch, w, seed=seed)
    eng.loss_and_backward(w)
    torch.cuda.synchronize()
    st = w.pctrl_b.cpu().numpy()
    assert eng.persist_bwd_fallbacks == 0 and st[1] == 0 and st[2] == 256, st[:3]
    a = _bwd_snapshot(eng, w)
    for k in BWD:
        getattr(w, k).zero_()
    w.d_in0.zero_()
    w.persist_bwd = False
    eng.loss_and_backward(w)
    torch.cuda.synchronize()
    b = _bwd_snapshot(eng, w)
    bad = {}
    for k in a:
        assert np.isfinite(a[k]).all() and np.isfinite(b[k]).all(), k
        e = rel_err(a[k], b[k])
        if e > (5e-4 if L > 20 else 5e-5):
            bad[k] = e
    assert not bad, bad


def test_persistent_bptt_is_deterministic_and_falls_back(dev):
    import ctypes as C
    from multi_speaker_tts_amd import lib
    eng, od = _engine(dev)
    if not eng.persist_bwd:
        pytest.skip("persistent BPTT not available on this device")
    B, Te, L = 32, 128, 7
    batch = to_dev(OT.synthetic_batch(od, B, Te, L, seed=9, ragged=True), dev)
    w = eng.plan(B, Te, L)
    eng.forward(batch, w, seed=77)
    eng.loss_and_backward(w)
    torch.cuda.synchronize()
    a = {k: t2n(getattr(w, k)).copy() for k in BWD}
    for _ in range(2):                                   # the launch alone, on identical inputs: bit-identical outputs
        for k in BWD:
            getattr(w, k).zero_()
        lib.call("mstts_decoder_train_bwd_persistent", C.byref(w.dec_b), C.byref(w.pdesc_b))
        torch.cuda.synchronize()
        st = w.pctrl_b.cpu().numpy()
        assert st[1] == 0 and st[2] == 256, st[:3]
        for k in BWD:
            assert np.array_equal(a[k], t2n(getattr(w, k))), k
    ref = t2n(eng.params.grad).copy()
    eng.persist_bwd_selftest = 3                         # abort at step index 2 of the BPTT: the packed operands are unpacked and the
    eng.forward(batch, w, seed=77)                       # launch-per-step loop takes over
    eng.loss_and_backward(w)
    torch.cuda.synchronize()
    eng.persist_bwd_selftest = 0
    assert eng.persist_bwd_fallbacks == 1 and eng.persist_last_status[1] == 3
    assert rel_err(t2n(eng.params.grad), ref) < 5e-5


def test_persistent_launches_back_to_back_on_changed_inputs(dev):
    """Two persistent launches in direct succession on the SAME buffers with DIFFERENT data (nothing in between but one small
    elementwise kernel): whatever the first launch left in the hand-off rings - in memory or in an XCD's L2 (the same-XCD publish path
    keeps pieces there) - must not be taken for data of the second.  Forward: the hoisted prenet product halved in place; BPTT: the
    projection's input gradient halved in place; each second launch against the launch-per-step loop on the same changed inputs."""
    import ctypes as C
    from multi_speaker_tts_amd import lib
    eng, od = _engine(dev)
    B, Te, L = 32, 128, 24
    batch = to_dev(OT.synthetic_batch(od, B, Te, L, seed=13, ragged=True), dev)
    w = eng.plan(B, Te, L)
    eng.forward(batch, w, seed=5)                         # launch A (inside), on the original inputs
    w.pre_d[-1].mul_(0.5)                                 # changed input of launch B (the folded form reads the prenet output, the
    od_ = eng.d                                           # launch-per-step loop below its hoisted product xw0 = pre . W0[:P] + b0)
    k0, o0 = eng.P("decoder/decoder/attention_wrapper/multi_rnn_cell/cell_0/zoneout_lstm_cell/kernel")
    b0, ob0 = eng.P("decoder/decoder/attention_wrapper/multi_rnn_cell/cell_0/zoneout_lstm_cell/bias")
    lib.gemm(w.pre_d[-1], k0, w.xw0, w.S * B, 4 * od_.dec_lstm, od_.prenet, od_.prenet, 4 * od_.dec_lstm, 4 * od_.dec_lstm, bias=b0, b_off=o0, bias_off=ob0)
    lib.call("mstts_decoder_train_fwd_persistent", C.byref(w.dec), C.byref(w.pdesc))        # launch B, directly behind
    torch.cuda.synchronize()
    st = w.pctrl.cpu().numpy()
    assert st[1] == 0 and st[2] == 256, st[:4]
    near_groups = int(st[3])
    eng.unpack_history(w)
    cut = lambda k, x: x[:-1] if k in ("c0", "c1") else x
    a = {k: cut(k, t2n(getattr(w, k))).copy() for k in HIST}
    eng._ensure_fallback_packs()                          # (the launch-per-step kernels' packed copies are made on demand)
    lib.call("mstts_decoder_train_fwd", C.byref(w.dec))
    torch.cuda.synchronize()
    bad = {k: rel_err(a[k], cut(k, t2n(getattr(w, k)))) for k in HIST}
    bad = {k: v for k, v in bad.items() if v > 1e-4}
    assert not bad, (bad, near_groups)
    if not getattr(w, "persist_bwd", False):
        return
    eng._forward_tail(w)
    eng.loss_and_backward(w)                              # BPTT launch A
    w.d_pj.mul_(0.5)
    lib.call("mstts_decoder_train_bwd_persistent", C.byref(w.dec_b), C.byref(w.pdesc_b))     # launch B
    torch.cuda.synchronize()
    st = w.pctrl_b.cpu().numpy()
    assert st[1] == 0 and st[2] == 256, st[:4]
    a = {k: t2n(getattr(w, k)).copy() for k in BWD}
    w.dq_hist.zero_()
    eng._ensure_fallback_packs()
    lib.call("mstts_decoder_train_bwd", C.byref(w.dec_b))
    torch.cuda.synchronize()
    bad = {k: rel_err(a[k], t2n(getattr(w, k))) for k in BWD}
    bad = {k: v for k, v in bad.items() if v > 5e-4}
    assert not bad, bad
    print("slice groups publishing through their XCD's L2: %d of 8" % near_groups)


# ---------------------------------------------------------------------------------------------------------------- config 3: bf16 recurrent products
def _bf16_engines(dev, seed=3):
    pd, od = dims_pair(**WIDE)
    values = OM.init_params(od, seed)
    g = np.random.default_rng(seed + 1)
    for k in values:
        if k.endswith(("bias", "bias_b")):
            values[k] = g.normal(0, 0.1, values[k].shape)
    eng = TrainEngine(pd, device=dev, values=values, recurrent_dtype="bf16", gemm_dtype="bf16")
    if not (eng.persist and eng.persist_bf16 and eng.persist_bwd):
        pytest.skip("persistent bf16 loops not available on this device")
    return eng, od


@pytest.mark.parametrize("B,Te,L,ragged", [(32, 128, 9, False), (8, 40, 12, True), (5, 160, 6, True), (1, 7, 2, False)])
def test_persistent_bf16_equals_launch_per_step_bf16(dev, B, Te, L, ragged):
    """BASELINE config 3 arithmetic inside the persistent launches (round 5: the BF16 instantiations of persist_fwd_kernel / persist_bwd_kernel,
    v_mfma_f32_16x16x32_bf16) against the launch-per-step bf16 loops (csrc/skinny_bf16.hip, cell_fwd_bf16_kernel) on the same engine, inputs and
    masks: both form bf(X) . bf(W) with fp32 accumulation from the same fp32 operands, in different summation orders.  Rounding to 8 mantissa bits
    is discontinuous - an operand that differs in the last fp32 bit between the two paths may round to a different bf16 value, 2^-9 of that
    operand - so the bound is that of a few bf16 flips, not fp32 rounding: every tensor of the loop (histories, linear / stop outputs) <= 4e-3 of
    its scale (typically 1-2e-3; the operands of the two runs are not bit-equal to begin with - the encoder's batch-norm statistics in front of
    the loop are fp32 atomic sums - and one run in eight of the (8, 40, 12) case came out at 2.2e-3 on the stop output: the persistent launch run
    TWICE is compared too and its own run-to-run difference is printed next to the cross-path one).  Behind the loop the postnet's five bf16 convolutions + batch norms amplify those flips (measured 2-4e-2 on mel_out, 1-5e-2 in
    relative L2 on the gradient slab - the level at which the emulating oracle's own gradients move under a 1e-6 perturbation,
    tests/test_cpu_oracle.py::test_bf16_emulation_sensitivity): bounded at 8e-2 / 1e-1.  Zero fallbacks, and the persistent launches really
    ran in bf16 mode."""
    eng, od = _bf16_engines(dev)
    batch = to_dev(OT.synthetic_batch(od, B, Te, L, seed=5, ragged=ragged), dev)
    seed = OT.step_seed(1234, 0)
    w = eng.plan(B, Te, L)
    assert w.persist and w.persist_bwd
    eng.forward(batch, w, seed=seed)
    eng.loss_and_backward(w)
    torch.cuda.synchronize()
    assert w.fold_prenet and w.pdesc.recurrent_bf16 == 1 and w.pdesc_b.recurrent_bf16 == 1
    assert eng.persist_fallbacks == 0 and eng.persist_bwd_fallbacks == 0
    a = _snapshot(w, eng)
    ga = t2n(eng.params.grad).astype(np.float64)
    eng.forward(batch, w, seed=seed)              # the same path once more: what two runs differ by on their own
    torch.cuda.synchronize()
    a2 = _snapshot(w, eng)
    self_noise = max(rel_err(a[k], a2[k]) for k in a if k != "mel_out")
    w.persist = w.persist_bwd = False
    eng.forward(batch, w, seed=seed)
    eng.loss_and_backward(w)
    torch.cuda.synchronize()
    b = _snapshot(w, eng)
    gb = t2n(eng.params.grad).astype(np.float64)
    errs = {k: rel_err(a[k], b[k]) for k in a}
    gl2 = float(np.sqrt(((ga - gb) ** 2).sum() / (gb ** 2).sum()))
    loop = {k: e for k, e in errs.items() if k != "mel_out"}
    print("persistent bf16 vs launch-per-step bf16: worst loop tensor %s (the persistent path against itself: %.2e), mel_out %.2e, gradient slab relative L2 %.2e"
          % (max(loop.items(), key=lambda kv: kv[1]), self_noise, errs["mel_out"], gl2))
    assert all(np.isfinite(v).all() for v in a.values())
    bad = {k: e for k, e in loop.items() if e > 4e-3}
    assert not bad, bad
    assert errs["mel_out"] < 8e-2 and gl2 < 1e-1, (errs["mel_out"], gl2)
    # ... and the mode really is bf16: against the fp32 persistent loops the same tensors are off by more than fp32 rounding
    e32 = TrainEngine(eng.d, device=dev, values=eng.params.export(), gemm_dtype="bf16")
    w32 = e32.plan(B, Te, L)
    e32.forward(batch, w32, seed=seed)
    torch.cuda.synchronize()
    assert rel_err(_snapshot(w32, e32)["pj"], a["pj"]) > 1e-4
