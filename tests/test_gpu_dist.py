"""The data-parallel exchange step on real RCCL: a one-rank `nccl` group on the GPU box drives GradAllReduce / broadcast /
statistics averaging through actual collectives inside a train step (the 8-GPU run itself belongs to the driver)."""
import os
import socket

import numpy as np
import pytest
import torch

from multi_speaker_tts_amd.engine import TrainEngine
from oracle import train as OT
from tests.helpers import dims_pair, rel_err, t2n, to_dev

pytestmark = pytest.mark.gpu


def test_train_step_through_rccl_one_rank(dev):
    import torch.distributed as dist
    from multi_speaker_tts_amd import dist as D
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    try:
        pd, od = dims_pair(dec_lstm=64, enc_lstm=32, spk=64, prenet=32)
        batch = to_dev(OT.synthetic_batch(od, 4, 9, 6, seed=3, ragged=True), dev)
        plain = TrainEngine(pd, device=dev, seed=5)
        dp = TrainEngine(pd, device=dev, seed=5, rank=0, world=1)
        dp.broadcast_state(src=0)                                         # broadcast of the slabs + step through RCCL
        red = D.GradAllReduce(dp.params.grad, 1, bucket_mb=0.25, force=True)   # several buckets per announced range
        assert red.active and len(red.bounds) > 1
        n_coll = {"n": 0}
        orig = dist.all_reduce

        def counting(*a, **k):
            n_coll["n"] += 1
            return orig(*a, **k)
        dist.all_reduce = counting
        try:
            for _ in range(3):
                wa = plain.train_step(batch)
                wb = dp.train_step(batch, all_reduce=red)
        finally:
            dist.all_reduce = orig
        torch.cuda.synchronize()
        assert n_coll["n"] >= 3 * 3                                       # >= one collective per announced range per step
        # the sum over one rank is the identity: both engines walked the same trajectory (up to split-K atomics order)
        a, b = plain.params.train, dp.params.train
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
        sa, sb = plain.scalars(wa), dp.scalars(wb, average=True)          # averaged scalars: an all-reduce of four floats
        assert abs(sa["Loss"] - sb["Loss"]) < 1e-4 * abs(sa["Loss"])
        # config-3 exchange on real RCCL (all-to-all + all-gather of bf16 on the exchange's own stream, HIP pack / sum / unpack
        # kernels): over one rank the result is the gradient rounded to bf16, and the step after it must stay finite
        g = dp.params.grad
        dp.forward(batch, wb)
        dp.loss_and_backward(wb)
        ref16 = g.clone().to(torch.bfloat16).float()
        red16 = D.GradAllReduce(g, 1, bucket_mb=0.25, force=True, comm_dtype="bf16")
        red16.start(g, g.numel() // 2, g.numel())
        red16.start(g, 0, g.numel() // 3)
        red16.finish(g)                                                   # picks up the middle third itself
        torch.cuda.synchronize()
        assert torch.equal(g, ref16)
        wc = dp.train_step(batch, all_reduce=red16)
        torch.cuda.synchronize()
        assert np.isfinite(dp.scalars(wc)["Loss"]) and bool(torch.isfinite(dp.params.train).all())
        before = dp.params.frozen.clone()
        dp.sync_statistics()                                              # BN moving statistics averaged over (one) rank
        assert len(dp.moving_stat_ranges()) >= 1 and torch.allclose(before, dp.params.frozen)
    finally:
        dist.destroy_process_group()


def _two_rank_worker(rank, world, port, cfg, B, Te, L, q, backend="gloo"):
    """One data-parallel rank of a 2-rank job; both ranks share the one GPU of the box, the collectives go through gloo (which
    stages device tensors through the host) - everything else is the product path: sample-keyed masks, broadcast of rank 0's
    state, gradient all-reduce overlapped with the backward pass, 1/world folded into Adam, statistics averaging."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world))
    import torch.distributed as dist
    from multi_speaker_tts_amd import dist as D
    from multi_speaker_tts_amd.engine import TrainEngine as TE
    from tests.helpers import dims_pair as dp_, to_dev as td_
    from oracle import model as OM_, train as OT_
    dev = torch.device("cuda:%d" % (rank if backend == "nccl" else 0))     # RCCL: one GPU per rank; gloo: both ranks share the one GPU
    if backend == "nccl":
        os.environ["LOCAL_RANK"] = str(rank)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pd, od = dp_(**cfg)
        values = OM_.init_params(od, 21 + rank)                      # ranks start DIFFERENT: the broadcast must fix that
        eng = TE(pd, device=dev, values=values, seed=1234, rank=rank, world=world)
        eng.broadcast_state(src=0)
        full = OT_.synthetic_batch(od, world * B, Te, L, seed=9, ragged=True)
        mine = {k: v[rank * B:(rank + 1) * B] for k, v in full.items()}
        # a rank pads to ITS longest sequence; give every rank the global maximum so the shards line up with the oracle's
        red = D.GradAllReduce(eng.params.grad, world, bucket_mb=0.25)
        w = eng.train_step(td_(mine, dev), all_reduce=red)
        eng.sync_statistics()
        torch.cuda.synchronize()
        sc = eng.scalars(w, average=True)
        extra = {}
        if backend == "nccl":
            # the config-3 exchange with TWO ranks on RCCL: bf16 message, fp32 sum in rank order, one rounding - checked against what the
            # ranks' own gradients (collected with a plain fp32 all-gather) say the result must be, bit for bit
            g = eng.params.grad
            eng.forward(td_(mine, dev), w)
            eng.loss_and_backward(w)
            mine_g = g.clone()
            parts = [torch.empty_like(mine_g) for _ in range(world)]
            dist.all_gather(parts, mine_g)
            want = sum(p.to(torch.bfloat16).float() for p in parts).to(torch.bfloat16).float()
            red16 = D.GradAllReduce(g, world, bucket_mb=0.25, comm_dtype="bf16")
            red16.start(g, g.numel() // 2, g.numel())
            red16.finish(g)
            torch.cuda.synchronize()
            extra["bf16_exchange_exact"] = bool(torch.equal(g, want))
            extra["exposed_ms"] = red.exposed_ms()
        q.put((rank, eng.params.export(), sc["Loss"], eng.exchange_timeouts(w), extra))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_rccl_against_oracle(dev):
    """The same two-rank step on TWO GPUs over RCCL (one process per GPU, backend nccl) - self-skipping on a one-GPU box, so that a
    multi-GPU box exercises the real collectives: fp32 all-reduce overlapped with the backward pass against the oracle, then the
    bf16 exchange (all-to-all + fp32 sum + all-gather) with nr = 2, bit for bit."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL between two ranks)")
    _run_two_ranks("nccl")


def test_two_ranks_share_one_gpu_against_oracle(dev):
    """N = 2 data parallel end to end on the hardware at hand: two processes on the one GPU, gloo collectives.  Expected result
    from the oracle: every rank's gradient on its shard (masks keyed by global sample index, per-rank BN batch statistics), the
    mean of the two, ONE TF-Adam update from rank 0's initial state; BN moving statistics = mean of the ranks' updates."""
    _run_two_ranks("gloo")


def _run_two_ranks(backend):
    import torch.multiprocessing as mp
    from oracle import model as OM
    cfg = dict(dec_lstm=64, enc_lstm=32, spk=64, prenet=32)
    B, Te, L, world = 3, 11, 7, 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, cfg, B, Te, L, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, params, loss, timeouts, extra = q.get(timeout=600)
        got[r] = (params, loss, timeouts)
        if backend == "nccl":
            assert extra["bf16_exchange_exact"], r
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # ---- the same step in the oracle
    pd, od = dims_pair(**cfg)
    values = OM.init_params(od, 21)                                   # rank 0's state
    full = OT.synthetic_batch(od, world * B, Te, L, seed=9, ragged=True)
    grads, stats, losses = [], [], []
    for r in range(world):
        mine = {k: v[r * B:(r + 1) * B] for k, v in full.items()}
        S = int(mine["Mel"].shape[1]) + 1
        masks = OT.make_masks(od, B, Te, S, True, seed=OT.step_seed(1234, 0), rank=r)
        new_p, _, sc, g, _ = OT.train_step(values, None, od, mine, masks, 0, return_grads=True)
        grads.append(g); stats.append(new_p); losses.append(sc["Loss"])
    lr = OT.learning_rate(0)
    for r in range(world):
        params, loss, timeouts = got[r]
        assert timeouts == 0
        assert abs(loss - float(np.mean(losses))) < 2e-4 * max(1.0, abs(float(np.mean(losses))))
        bad = {}
        for k in values:
            if k.startswith("speaker_embedding"):
                continue
            p0 = torch.tensor(np.asarray(values[k]), dtype=torch.float64)
            if k in grads[0]:
                gm = sum(g[k] for g in grads) / world
                want, _, _ = OT.adam_tf(p0, gm, torch.zeros_like(p0), torch.zeros_like(p0), 1, lr)
            elif OM.is_trainable(k):
                want = p0
            else:
                want = sum(s[k].double() if torch.is_tensor(s[k]) else torch.tensor(np.asarray(s[k])) for s in stats) / world
            e = rel_err(params[k], want.numpy())
            if e > 2e-3:
                bad[k] = e
        assert not bad, (r, bad)
    # both ranks end with the same variables, bit for bit
    for k in got[0][0]:
        assert np.array_equal(got[0][0][k], got[1][0][k]), k


# ---------------------------------------------------------------------------------------------------------------- one rank's launch gives up
REFW = dict(emb=64, enc_conv_ch=64, enc_lstm=256, spk=256, prenet=256, dec_lstm=1024, n_mel=16, post_ch=32)


def _abort_worker(rank, world, port, B, Te, L, q, ev, fail):
    """One rank of a 2-rank job at the reference's recurrent widths (persistent launches on).  Both ranks share the one GPU, and a persistent
    launch wants all of its CUs, so the ranks take turns on the device (events): forward 0, forward 1, backward 0 up to its `agree`, backward 1.
    `fail`: rank 1's persistent BPTT (or encoder BPTT) launch gives up through the self-test knob - rank 0's own launches are healthy."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world))
    import torch.distributed as dist
    from multi_speaker_tts_amd import dist as D
    from multi_speaker_tts_amd.engine import TrainEngine as TE
    from tests.helpers import dims_pair as dp_, to_dev as td_
    from oracle import model as OM_, train as OT_
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pd, od = dp_(**REFW)
        eng = TE(pd, device=dev, values=OM_.init_params(od, 21), seed=1234, rank=rank, world=world)
        full = OT_.synthetic_batch(od, world * B, Te, L, seed=9, ragged=True)
        mine = td_({k: v[rank * B:(rank + 1) * B] for k, v in full.items()}, dev)
        red = D.GradAllReduce(eng.params.grad, world, bucket_mb=4.0)
        fwd0, fwd1, agree0 = ev
        orig_fwd, orig_bwd, orig_agree = eng.forward, eng.loss_and_backward, red.agree

        def fwd(*a, **k):
            if rank == 1:
                fwd0.wait(300)
            out = orig_fwd(*a, **k)
            torch.cuda.synchronize()
            (fwd0 if rank == 0 else fwd1).set()
            return out

        def bwd(*a, **k):
            (fwd1 if rank == 0 else agree0).wait(300)
            if rank == 1 and fail == "encoder" and not k.get("_redo"):
                eng.persist_enc_selftest = 1                 # (set here: the forward pass's own encoder check would consume it)
            return orig_bwd(*a, **k)

        def agree(ok):
            if rank == 0:
                torch.cuda.synchronize()
                agree0.set()
            return orig_agree(ok)
        eng.forward, eng.loss_and_backward, red.agree = fwd, bwd, agree
        if rank == 1 and fail == "bptt":
            eng.persist_bwd_selftest = 3
        w = eng.train_step(mine, all_reduce=red)
        torch.cuda.synchronize()
        info = dict(persist=bool(w.persist and w.persist_bwd and w.persist_enc), fwd_fallbacks=eng.persist_fallbacks, bwd_fallbacks=eng.persist_bwd_fallbacks,
                    enc_fallbacks=eng.persist_enc_fallbacks, collective_redos=eng.collective_redos)
        q.put((rank, eng.params.train.cpu().numpy(), eng.params.grad.cpu().numpy(), eng.scalars(w, average=True)["Loss"], info))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_launch_gives_up(dev):
    """engine.loss_and_backward's re-run decision is collective (GradAllReduce.agree): when ONE rank's persistent BPTT launch gives up after
    the gradient all-reduces of the pass have started, BOTH ranks drain them and run the pass again - the ranks issue the same sequence of
    collectives (no hang, no pairing with the peer's next step), nobody applies the junk partial gradients, and the job ends where the
    undisturbed job ends: every variable after Adam equal on both ranks bit for bit, and equal to the healthy run's to fp32 rounding."""
    import torch.multiprocessing as mp
    eng_probe = TrainEngine(dims_pair(**REFW)[0], device=dev)
    if not (eng_probe.persist and eng_probe.persist_bwd and eng_probe.persist_enc):
        pytest.skip("persistent launches not available on this device")
    del eng_probe
    torch.cuda.empty_cache()
    B, Te, L, world = 3, 11, 7, 2
    results = {}
    for mode in ("none", "bptt", "encoder"):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        ev = tuple(ctx.Event() for _ in range(3))
        procs = [ctx.Process(target=_abort_worker, args=(r, world, port, B, Te, L, q, ev, mode)) for r in range(world)]
        for p in procs:
            p.start()
        got = {}
        for _ in range(world):
            r, params, grad, loss, info = q.get(timeout=600)
            got[r] = (params, loss, info, grad)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        results[mode] = got
        assert np.array_equal(got[0][0], got[1][0]), mode                 # both ranks end with the same variables, bit for bit
        assert got[0][2]["persist"] and got[1][2]["persist"]
    assert results["none"][0][2]["collective_redos"] == 0 and results["none"][1][2]["collective_redos"] == 0
    assert all(results["none"][r][2][k] == 0 for r in (0, 1) for k in ("fwd_fallbacks", "bwd_fallbacks", "enc_fallbacks"))
    for fail in ("bptt", "encoder"):
        i0, i1 = results[fail][0][2], results[fail][1][2]
        assert (i1["bwd_fallbacks"] if fail == "bptt" else i1["enc_fallbacks"]) == 1 and i1["collective_redos"] == 0, (fail, i1)
        assert i0["bwd_fallbacks"] == 0 and i0["enc_fallbacks"] == 0 and i0["collective_redos"] == 1, (fail, i0)      # rank 0 re-ran because its PEER asked
        a, b = results["none"][0][3], results[fail][0][3]                  # the summed gradient both ranks applied
        assert float(np.abs(a - b).max()) <= 2e-4 * float(np.abs(a).max()), (fail, float(np.abs(a - b).max()), float(np.abs(a).max()))
        assert abs(results["none"][0][1] - results[fail][0][1]) < 1e-4 * max(1.0, abs(results["none"][0][1]))
