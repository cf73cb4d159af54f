"""The data-parallel exchange step on real RCCL: a one-rank `nccl` group on the GPU box drives GradAllReduce / broadcast /
statistics averaging through actual collectives inside a train step (the 8-GPU run itself belongs to the driver)."""
import os
import socket

import numpy as np
import pytest
import torch

from multi_speaker_tts_amd.engine import TrainEngine
from oracle import train as OT
from tests.helpers import dims_pair, t2n, to_dev

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
        assert len(dp.moving_stat_ranges()) >= 2 and torch.allclose(before, dp.params.frozen)
    finally:
        dist.destroy_process_group()
