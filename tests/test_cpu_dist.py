"""world_size-2 gloo test of the data-parallel exchange step (the only collective on the path)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multi_speaker_tts_amd.dist import GradAllReduce
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    red = GradAllReduce(g, world, bucket_mb=0.001)          # forces several buckets
    assert len(red.bounds) > 1
    red(g)
    # ranged form used by the train step: ranges started out of order while "backward" goes on, the rest picked up by finish()
    h = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    red.start(h, 700, 1000)
    red.start(h, 200, 700)
    red.finish(h)
    assert torch.equal(g, h)
    q.put((rank, g.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_all_reduce_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.arange(1000, dtype=np.float32) * 3
    assert np.array_equal(res[0], want) and np.array_equal(res[1], want)
