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


def _worker_helpers(rank, world, port, q, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    from multi_speaker_tts_amd import dist as D
    assert D.env_ranks() == (rank, rank, world)
    assert D.init_process_group(backend="gloo") == (rank, rank, world) and dist.is_initialized()
    # initial state: every rank ends with rank 0's copy
    a, b = torch.full((7,), float(rank + 1)), torch.arange(5, dtype=torch.float32) + 10 * rank
    D.broadcast_([a, b], src=0)
    assert torch.equal(a, torch.ones(7)) and torch.equal(b, torch.arange(5, dtype=torch.float32))
    # BN moving statistics / loss scalars: mean over the ranks
    st = torch.tensor([1.0, 2.0, 3.0]) * (rank + 1)
    D.average_([st])
    assert torch.allclose(st, torch.tensor([1.5, 3.0, 4.5]))
    # one-rank-style forced path is the same code as the multi-rank one
    g = torch.ones(10) * (rank + 1)
    red = D.GradAllReduce(g, world, force=True)
    red(g)
    assert torch.equal(g, torch.full((10,), 3.0))
    # config-3 exchange: bf16 message, fp32 accumulation in rank order, one rounding of the sum; pieces that do not divide by the
    # rank count, started out of order like the train step does
    base = torch.linspace(-3.0, 3.0, 1001) * 1.2345
    mine = base * (rank + 1) + 0.001 * rank
    want = ((base * 1 + 0.0).to(torch.bfloat16).float() + (base * 2 + 0.001).to(torch.bfloat16).float()).to(torch.bfloat16).float()
    h = mine.clone()
    red16 = D.GradAllReduce(h, world, bucket_mb=0.001, comm_dtype="bf16")      # 262-element pieces
    assert red16.bf16 and len(red16.bounds) > 2
    red16.start(h, 500, 1001)
    red16.start(h, 100, 500)
    red16.finish(h)
    assert torch.equal(h, want), (h - want).abs().max()
    # feeder sharding: the ranks walk the same shuffled batch list of an epoch and take disjoint, interleaved parts of it
    import pickle
    import random
    from multi_speaker_tts_amd import Feeder as F
    from multi_speaker_tts_amd import Hyper_Parameters as hp
    hp.Train.Pattern_Path, hp.Train.Batch_Size, hp.Train.Max_Pattern_Queue = tmp, 2, 100
    f = F.Feeder(is_Training=True, device="cpu", rank=rank, world=world)
    want = F.epoch_batches(F.train_file_order(f.metadata_Dict), random.Random(1234))[rank::world]
    got = [f.Get_Train_Pattern() for _ in range(len(want))]
    f.close()
    toks = [[int(t[1]) for t in p["Token"]] for p in got]              # first real token of every row identifies the file
    q.put((rank, toks, [[int(n.split("_")[1].split(".")[0]) for n in names] for names in want]))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_helpers_and_feeder_sharding_two_ranks(tmp_path):
    """broadcast / average / forced all-reduce helpers over gloo, and the data-parallel feeder contract."""
    import pickle
    from multi_speaker_tts_amd import Hyper_Parameters as hp
    from multi_speaker_tts_amd import Pattern_Generate as PG
    files = []
    for i in range(12):                                              # 12 tiny patterns, token i + 2 marks file i
        name = "LJ.P_%d.PICKLE" % i
        with open(tmp_path / name, "wb") as f:
            pickle.dump({"Token": np.array([i + 2, 5], np.int32), "Mel": np.zeros((60 + i, hp.Sound.Mel_Dim), np.float32), "Text": "x", "Dataset": "VCTK"}, f, protocol=2)
        files.append(name)
    PG.Metadata_Generate(pattern_path=str(tmp_path))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_helpers, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, toks, want = q.get(timeout=180)
        res[r] = (toks, want)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = []
    for r in range(2):
        toks, want = res[r]
        assert [[t - 2 for t in row] for row in toks] == want            # each rank got exactly its interleaved share, in order
        seen += [i for row in want for i in row]
    assert sorted(seen) == list(range(12))                             # disjoint and complete


def _worker_agree(rank, world, port, q):
    """The protocol of engine.loss_and_backward's last lines on two gloo ranks: ranges announced while the pass runs (async all-reduces in
    flight), then the verdict; rank 1's pass 'gave up' in step 1, so BOTH ranks drain and run the pass again."""
    import types
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multi_speaker_tts_amd.dist import GradAllReduce
    from multi_speaker_tts_amd.engine import TrainEngine
    eng = types.SimpleNamespace(collective_redos=0)
    verdict = types.MethodType(TrainEngine._pass_verdict, eng)
    n = 3000
    g = torch.zeros(n)
    red = GradAllReduce(g, world, bucket_mb=0.002)
    assert red.agree(True) is True and red.agree(rank == 0) is False and red.agree(False) is False
    log = []

    def backward(step, redo=False):
        junk = (step == 1 and rank == 1 and not redo)           # this rank's persistent launch left junk gradients behind
        g.zero_()
        g.add_(float("nan") if junk else float(rank + 1) * (step + 1))
        for lo, hi in ((2000, 3000), (500, 2000), (0, 500)):     # postnet -> decoder/attention -> encoder
            red.start(g, lo, hi)
            log.append(("start", step, redo))
        if not verdict(not junk, red.agree, redo):
            red.finish(g)                                        # on_abort
            log.append(("drain", step))
            return backward(step, redo=True)

    out = []
    for step in range(3):
        backward(step)
        red.finish(g)
        out.append(g.clone())
    q.put((rank, [o.numpy().copy() for o in out], eng.collective_redos, log))
    dist.barrier()
    dist.destroy_process_group()


def test_the_redo_decision_is_collective():
    """one rank's launch gives up after the pass's collectives have started.  Every rank must take the same decision, or
    the ranks' collective sequences fall out of step.  Here: three steps, the middle one re-run by both ranks; every step ends with the
    plain sum on both ranks (no NaN of the abandoned pass survives), rank 0 counts one peer-induced redo, rank 1 none."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_agree, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, out, redos, log = q.get(timeout=120)
        res[r] = (out, redos, log)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        for step, o in enumerate(res[r][0]):
            assert np.array_equal(o, np.full(3000, 3.0 * (step + 1), np.float32)), (r, step, o[:4])
    assert res[0][1] == 1 and res[1][1] == 0
    assert res[0][2] == res[1][2] and res[0][2].count(("drain", 1)) == 1 and len([e for e in res[0][2] if e[0] == "start"]) == 12


_DYING_RANK = r"""
import os, sys, time
import torch.distributed as dist
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
dist.barrier()
if rank == 1:
    os._exit(7)                    # dies between two collectives, without telling anybody
import torch
t = torch.ones(4)
dist.all_reduce(t)                 # rank 0 now waits for a peer that will never come (gloo: 30 minutes)
time.sleep(600)
"""


def test_launcher_stops_the_job_when_a_rank_dies():
    """bench.py's own launcher used to wait() for the ranks in order - one rank dying in init_process_group or falling
    out of a collective left the others blocked until the driver's time-out, and the one scaling run produced nothing.  Now all ranks are
    polled; the first non-zero exit stops the rest and becomes the launcher's exit code, within seconds."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    t0 = time.monotonic()
    try:
        bench.self_launch(2, cmd=[sys.executable, "-c", _DYING_RANK], grace_s=3.0)
        code = 0
    except SystemExit as e:
        code = e.code
    took = time.monotonic() - t0
    assert code == 7, code
    assert took < 15.0, took                      # (import torch + gloo rendezvous of the two children are ~5 s of it; the stop itself is < 4 s)
    # ... and a healthy job returns 0
    ok = "import os, sys; sys.exit(0)"
    assert bench.self_launch(2, cmd=[sys.executable, "-c", ok]) == 0
