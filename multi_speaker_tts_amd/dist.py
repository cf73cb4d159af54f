"""Data-parallel exchange step: one process per GPU, the flat fp32 gradient slab is summed across
ranks with RCCL (torch.distributed backend "nccl" on ROCm) over xGMI.  The reference has no
multi-GPU code (single tf.Session, MSTTS_SV.py:24); this is the only collective on the path.

The slab is reduced in a few large buckets (xGMI is point-to-point: fewer, larger messages keep
every link busy; 121 MB fp32 in total), started as async collectives at the three points of the backward pass where a
module's gradients become final (postnet -> decoder/attention -> encoder, the reverse of the forward order) so that they
run under the remaining backward work, and waited once before Adam.
The 1/world mean is folded into the Adam kernel's grad_scale, so no extra pass touches the slab.
"""
from __future__ import annotations

import torch


def env_ranks():
    """(rank, local_rank, world) of this process as torch.distributed.run / bench.py's own launcher export them."""
    import os
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_process_group(backend=None, device=None):
    """One process per GPU: join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (RCCL when a device is given,
    gloo otherwise).  No-op when the group exists already or the job has a single process and no MASTER_PORT."""
    import os
    import torch.distributed as dist
    rank, local_rank, world = env_ranks()
    if dist.is_initialized():
        return rank, local_rank, world
    if world == 1 and "MASTER_PORT" not in os.environ:
        return rank, local_rank, world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = backend or ("nccl" if device is not None and torch.device(device).type == "cuda" else "gloo")
    kw = {"device_id": torch.device(device)} if backend == "nccl" and device is not None else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def broadcast_(tensors, src=0, group=None):
    """Make every rank start from rank `src`'s copy (initial variables, Adam slots, BN statistics)."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return
    for t in tensors:
        dist.broadcast(t, src=src, group=group)


def average_(tensors, group=None):
    """In-place mean over the ranks (BN moving statistics, the loss scalars: SURVEY 8e)."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t.mul_(1.0 / world)


class GradAllReduce:
    def __init__(self, grad_slab: torch.Tensor, world: int, bucket_mb: float = 32.0, group=None, force: bool = False,
                 comm_dtype: str = "f32", overlap: bool = True, trace: bool = False):
        """force: issue the collectives even in a one-rank group (exercises the RCCL path on a single GPU).
        overlap=False: start() only notes the range; the whole slab is exchanged in finish(), behind the backward pass (the A/B
        switch for the overlapped form, `bench.py --no-overlap`).  exposed_ms() reports how long the compute stream waited in finish().
        comm_dtype "bf16" (BASELINE config 3): the message is bf16, the accumulation fp32 - each piece is rounded to bf16, every
        rank receives its 1/world shard of every rank's piece (all-to-all), sums the shards in fp32 in rank order, rounds the sum
        once and all-gathers it: half the bytes of the fp32 all-reduce on every xGMI link, and no bf16 partial sums anywhere.
        The exchange runs on its own stream (ordered after the producing kernels, joined in finish()).
        trace: bench.py - per step, a timing event where the FIRST range is announced (compute stream) and one when its first piece has
        been exchanged (a probe stream that waits for that piece only): together with the engine's event behind the persistent BPTT launch
        they show whether the first collective ran under the backward pass or queued behind it (first_piece_trace())."""
        if comm_dtype not in ("f32", "bf16"):
            raise ValueError("comm_dtype must be 'f32' or 'bf16'")
        self.world, self.group = world, group
        self.overlap = bool(overlap)
        self._exposed = []                   # (event before the waits, event after them) per finish() on a GPU
        self.active = world > 1 or force
        self.bf16 = comm_dtype == "bf16"
        n = grad_slab.numel()
        per = max(1, int(bucket_mb * (1 << 20) / 4))
        self.bounds = [(s, min(n, s + per)) for s in range(0, n, per)]

        self.n = n
        self._device = grad_slab.device
        self._works, self._done = [], []
        self._trace_on = bool(trace) and grad_slab.is_cuda
        self._trace, self._probe = None, None
        self._side = None
        # The keep-or-re-run decision of a backward pass (agree / agree_async) travels on its OWN communicator: on the gradient buckets' one it
        # would queue behind every bucket of the pass (a communicator runs its collectives in issue order) and could not be read before the
        # whole exchange has ended.  new_group is itself collective: every rank constructs its GradAllReduce at the same point.
        self._flag_group, self.async_flag = None, False
        if self.active:
            import os
            import torch.distributed as dist
            if dist.is_initialized() and dist.get_backend(group) == "nccl" and grad_slab.is_cuda and os.environ.get("MSTTS_ASYNC_AGREE", "1") != "0":
                self._flag_group = dist.new_group(ranks=dist.get_process_group_ranks(group) if group is not None else None)
                self.async_flag = True
                # bring the communicator up NOW, while nothing else is in flight: a communicator created lazily by its first collective -
                # at the end of the first backward pass, with the gradient buckets of the other communicator running - allocates device
                # memory and synchronises the device in the middle of them, on every rank at a slightly different moment
                warm = torch.ones(1, dtype=torch.int32, device=grad_slab.device)
                dist.all_reduce(warm, op=dist.ReduceOp.MIN, group=self._flag_group)
                torch.cuda.synchronize(grad_slab.device)
                if int(warm.item()) != 1:
                    raise RuntimeError("GradAllReduce: the verdict communicator's warm-up all-reduce returned %d" % int(warm.item()))
        if self.bf16 and self.active:
            import torch.distributed as dist
            self._nr = dist.get_world_size(group)                      # ranks in the exchange (1 in the forced one-rank form)
            cap = (per + self._nr - 1) // self._nr * self._nr        # a piece padded to a multiple of the rank count
            z = lambda m: torch.zeros(m, dtype=torch.bfloat16, device=grad_slab.device)
            self._send, self._recv, self._shard, self._out = z(cap), z(cap), z(cap // self._nr), z(cap)
            if grad_slab.is_cuda:
                self._side = torch.cuda.Stream(device=grad_slab.device)

    def __call__(self, grad_slab: torch.Tensor):
        """Whole slab at once (after backward)."""
        self.start(grad_slab, 0, self.n)
        self.finish(grad_slab)

    # -- bf16 message, fp32 accumulate: one piece, on the current stream (the side stream on a GPU) --
    def _exchange_bf16(self, g: torch.Tensor):
        import torch.distributed as dist
        m, nr = g.numel(), self._nr
        chunk = (m + nr - 1) // nr
        send, recv, shard, out = self._send[:chunk * nr], self._recv[:chunk * nr], self._shard[:chunk], self._out[:chunk * nr]
        if g.is_cuda:
            from . import lib
            lib.call("mstts_f32_to_bf16", lib.ptr(g), lib.ptr(send), m)
            if chunk * nr > m:
                send[m:].zero_()
            dist.all_to_all_single(recv, send, group=self.group)                    # recv[r*chunk:(r+1)*chunk] = rank r's shard for me
            lib.call("mstts_bf16_chunks_sum", lib.ptr(recv), nr, chunk, chunk, lib.ptr(shard))
            dist.all_gather_into_tensor(out, shard, group=self.group)
            lib.call("mstts_bf16_to_f32", lib.ptr(out), lib.ptr(g), m)
        else:                               # CPU tensors (the gloo tests): the same arithmetic with torch ops
            send[:m] = g.to(torch.bfloat16)
            send[m:].zero_()
            dist.all_to_all_single(recv, send, group=self.group)
            acc = torch.zeros(chunk, dtype=torch.float32)
            for r in range(nr):
                acc += recv[r * chunk:(r + 1) * chunk].float()
            shard.copy_(acc.to(torch.bfloat16))
            parts = [torch.empty_like(shard) for _ in range(nr)]
            dist.all_gather(parts, shard, group=self.group)
            g.copy_(torch.cat(parts)[:m].float())

    def start(self, grad_slab: torch.Tensor, lo: int, hi: int):
        """Asynchronously sum grad_slab[lo:hi] over the ranks (in <= bucket-size pieces).  Call it at the point of the
        backward pass where that range is final: the collective is ordered after everything enqueued so far on the
        current stream and runs beside what is enqueued next."""
        if not self.active or hi <= lo or (not self.overlap and not getattr(self, "_finishing", False)):
            return
        import torch.distributed as dist
        per = self.bounds[0][1] - self.bounds[0][0]
        first = self._trace_on and not self._done and not self._works
        if first:
            self._trace = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), min(hi, lo + per) - lo]
            self._trace[0].record()
        if self.bf16:
            if self._side is not None:
                self._side.wait_stream(torch.cuda.current_stream(grad_slab.device))
                with torch.cuda.stream(self._side):
                    for a in range(lo, hi, per):
                        self._exchange_bf16(grad_slab[a:min(hi, a + per)])
                        if first and a == lo:
                            self._trace[1].record()
            else:
                for a in range(lo, hi, per):
                    self._exchange_bf16(grad_slab[a:min(hi, a + per)])
            self._done += [(a, min(hi, a + per)) for a in range(lo, hi, per)]
            return
        for a in range(lo, hi, per):
            b = min(hi, a + per)
            self._works.append(dist.all_reduce(grad_slab[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self._done.append((a, b))
            if first and a == lo:
                if self._probe is None:
                    self._probe = torch.cuda.Stream(device=grad_slab.device)
                with torch.cuda.stream(self._probe):
                    self._works[-1].wait()               # (the probe stream waits for this piece only)
                    self._trace[1].record()

    def finish(self, grad_slab: torch.Tensor):
        """Wait for every started piece; ranges that were never started are reduced now (so the slab is always complete)."""
        if not self.active:
            return
        covered, pos = sorted(self._done), 0
        self._finishing = True
        for a, b in covered + [(self.n, self.n)]:
            if a > pos:
                self.start(grad_slab, pos, a)
            pos = max(pos, b)
        self._finishing = False
        ev = None
        if grad_slab.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self._works:
            w.wait()
        if self._side is not None:
            torch.cuda.current_stream(grad_slab.device).wait_stream(self._side)
        if ev is not None:
            ev[1].record()
            self._exposed = (self._exposed + [ev])[-64:]
        self._works, self._done = [], []

    def agree(self, ok: bool) -> bool:
        """True only if `ok` is true on EVERY rank (one MIN all-reduce of a single word, ordered behind the gradient collectives started so
        far; identity in a one-process job).  engine.loss_and_backward asks this before it keeps or re-runs a backward pass whose
        collectives are already in flight: the decision has to be the same on every rank or the ranks' collective sequences diverge."""
        if not self.active or self.world <= 1:
            return bool(ok)
        import torch.distributed as dist
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self._device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()))

    def agree_async(self, flag: torch.Tensor):
        """The same decision without a host round trip: `flag` (int32[1] on the device, 1 = this rank's persistent launches of the pass ran to
        their end; written by work already enqueued on the CURRENT stream) becomes the MINIMUM over the ranks, in place, ordered on the
        current stream.  The caller runs this on a side stream right behind its launches' status words and reads the result together with
        them at the pass's one host sync - by then the hoisted products are still running, so nothing waits for the exchange.
        Only offered on RCCL (async_flag); every rank must call it once per pass, like agree()."""
        import torch.distributed as dist
        work = dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self._flag_group, async_op=True)
        work.wait()                          # (stream-level: the current stream waits for the collective, the host does not)

    def first_piece_trace(self, reference_event):
        """(ms from `reference_event` to the announcement of the step's first range, ms from `reference_event` to the end of that range's
        first piece, elements of the piece) of the last traced step, or None.  Synchronises."""
        if not self._trace:
            return None
        torch.cuda.synchronize()
        return reference_event.elapsed_time(self._trace[0]), reference_event.elapsed_time(self._trace[1]), self._trace[2]

    def exposed_ms(self):
        """Mean time per finish() that the compute stream spent waiting for the exchange (what the overlap did not hide).  Synchronises."""
        if not self._exposed:
            return 0.0
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self._exposed) / len(self._exposed)
