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


class GradAllReduce:
    def __init__(self, grad_slab: torch.Tensor, world: int, bucket_mb: float = 32.0, group=None):
        self.world, self.group = world, group
        n = grad_slab.numel()
        per = max(1, int(bucket_mb * (1 << 20) / 4))
        self.bounds = [(s, min(n, s + per)) for s in range(0, n, per)]

        self.n = n
        self._works, self._done = [], []

    def __call__(self, grad_slab: torch.Tensor):
        """Whole slab at once (after backward)."""
        self.start(grad_slab, 0, self.n)
        self.finish(grad_slab)

    def start(self, grad_slab: torch.Tensor, lo: int, hi: int):
        """Asynchronously sum grad_slab[lo:hi] over the ranks (in <= bucket-size pieces).  Call it at the point of the
        backward pass where that range is final: the collective is ordered after everything enqueued so far on the
        current stream and runs beside what is enqueued next."""
        if self.world <= 1 or hi <= lo:
            return
        import torch.distributed as dist
        per = self.bounds[0][1] - self.bounds[0][0]
        for a in range(lo, hi, per):
            b = min(hi, a + per)
            self._works.append(dist.all_reduce(grad_slab[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self._done.append((a, b))

    def finish(self, grad_slab: torch.Tensor):
        """Wait for every started piece; ranges that were never started are reduced now (so the slab is always complete)."""
        if self.world <= 1:
            return
        covered, pos = sorted(self._done), 0
        for a, b in covered + [(self.n, self.n)]:
            if a > pos:
                self.start(grad_slab, pos, a)
            pos = max(pos, b)
        for w in self._works:
            w.wait()
        self._works, self._done = [], []
