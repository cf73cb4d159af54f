"""Data-parallel exchange step: one process per GPU, the flat fp32 gradient slab is summed across
ranks with RCCL (torch.distributed backend "nccl" on ROCm) over xGMI.  The reference has no
multi-GPU code (single tf.Session, MSTTS_SV.py:24); this is the only collective on the path.

The slab is reduced in a few large buckets (xGMI is point-to-point: fewer, larger messages keep
every link busy; 121 MB fp32 in total) issued back-to-back as async collectives and waited once.
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

    def __call__(self, grad_slab: torch.Tensor):
        if self.world <= 1:
            return
        import torch.distributed as dist
        works = [dist.all_reduce(grad_slab[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for a, b in self.bounds]
        for w in works:
            w.wait()
