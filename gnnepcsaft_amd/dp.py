"""Data parallelism for the hot path: one process per GPU, graphs sharded by rank, ONE exchange step per training
step — the gradient all-reduce (average), as Lightning DDP does for the reference
(``/root/reference/gnnepcsaft/train/train.py:85-88``: ``devices="auto", strategy="auto"``, no SyncBatchNorm, so BN
statistics and the loss mean stay per-rank).

All parameter gradients live in one flat fp32 buffer (``p.grad`` are views into it), so the exchange is a single
``all_reduce`` on 8.8 MB (cfg-2/4) .. 40 MB (cfg-5): over xGMI (7 links x ~153 GB/s per GPU) that is latency-
not bandwidth-bound, and one call beats bucketed calls.  ``backend="nccl"`` is RCCL on ROCm; ``gloo`` is used by the
CPU tests of this logic.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist

from . import ops


class FlatGradAllReduce:
    """Owns a flat gradient buffer for ``module``'s parameters and averages it across ranks after backward."""

    def __init__(self, module: torch.nn.Module, process_group: Optional["dist.ProcessGroup"] = None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev, dtype = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=dtype, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    def zero_grad(self) -> None:
        """Zero in place (keeps ``p.grad`` as views; autograd then accumulates into the flat buffer)."""
        if self.flat.is_cuda:
            ops.zero_(self.flat)  # a fill launch: zero_() is a hipMemsetAsync, ~50 us of host time
        else:
            self.flat.zero_()

    def all_reduce(self, async_op: bool = False):
        """Sum over ranks then scale by 1/world (DDP's gradient averaging).  No-op for world size 1."""
        if self.world == 1:
            return None
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if async_op:
            return work
        self.flat.mul_(1.0 / self.world)
        return None

    def finish(self, work) -> None:
        if work is not None:
            work.wait()
            self.flat.mul_(1.0 / self.world)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, process_group=None) -> None:
    """Make every rank start from rank ``src``'s weights and buffers (DDP's initial broadcast)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return
    for t in list(module.parameters()) + [b for b in module.buffers() if b.dtype.is_floating_point]:
        dist.broadcast(t.data, src=src, group=process_group)


def reduce_logged(metrics: Dict[str, torch.Tensor], process_group=None) -> Dict[str, float]:
    """``sync_dist=True`` of ``self.log`` (reference models.py:99,106): mean of scalar metrics across ranks."""
    if not metrics:
        return {}
    keys = sorted(metrics)
    vals = torch.stack([torch.as_tensor(metrics[k], dtype=torch.float32).detach().reshape(()).to(
        metrics[keys[0]].device if isinstance(metrics[keys[0]], torch.Tensor) else "cpu") for k in keys])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.all_reduce(vals, op=dist.ReduceOp.SUM, group=process_group)
        vals = vals / dist.get_world_size(process_group)
    return {k: float(v) for k, v in zip(keys, vals.cpu())}
