"""Data parallelism for the hot path: one process per GPU, graphs sharded by rank, ONE exchange per training step —
the gradient all-reduce (average), as Lightning DDP does for the reference
(``/root/reference/gnnepcsaft/train/train.py:85-88``: ``devices="auto", strategy="auto"``, no SyncBatchNorm, so BN
statistics and the loss mean stay per-rank).

All parameter gradients live in one flat fp32 buffer (``p.grad`` are views into it): 8.8 MB (cfg-2/4) .. 40 MB (cfg-5).
The exchange is issued in slices that become final at different times of the backward pass:

  * one slice per conv layer, started the moment that layer's weight-gradient launches have been ISSUED (hook from
    ``ops.finish_backward``), on the stream those launches run on (the library's weight-gradient stream), so RCCL's
    own stream is ordered behind them by an event and the exchange of layer l overlaps the input-gradient chain of
    layers l-1 .. 0 (the reference gets the same from DDP's bucketed reducer);
  * the rest (embeddings, BatchNorm, readout) in one call after backward.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a slice of 1.4 MB (cfg-2's conv layer) is latency-bound, so
slices are never split further.  The sum is turned into DDP's average by one gnx_scale launch over the flat buffer in
``finish``.  ``backend="nccl"`` is RCCL on ROCm; ``gloo`` is used by the CPU tests of this logic.
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import ops


class FlatGradAllReduce:
    """Owns a flat gradient buffer for ``module``'s parameters and averages it across ranks after backward."""

    def __init__(self, module: torch.nn.Module, process_group: Optional["dist.ProcessGroup"] = None,
                 force_collective: bool = False):
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        self.params = [p for _, p in named]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev, dtype = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=dtype, device=dev)
        off = 0
        layer_range: Dict[int, List[int]] = {}
        for n, p in named:
            k = p.numel()
            p.grad = self.flat[off:off + k].view_as(p)
            m = re.search(r"(?:^|\.)convs\.(\d+)\.", n)
            if m:  # parameters of one conv layer are adjacent in module.parameters() order
                r = layer_range.setdefault(int(m.group(1)), [off, off + k])
                r[0], r[1] = min(r[0], off), max(r[1], off + k)
            off += k
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        # run the collective even for one rank (rehearses the RCCL path on a one-GPU box)
        self.collective = self.world > 1 or (force_collective and dist.is_available() and dist.is_initialized())
        self.layer_slices: List[Tuple[int, int]] = [tuple(layer_range[l]) for l in sorted(layer_range)]
        self._done: List[Tuple[int, int]] = []   # slices already handed to the collective this step
        self._works: list = []
        self._overlap = False
        self._side = None

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    def zero_grad(self) -> None:
        """Zero in place (keeps ``p.grad`` as views; autograd then accumulates into the flat buffer)."""
        if self._works:
            raise RuntimeError("zero_grad() while an exchange is in flight: call finish() first")
        self._done = []
        if self.flat.is_cuda:
            ops.zero_(self.flat)  # a fill launch: zero_() is a hipMemsetAsync, ~50 us of host time
        else:
            self.flat.zero_()

    # ---- overlapped exchange ------------------------------------------------------------------------------------
    def enable_overlap(self, enabled: bool = True) -> None:
        """Start every conv layer's slice of the exchange as soon as its weight-gradient launches are issued.  Needs
        ``functional.set_grad_in_place(True)`` (the kernels write the flat buffer directly).  Not under HIP-graph
        capture (a collective cannot be captured here): disable it there and call ``all_reduce`` after the replay."""
        self._overlap = bool(enabled) and self.collective
        ops.set_wgrad_done_hook(self._on_wgrads_issued if self._overlap else None)

    def _slice_of(self, t: torch.Tensor) -> Optional[Tuple[int, int]]:
        off = (t.data_ptr() - self.flat.data_ptr()) // self.flat.element_size()
        for lo, hi in self.layer_slices:
            if lo <= off < hi:
                return lo, hi
        return None

    def _on_wgrads_issued(self, sinks) -> None:
        first = next((s for s in sinks if s is not None), None)
        if first is None or first.device != self.flat.device:
            return
        sl = self._slice_of(first)
        if sl is None or sl in self._done:
            return  # a Linear of the readout / not one of ours: goes with the rest
        self.reduce_slice(*sl)

    def reduce_slice(self, lo: int, hi: int) -> None:
        """Asynchronous all-reduce (sum) of flat[lo:hi], ordered behind everything issued so far on the stream the
        weight gradients run on."""
        if not self.collective or (lo, hi) in self._done:
            return
        self._done.append((lo, hi))
        view = self.flat[lo:hi]
        if self.flat.is_cuda and ops.wgrad_stream_enabled():
            if self._side is None:
                self._side = ops.side_stream(self.flat.device)
            if ops.second_side_stream_in_use(self.flat.device):
                # part of a conv layer's gradients (bond-table chain) is produced on side stream 1
                self._side.wait_stream(ops.side_stream(self.flat.device, 1))
            with torch.cuda.stream(self._side):
                self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    # ---- the exchange -------------------------------------------------------------------------------------------
    def all_reduce(self, async_op: bool = False):
        """Sum over ranks of everything not exchanged yet, then (``finish``) scale by 1/world: DDP's gradient
        averaging.  With ``async_op`` the caller calls ``finish()`` later.  No-op for world size 1."""
        if not self.collective:
            return None
        lo = 0
        for a, b in sorted(self._done) + [(self.flat.numel(), self.flat.numel())]:
            if a > lo:
                self._works.append(dist.all_reduce(self.flat[lo:a], op=dist.ReduceOp.SUM, group=self.group,
                                                   async_op=True))
            lo = max(lo, b)
        if async_op:
            return self._works
        self.finish()
        return None

    def finish(self, work=None) -> None:  # pylint: disable=unused-argument
        """Wait (stream-wise on a GPU) for every slice in flight, then sum -> average with one scale launch."""
        if not self._works:
            return
        for w in self._works:
            w.wait()
        self._works = []
        # the slice bookkeeping belongs to ONE exchange: a second backward + all_reduce without zero_grad() (gradient
        # accumulation, or a caller that zeroes .grad itself) must exchange every slice again
        self._done = []
        if self.world > 1:
            ops.scale_(self.flat, 1.0 / self.world)


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
