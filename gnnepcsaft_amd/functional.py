"""torch.autograd.Function wrappers: each forward/backward is a hand-written sequence of gnx kernel launches.

autograd is used as the tape between layers only; no torch arithmetic runs inside these functions (allocation with
``torch.empty/zeros`` is storage plumbing).  Semantics cited per function; ``[3P]`` = torch_geometric / ogb module the
reference wires at ``/root/reference/gnnepcsaft/train/models.py``.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import ops
from .ops import GraphPack


_GRAD_IN_PLACE = False
_USE_DEGREE_CLASSES = True
_MERGE_LAST_POST = True
_PREPARE_AHEAD = True


def set_merge_last_post(enabled: bool) -> None:
    """A/B switch: PNA's last post layer and ``lin`` as ONE product with a merged H x H weight (default on)."""
    global _MERGE_LAST_POST
    _MERGE_LAST_POST = bool(enabled)


def set_prepare_ahead(enabled: bool) -> None:
    """A/B switch: the model issues a PNA layer's weight-only launches one layer ahead on the side stream (default on)."""
    global _PREPARE_AHEAD
    _PREPARE_AHEAD = bool(enabled)


def prepare_ahead_enabled() -> bool:
    return _PREPARE_AHEAD


_BOND_CHAIN_ASIDE = True
_NATIVE_LAYER_BWD = True
_FUSED_EDGE = True
_CLASS_WGRAD_AFTER_AGG = True  # the per-class weight gradient starts behind the aggregate backward, not beside it: 6.640 -> 6.609 ms,
#                                cfg-5 55.88 -> 55.63 (tools/ab_bench.py classearly); at the END of the layer it was worse (6.662 vs 6.579)
_WGRAD_FLUSH_BEFORE_DX = True  # the batched weight gradients of a layer start in front of its dx product: 6.578 -> 6.541 ms, cfg-5
#                                56.03 -> 55.90 (tools/ab_bench.py flushlate)
_TAIL_WGRAD_ALL_CUS = True  # A/B switch: layer 0's batched weight gradients on every CU (tools/ab_bench.py notail)
_TAIL_WGRAD_EARLY = False   # ... and its post-layer ones launched before the edge backward: measured 6.612 vs 6.554 ms (they
#                             compete with the edge backward on the main stream), off


def set_class_wgrad_after_agg(on: bool) -> None:
    global _CLASS_WGRAD_AFTER_AGG
    _CLASS_WGRAD_AFTER_AGG = bool(on)


def set_wgrad_flush_before_dx(on: bool) -> None:
    global _WGRAD_FLUSH_BEFORE_DX
    _WGRAD_FLUSH_BEFORE_DX = bool(on)


def set_tail_wgrad_all_cus(on: bool, early: bool = False) -> None:
    global _TAIL_WGRAD_ALL_CUS, _TAIL_WGRAD_EARLY
    _TAIL_WGRAD_ALL_CUS, _TAIL_WGRAD_EARLY = bool(on), bool(early)

_BATCH_WEIGHT_ONLY = True


def set_batch_weight_only(enabled: bool) -> None:
    """A/B switch: the weight-only work of ALL PNA layers of a model in a few batched launches -- forward: bond-table chain,
    Weff(d), merged lin o last-post weights (gnx_pna_weight_only_all); backward: their gradients, deferred to the end of
    the pass (gnx_pna_stack_finish) -- instead of ~14 small launches per layer and direction (default on)."""
    global _BATCH_WEIGHT_ONLY
    _BATCH_WEIGHT_ONLY = bool(enabled)


def batch_weight_only_enabled() -> bool:
    return _BATCH_WEIGHT_ONLY


def set_fused_edge(enabled: bool) -> None:
    """A/B switch: PNAConv's edge pipeline (message assembly -> pre-layer 1 -> aggregate) as one fused kernel when the
    layer has two pre layers and the batch's in-degree bound admits edge tiles (default on; bit-identical results)."""
    global _FUSED_EDGE
    _FUSED_EDGE = bool(enabled)



def set_native_layer_backward(enabled: bool) -> None:
    """A/B switch: PNAConv's backward as ONE native call (gnx_pna_conv_bwd: the same launches issued from C++, ~0.09
    instead of ~0.32 ms of host time per layer) whenever its preconditions hold; off = launch by launch from Python."""
    global _NATIVE_LAYER_BWD
    _NATIVE_LAYER_BWD = bool(enabled)


def set_bond_chain_aside(enabled: bool) -> None:
    """A/B switch: PNA layers accumulate the bond-embedding gradient into one shared buffer on side stream 1 (default
    on); off = every layer computes it on the main stream and autograd sums the layers' contributions."""
    global _BOND_CHAIN_ASIDE
    _BOND_CHAIN_ASIDE = bool(enabled)


def bond_chain_aside_enabled() -> bool:
    return _BOND_CHAIN_ASIDE


def set_degree_classes(enabled: bool) -> None:
    """A/B switch for PNA's per-degree-class post-layer 0 (default on; off = the 4-segment 13F-wide product)."""
    global _USE_DEGREE_CLASSES
    _USE_DEGREE_CLASSES = bool(enabled)


def set_grad_in_place(enabled: bool) -> None:
    """Opt-in: weight-gradient kernels accumulate (+=) straight into an existing ``param.grad`` (e.g. the views of
    ``dp.FlatGradAllReduce``'s flat buffer) and the Functions return ``None`` for those parameters, instead of
    materialising a zero-filled gradient per parameter per layer for autograd to add.  Same result as autograd's own
    accumulation for the usual ``loss.backward()`` loop; leave it off when calling ``torch.autograd.grad``."""
    global _GRAD_IN_PLACE
    _GRAD_IN_PLACE = bool(enabled)


def grad_sinks(params: Sequence[torch.Tensor]) -> tuple:
    """Per parameter: its ``.grad`` buffer if in-place accumulation is enabled and usable, else ``None``."""
    if not _GRAD_IN_PLACE:
        return tuple(None for _ in params)
    out = []
    for p in params:
        g = p.grad
        ok = g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.shape == p.shape and \
            g.device == p.device
        out.append(g if ok else None)
    return tuple(out)


def _tower_slices(T: int, F: int):
    """Column slice of every tower; ``None`` for the single-tower layout (the whole tensor, no view is created: a view
    costs ~2 us of host time and the backward of one layer would create ~40 of them)."""
    return [None] if T == 1 else [slice(t * F, (t + 1) * F) for t in range(T)]


def _cols(t: torch.Tensor, sl_t):
    return t if sl_t is None else t[:, sl_t]


def _empty(rows: int, cols: int, like: torch.Tensor) -> torch.Tensor:
    return torch.empty(rows, cols, dtype=torch.float32, device=like.device)


def _zeros_like(t: torch.Tensor) -> torch.Tensor:
    return ops.zeros(*t.shape, device=t.device)


def adjacent_rows(tensors: Sequence[torch.Tensor]) -> Optional[torch.Tensor]:
    """If the 2-D fp32 tensors are consecutive row blocks of one allocation (same width, contiguous, back to back),
    returns a [sum rows, H] view over all of them (no copy), else None."""
    t0 = tensors[0]
    if t0.dim() != 2 or t0.dtype is not torch.float32 or not t0.is_contiguous():
        return None
    H = t0.size(1)
    ptr, rows = t0.data_ptr(), 0
    for t in tensors:
        if t.dim() != 2 or t.size(1) != H or t.dtype is not torch.float32 or not t.is_contiguous() or \
                t.device != t0.device or t.data_ptr() != ptr + rows * H * 4:
            return None
        rows += t.size(0)
    if t0.untyped_storage().nbytes() - t0.storage_offset() * 4 < rows * H * 4:
        return None
    return torch.as_strided(t0.detach(), (rows, H), (H, 1))


class EmbedSumFn(torch.autograd.Function):
    """[3P] ogb AtomEncoder/BondEncoder.forward: sum_k Embedding_k(idx[:,k]) (models.py:205-206).

    ``weights`` are the K embedding tables.  When they are consecutive row blocks of one buffer (the encoders keep
    them that way, and so do the flat parameter / gradient buffers) the concatenated table and its gradient are views:
    no ``torch.cat`` in forward, no per-table slice-and-add in backward."""

    @staticmethod
    def forward(ctx, idx, offsets, *weights):
        table = adjacent_rows(weights)
        if table is None:
            table = torch.cat([w.detach() for w in weights], dim=0)
        ctx.save_for_backward(idx)
        ctx.offsets = tuple(offsets)
        ctx.sinks = grad_sinks(weights)
        return ops.embed_sum_fwd(idx, table, offsets)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        sinks, offs = ctx.sinks, ctx.offsets
        if all(g is not None for g in sinks):
            gtable = adjacent_rows(sinks)
            if gtable is not None:  # accumulate straight into the tables' (flat) gradient buffer
                ops.embed_sum_bwd(idx, offs, dout.contiguous(), out=gtable)
                return (None, None) + (None,) * len(sinks)
        dtable = ops.embed_sum_bwd(idx, offs, dout.contiguous())
        return (None, None) + tuple(dtable[offs[k]:offs[k + 1]] for k in range(len(sinks)))


class DropoutFn(torch.autograd.Function):
    """torch.nn.Dropout(p) in training mode (reference models.py:177, 209): y = x * Bernoulli(1 - p) / (1 - p).  The
    mask is a pure function of (seed, offset) (gnx_dropout, Philox4x32-10), so backward recomputes it."""

    @staticmethod
    def forward(ctx, x, p, seed, offset):
        ctx.key = (float(p), int(seed), int(offset))
        return ops.dropout(x, p, seed, offset)

    @staticmethod
    def backward(ctx, dy):
        p, seed, offset = ctx.key
        return ops.dropout(dy, p, seed, offset), None, None, None


class LinearFn(torch.autograd.Function):
    """y = x W^T + b (torch.nn.Linear / PyG Linear; readout MLP models.py:186-194)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.sinks = grad_sinks([weight] + ([bias] if bias is not None else []))
        y = _empty(x.size(0), weight.size(0), x)
        return ops.gemm([(x, None, weight)], y, bias=bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        sw = ctx.sinks[0]
        sb = ctx.sinks[1] if ctx.has_bias else None
        dw = sw if sw is not None else _zeros_like(weight)
        db = None
        if ctx.has_bias:
            db = sb if sb is not None else ops.zeros(weight.size(0), device=dy.device)
        ops.gemm_wgrad(dy, x, dw, dbias=db)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm([(dy, None, weight)], _empty(x.size(0), x.size(1), x), b_trans=False)
        ops.finish_backward(dy.device, sw is not None and (sb is not None or not ctx.has_bias), ctx.sinks)
        return dx, (None if sw is not None else dw), (None if (sb is not None or not ctx.has_bias) else db)


class BatchNormFn(torch.autograd.Function):
    """torch.nn.BatchNorm1d (PyG BatchNorm wrapper; models.py:184, 212-214) with the following F.relu fused."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, training, relu, nbt=None):
        x = x.contiguous()
        y, mean, rstd = ops.batchnorm_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, training, relu,
                                          num_batches_tracked=nbt)
        ctx.save_for_backward(x, y, gamma, mean, rstd)
        ctx.relu, ctx.training = relu, training
        ctx.sinks = grad_sinks([gamma, beta])
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, rstd = ctx.saved_tensors
        if not ctx.training:
            raise NotImplementedError("BatchNorm backward in eval mode is not on the reference's training path")
        sg, sb = ctx.sinks
        dx, dgamma, dbeta = ops.batchnorm_bwd(dy.contiguous(), x, y, gamma, mean, rstd, ctx.relu, dgamma=sg, dbeta=sb)
        return dx, (None if sg is not None else dgamma), (None if sb is not None else dbeta), None, None, None, None, \
            None, None, None


class SegmentPoolFn(torch.autograd.Function):
    """[3P] Sum/Mean/MaxAggregation over ``batch`` (models.py:218-225, 587-595); contiguous segments via graph_ptr."""

    @staticmethod
    def forward(ctx, x, ptr, B, mode):
        x = x.contiguous()
        out = ops.segment_pool_fwd(x, ptr, B, mode)
        ctx.save_for_backward(x, out, ptr)
        ctx.B, ctx.mode = B, mode
        return out

    @staticmethod
    def backward(ctx, dout):
        x, out, ptr = ctx.saved_tensors
        return ops.segment_pool_bwd(dout.contiguous(), x, out, ptr, ctx.B, ctx.mode), None, None, None


class HuberAPEFn(torch.autograd.Function):
    """loss = huber((pred-target)/target, 0, delta=0.01), mean reduction (models.py:89-91); also returns MAPE (:92)."""

    @staticmethod
    def forward(ctx, pred, target, delta):
        out2, dpred = ops.huber_ape(pred, target, delta, need_grad=True)
        ctx.save_for_backward(dpred)
        ctx.mark_non_differentiable(out2)
        return out2[0], out2

    @staticmethod
    def backward(ctx, dloss, _):
        (dpred,) = ctx.saved_tensors
        # dloss is 1 for a plain .backward(); scaling by another scalar is storage-level plumbing
        return dpred * dloss, None, None


def _pna_weight_only(BE, T, F, pre_layers, post_layers, avg_deg_log, params, D):
    """Everything a PNAConv forward needs that depends on the weights alone (plus the number of degree classes D, 0 =
    ungrouped): EE = BondEmb W_enc^T + b_enc [R,F];  Te = per tower EE W_e^T + b (the edge part of pre-layer 0 on the
    60-row bond table) [R,H];  Weff(d) per tower;  the merged (lin o last post layer) weight / bias."""
    R, H = BE.size(0), T * F
    merged = post_layers > 1 and _MERGE_LAST_POST
    if _NATIVE_LAYER_BWD and BE.is_cuda and T <= 8:
        # one native call (gnx_pna_weight_only) instead of 5 + 3 (T - 1) launches from Python
        import ctypes as C
        from . import _lib
        EE, Te = _empty(R, F, BE), _empty(R, H, BE)
        weffs = [torch.empty(D, F, 4 * F, dtype=torch.float32, device=BE.device) for _ in range(T)] if D > 0 else []
        Wm = _empty(H, H, BE) if merged else None
        bm = torch.empty(H, dtype=torch.float32, device=BE.device) if merged else None
        n = len(params)
        parr = (C.c_void_p * n)(*[p.data_ptr() for p in params])
        warr = (C.c_void_p * max(T, 1))(*[w.data_ptr() for w in weffs]) if weffs else None
        ops.check(_lib.load().gnx_pna_weight_only(
            _lib.handle(BE.device), BE.data_ptr(), R, T, F, pre_layers, post_layers, D, float(avg_deg_log),
            C.cast(parr, C.POINTER(C.c_void_p)), int(merged), EE.data_ptr(), Te.data_ptr(),
            None if warr is None else C.cast(warr, C.POINTER(C.c_void_p)), None if Wm is None else Wm.data_ptr(),
            None if bm is None else bm.data_ptr()))
        return EE, Te, weffs, Wm, bm
    enc_w, enc_b, lin_w, lin_b = params[:4]
    per = 2 * (pre_layers + post_layers)
    sl = _tower_slices(T, F)
    EE = ops.gemm([(BE, None, enc_w)], _empty(R, F, BE), bias=enc_b)
    Te = _empty(R, H, BE)
    weffs, last = [], []
    for t in range(T):
        W0, b0 = params[4 + t * per], params[4 + t * per + 1]
        ops.gemm([(EE, None, W0[:, 2 * F:3 * F])], _cols(Te, sl[t]), bias=b0)
        Wp = params[4 + t * per + 2 * pre_layers]
        if D > 0:
            weffs.append(ops.pna_weff(Wp, F, D, avg_deg_log))
        k = 4 + t * per + 2 * (pre_layers + post_layers - 1)
        last.append((params[k], params[k + 1]))
    Wm = bm = None
    if merged:
        Wm, bm = _merge_last_post_with_lin(lin_w, lin_b, last, sl, BE)
    return EE, Te, weffs, Wm, bm


class BondGradAccumulator:
    """One [R, H] buffer per model forward into which every conv layer's backward ACCUMULATES its bond-embedding
    gradient on side stream 1 (in-order there, so the adds never race); only layer 0 -- whose backward runs last --
    joins that stream and returns the total to autograd.  ``depth`` = number of conv layers sharing the bond table."""

    def __init__(self, R: int, H: int, device, depth: int):
        self.buf = torch.empty(R, H, dtype=torch.float32, device=device)
        self.depth = depth
        self.encoder = None  # (combos, offsets, tables) of the BondEncoder whose table() this gradient belongs to
        # deferred small work (functional.set_batch_weight_only): every PNA layer's backward leaves its 60-row bond-table
        # chain, its lin o last-post un-merge and its Weff gradient to ONE batched finish at the end of the pass
        self.defer = False
        self.slab = None      # zero-filled fp32 slab: [acc.buf | per layer: dTe, dEE, dWm, dbm, dWeff x T]
        self.deferred = []    # per deferring layer: dict of what gnx_pna_stack_finish needs (keeps the tensors alive)

    def layer_slab(self, layer_index: int, R: int, H: int, F: int, T: int, D: int, device):
        """This layer's zeroed accumulators (dTe [R,H], dEE [R,F], dWm [H,H], dbm [H], dWeff [T,D,F,4F]) inside the slab
        shared by the model's layers: ONE fill launch per backward pass, issued by the layer whose backward runs first
        (which also stands in for clearing ``buf``, the slab's first R x H entries)."""
        per = R * H + R * F + H * H + H + T * D * F * 4 * F
        if self.slab is None:
            self.slab = ops.zeros(R * H + self.depth * per, device=device)
            self.buf = self.slab[:R * H].view(R, H)
        o = R * H + layer_index * per
        out = []
        for n in (R * H, R * F, H * H, H, T * D * F * 4 * F):
            out.append(self.slab[o:o + n])
            o += n
        return out

    def finish_deferred(self, device) -> None:
        """gnx_pna_stack_finish over the layers that deferred (side stream 1: bond chain tails; side stream 0: un-merge and
        Weff gradients, behind the layers' weight gradients)."""
        if not self.deferred:
            return
        import ctypes as C
        from . import _lib
        d0 = self.deferred[0]
        L = len(self.deferred)
        T, F, pre, post, R, D, merged = d0["T"], d0["F"], d0["pre"], d0["post"], d0["R"], d0["D"], d0["merged"]
        np_ = 4 + T * 2 * (pre + post)
        a = _lib.PnaFinishArgs()
        a.L, a.T, a.F, a.pre_layers, a.post_layers, a.R, a.D, a.merged = L, T, F, pre, post, R, D, int(merged)
        a.use_side_streams = int(ops.wgrad_stream_enabled())
        ones = ops.ones_vector(device, max(R, 1))
        a.BE, a.acc_buf, a.ones = d0["BE"].data_ptr(), self.buf.data_ptr(), ones.data_ptr()
        avg = (C.c_float * L)(*[d["avg"] for d in self.deferred])
        vp = C.c_void_p
        params = (vp * (L * np_))(*[p.data_ptr() for d in self.deferred for p in d["params"]])
        grads = (vp * (L * np_))(*[g.data_ptr() for d in self.deferred for g in d["sinks"]])
        EE = (vp * L)(*[d["EE"].data_ptr() for d in self.deferred])
        dTe = (vp * L)(*[d["dTe"].data_ptr() for d in self.deferred])
        dEE = (vp * L)(*[d["dEE"].data_ptr() for d in self.deferred])
        dWm = (vp * L)(*[d["dWm"].data_ptr() for d in self.deferred])
        dbm = (vp * L)(*[d["dbm"].data_ptr() for d in self.deferred])
        per_w = D * F * 4 * F
        dWeff = (vp * (L * T))(*[d["dWeff"].data_ptr() + 4 * t * per_w for d in self.deferred for t in range(T)])
        a.avg_deg_log = C.cast(avg, C.POINTER(C.c_float))
        for name, arr in (("params", params), ("grads", grads), ("EE", EE), ("dTe", dTe), ("dEE", dEE), ("dWm", dWm),
                          ("dbm", dbm), ("dWeff", dWeff)):
            setattr(a, name, C.cast(arr, C.POINTER(vp)))
        ops.check(_lib.load().gnx_pna_stack_finish(_lib.handle(device), C.byref(a)))
        ops.keep_until_join(device, [self.slab, ones] + [t for d in self.deferred for t in (d["BE"], d["EE"], *d["params"], *d["sinks"])])
        self.deferred = []

    def first_in_backward(self, layer_index: int) -> bool:
        return layer_index == self.depth - 1

    def handoff(self, device) -> Optional[torch.Tensor]:
        """Called by the conv layer whose backward runs last.  If the bond encoder's tables have in-place gradient
        sinks (``set_grad_in_place`` + a flat gradient buffer), the encoder's own backward -- a scatter of the 60
        accumulated rows into the three tables -- is issued right here on side stream 1, behind the chain, and autograd
        gets ``None`` for the bond table: the main stream never waits for the chain.  Otherwise the main stream joins
        side stream 1 and the accumulated gradient goes back to autograd as usual."""
        if self.encoder is not None:
            idx, offs, weights = self.encoder
            sinks = grad_sinks(weights)
            gtable = adjacent_rows(sinks) if all(g is not None for g in sinks) else None
            if gtable is not None:
                buf = self.buf
                ops.run_on_second_side_stream(buf, [buf, idx, gtable], lambda: ops.embed_sum_bwd(idx, offs, buf, out=gtable))
                return None
        ops.join_side_stream(device, 1)
        return self.buf


class WeightOnlyAhead:
    """``_pna_weight_only`` issued on the library's side stream (when enabled) so that its ~5 tiny dependent launches
    leave the critical path: the model issues layer l+1's while layer l runs.  ``wait()`` orders the caller's stream
    behind them and returns the tensors."""

    def __init__(self, BE, T, F, pre_layers, post_layers, avg_deg_log, params, D):
        self.device = BE.device
        box = []
        # raw kernel launches on .data pointers: nothing here is recorded by autograd, so no detach() is needed
        ops.run_after_wgrads(BE, (), lambda: box.append(_pna_weight_only(BE, T, F, pre_layers, post_layers, avg_deg_log,
                                                                        params, D)))
        self.value = box[0]

    def wait(self):
        ops.join_side_stream(self.device)
        return self.value


class WeightOnlyAll:
    """``_pna_weight_only`` of ALL PNA layers of a model in three launches (gnx_pna_weight_only_all: batched 60-row products
    + batched Weff), issued on the library's side stream when enabled; ``get(l)`` orders the caller's stream behind them
    (once) and returns layer l's (EE, Te, weffs, Wm, bm).  ``layers``: per layer (avg_deg_log, params) with identical
    T / F / pre_layers / post_layers."""

    def __init__(self, BE, T, F, pre_layers, post_layers, layers, D):
        import ctypes as C
        from . import _lib
        self.device = BE.device
        L, R, H = len(layers), BE.size(0), T * F
        merged = post_layers > 1 and _MERGE_LAST_POST
        f32 = dict(dtype=torch.float32, device=BE.device)
        EE = torch.empty(L, R, F, **f32)
        Te = torch.empty(L, R, H, **f32)
        weff = torch.empty(L, T, D, F, 4 * F, **f32) if D > 0 else None
        Wm = torch.empty(L, H, H, **f32) if merged else None
        bm = torch.empty(L, H, **f32) if merged else None
        vp = C.c_void_p
        np_ = len(layers[0][1])
        avg = (C.c_float * L)(*[float(a_) for a_, _ in layers])
        params = (vp * (L * np_))(*[p.data_ptr() for _, ps in layers for p in ps])
        ptrs = lambda t_, n: (vp * n)(*[t_[i].data_ptr() for i in range(n)]) if t_ is not None else None  # noqa: E731
        a_EE, a_Te, a_Wm, a_bm = ptrs(EE, L), ptrs(Te, L), ptrs(Wm, L), ptrs(bm, L)
        a_weff = (vp * (L * T))(*[weff[l, t].data_ptr() for l in range(L) for t in range(T)]) if weff is not None else None
        cast = lambda arr: None if arr is None else C.cast(arr, C.POINTER(vp))  # noqa: E731

        def launch():
            ops.check(_lib.load().gnx_pna_weight_only_all(
                _lib.handle(BE.device), L, BE.data_ptr(), R, T, F, pre_layers, post_layers, D,
                C.cast(avg, C.POINTER(C.c_float)), cast(params), int(merged), cast(a_EE), cast(a_Te), cast(a_weff),
                cast(a_Wm), cast(a_bm)))

        keep = [BE, EE, Te, weff, Wm, bm] + [p for _, ps in layers for p in ps]
        ops.run_after_wgrads(BE, keep, launch)
        self.values = [(EE[l], Te[l], [weff[l, t] for t in range(T)] if weff is not None else [],
                        Wm[l] if merged else None, bm[l] if merged else None) for l in range(L)]
        self._joined = False

    def get(self, l: int):
        if not self._joined:
            ops.join_side_stream(self.device)
            self._joined = True
        return self.values[l]


def _merge_last_post_with_lin(lin_w, lin_b, last, sl, like):
    """Wm [H,H], bm [H] with  lin(cat_t(post_last_t(z_t))) = z Wm^T + bm  (``last`` = [(W_t, b_t)] per tower)."""
    H = lin_w.size(0)
    Wm, bm = _empty(H, H, like), _empty(1, H, like)
    for t, (Wt, _) in enumerate(last):
        ops.gemm([(_cols(lin_w, sl[t]), None, Wt)], _cols(Wm, sl[t]), b_trans=False)       # [H,F] @ [F,F]
    for t, (_, bt) in enumerate(last):  # bm = lin_b + sum_t b_t @ lin_w[:, t]^T   (no torch.cat: may run on the side stream)
        ops.gemm([(bt.view(1, -1), None, _cols(lin_w, sl[t]))], bm, bias=lin_b if t == 0 else None, accumulate=t > 0)
    return Wm, bm.view(-1)


def _unmerge_last_post_and_lin(dWm, dbm, lin_w, d_lin_w, d_lin_b, last, last_bias, sl):
    """Accumulates (+=) the gradients of ``lin`` and of the last post layer from those of the merged product
    (u_t = z_t W_t^T + b_t is the last post layer's output, which is never formed):
    d lin_w[:, t] += dout^T u_t = dWm[:, t] W_t^T + dbm b_t^T ;  dW_t += lin_w[:, t]^T dWm[:, t] ;
    db_t += dbm lin_w[:, t] ;  d lin_b += dbm."""
    row, col = dbm.view(1, -1), dbm.view(-1, 1)
    for t, (Wt, dWt, dbt) in enumerate(last):
        bt = last_bias[t]
        ops.gemm([(_cols(dWm, sl[t]), None, Wt)], _cols(d_lin_w, sl[t]), accumulate=True)                   # [H,F] @ [F,F]^T
        ops.gemm([(col, None, bt.view(-1, 1))], _cols(d_lin_w, sl[t]), accumulate=True)                  # [H,1] @ [F,1]^T
        ops.gemm_wgrad_inline(_cols(lin_w, sl[t]), _cols(dWm, sl[t]), dWt)                                   # [H,F]^T [H,F]
        ops.gemm([(row, None, _cols(lin_w, sl[t]))], dbt.view(1, -1), b_trans=False, accumulate=True)    # [1,H] @ [H,F]
    ops.axpy_(d_lin_b, dbm)


def _pna_forward_native(ctx, x, BE, pack, cfg, params, dc, prep):
    """PNAConvFn.forward through gnx_pna_conv_fwd (one call for the ~12 launches after the weight-only part)."""
    import ctypes as C
    from . import _lib
    T, F, pre_layers, post_layers, avg_deg_log = cfg[:5]
    EE, Te, weffs, Wm, bm = prep
    N, H = x.shape
    E, D = pack.E, dc.D
    dev = x.device
    merged = Wm is not None
    n_z = post_layers - 1 if merged else post_layers
    f32 = dict(dtype=torch.float32, device=dev)
    PQ = torch.empty(2, N, H, **f32)
    hs_all = torch.empty(pre_layers, max(E, 1), H, **f32)
    zs_all = torch.empty(max(n_z, 1), N, H, **f32)
    A = torch.empty(N, T * 4 * F, **f32)
    out = torch.empty(N, H, **f32)
    lib = _lib.load()
    ws_bytes = lib.gnx_pna_conv_bwd_workspace_bytes(T, F, D)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    a = _lib.PnaFwdArgs()
    a.N, a.E, a.T, a.F, a.pre_layers, a.post_layers, a.D, a.merged = N, E, T, F, pre_layers, post_layers, D, int(merged)
    a.rowptr, a.src, a.dst, a.code = pack.rowptr.data_ptr(), pack.src.data_ptr(), pack.dst.data_ptr(), pack.code.data_ptr()
    a.dperm, a.tiles, a.ntiles, a.max_tiles = dc.dperm.data_ptr(), dc.tiles_p.data_ptr(), dc.ntiles_p.data_ptr(), dc.max_tiles_p
    a.tile_rows = dc.tile_rows_p
    a.x, a.Te = x.data_ptr(), Te.data_ptr()
    for i, w in enumerate(weffs):
        a.weff[i] = w.data_ptr()
    a.Wm, a.bm = (Wm.data_ptr(), bm.data_ptr()) if merged else (None, None)
    n = len(params)
    parr = (C.c_void_p * n)(*[p.data_ptr() for p in params])
    a.params = C.cast(parr, C.POINTER(C.c_void_p))
    a.P, a.Q, a.A = PQ[0].data_ptr(), PQ[1].data_ptr(), A.data_ptr()
    hs = [hs_all[i][:E] if E != hs_all.size(1) else hs_all[i] for i in range(pre_layers)]
    zs = [zs_all[i] for i in range(n_z)]
    for i, t_ in enumerate(hs):
        a.hs[i] = t_.data_ptr()
    for i, t_ in enumerate(zs):
        a.zs[i] = t_.data_ptr()
    a.ws, a.ws_bytes, a.out = ws.data_ptr(), ws_bytes, out.data_ptr()
    # edge tiles of the fused gather -> pre-layer 1 -> aggregate kernel (D - 1 bounds the in-degree: a batch above the
    # hint trips the range flag); the library falls back to the three-launch sequence when it is not eligible
    etiles = pack.edge_tiles(D - 1) if (_FUSED_EDGE and pre_layers == 2) else None
    a.etile_info, a.etile_w = (etiles[0].data_ptr(), etiles[1]) if etiles is not None else (None, 0)
    ops.check(lib.gnx_pna_conv_fwd(_lib.handle(dev), C.byref(a)))
    amp, att = pack.degree_scalers(avg_deg_log)
    ctx.pack, ctx.cfg = pack, cfg[:5]
    ctx.n_h, ctx.n_z = len(hs), len(zs)
    ctx.sinks = grad_sinks(params)
    ctx.dc, ctx.weffs, ctx.Wm = dc, weffs, Wm
    ctx.save_for_backward(x, BE, EE, A, amp, att, *hs, *zs, *params)
    return out


def _pna_backward_native(ctx, dout, x, BE, EE, A, hs, zs, params, sinks, code_pos):
    """PNAConvFn.backward through gnx_pna_conv_bwd: Python only allocates the temporaries and fills the argument block."""
    import ctypes as C
    from . import _lib
    T, F, pre_layers, post_layers, avg_deg_log = ctx.cfg
    pack, dc, acc = ctx.pack, ctx.dc, ctx.bond_acc
    N, H = x.shape
    E, R, D = pack.E, BE.size(0), dc.D
    dev = x.device
    dout = dout.contiguous()
    merged = ctx.Wm is not None
    n_g = post_layers - 1 if merged else post_layers
    f32 = dict(dtype=torch.float32, device=dev)
    gbuf = torch.empty(max(n_g, 1), N, H, **f32)
    gebuf = torch.empty(pre_layers, max(E, 1), H, **f32)
    dA = torch.empty(N, T * 4 * F, **f32)
    pq = torch.empty(2, N, H, **f32)
    # small accumulators: private scratch, or (deferral) zeroed slices of the model's slab that stay alive until the ONE
    # batched finish at the end of the pass; a data-parallel exchange that hands every layer's slice to RCCL the moment
    # its launches are issued (ops.set_wgrad_done_hook) needs the layer's gradients complete here: no deferral then
    defer = _BATCH_WEIGHT_ONLY and acc is not None and ops._WGRAD_DONE_HOOK is None and \
        (acc.slab is not None or acc.first_in_backward(ctx.layer_index))  # pylint: disable=protected-access
    if defer:
        dTe_t, dEE_t, dWm_t, dbm_t, dWeff_t = acc.layer_slab(ctx.layer_index, R, H, F, T, D, dev)
        small = None
    else:
        small = torch.empty(R * H + R * F + H * H + H + T * D * F * 4 * F, **f32)
    dx = torch.empty(N, H, **f32)
    lib = _lib.load()
    ws_bytes = lib.gnx_pna_conv_bwd_workspace_bytes(T, F, D)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    a = _lib.PnaBwdArgs()
    a.N, a.E, a.T, a.F, a.pre_layers, a.post_layers, a.R, a.D = N, E, T, F, pre_layers, post_layers, R, D
    a.avg_deg_log, a.merged = float(avg_deg_log), int(merged)
    a.acc_first, a.use_side_streams = int(acc.first_in_backward(ctx.layer_index)), int(ops.wgrad_stream_enabled())
    a.n_h, a.n_z = len(hs), len(zs)
    a.rowptr, a.colptr, a.cpos, a.code = pack.rowptr.data_ptr(), pack.colptr.data_ptr(), pack.cpos.data_ptr(), pack.code.data_ptr()
    a.code_pos = None if code_pos is None else code_pos.data_ptr()
    a.dperm, a.tiles, a.ntiles = dc.dperm.data_ptr(), dc.tiles.data_ptr(), dc.ntiles.data_ptr()
    a.chunks, a.nchunks, a.max_tiles, a.max_chunks = dc.chunks.data_ptr(), dc.nchunks.data_ptr(), dc.max_tiles, dc.max_chunks
    a.x, a.BE, a.EE, a.A = x.data_ptr(), BE.data_ptr(), EE.data_ptr(), A.data_ptr()
    for i, t_ in enumerate(hs):
        a.hs[i] = t_.data_ptr()
    for i, t_ in enumerate(zs):
        a.zs[i] = t_.data_ptr()
    for i, t_ in enumerate(ctx.weffs):
        a.weff[i] = t_.data_ptr()
    a.Wm = ctx.Wm.data_ptr() if merged else None
    n = len(params)
    parr = (C.c_void_p * n)(*[p.data_ptr() for p in params])
    garr = (C.c_void_p * n)(*[g_.data_ptr() for g_ in sinks])
    a.params, a.grads, a.dout = C.cast(parr, C.POINTER(C.c_void_p)), C.cast(garr, C.POINTER(C.c_void_p)), dout.data_ptr()
    for i in range(gbuf.size(0)):
        a.gbuf[i] = gbuf[i].data_ptr()
    for i in range(pre_layers):
        a.gebuf[i] = gebuf[i].data_ptr()
    a.dA, a.dP, a.dQ = dA.data_ptr(), pq[0].data_ptr(), pq[1].data_ptr()
    if defer:
        a.dTe, a.dEE, a.dWm, a.dbm, a.dWeff = (t_.data_ptr() for t_ in (dTe_t, dEE_t, dWm_t, dbm_t, dWeff_t))
        a.defer_small = 1
        acc.deferred.append(dict(T=T, F=F, pre=pre_layers, post=post_layers, R=R, D=D, merged=merged, avg=float(avg_deg_log),
                                 BE=BE, EE=EE, params=list(params), sinks=list(sinks), dTe=dTe_t, dEE=dEE_t, dWm=dWm_t,
                                 dbm=dbm_t, dWeff=dWeff_t))
    else:
        base, o = small.data_ptr(), 0
        a.dTe, o = base + 4 * o, o + R * H
        a.dEE, o = base + 4 * o, o + R * F
        a.dWm, o = base + 4 * o, o + H * H
        a.dbm, o = base + 4 * o, o + H
        a.dWeff = base + 4 * o
        a.defer_small = 0
    if _CLASS_WGRAD_AFTER_AGG:
        a.defer_small |= 8
    if _WGRAD_FLUSH_BEFORE_DX:
        a.defer_small |= 16
    if ctx.layer_index == 0 and _TAIL_WGRAD_ALL_CUS:
        a.defer_small |= 6 if _TAIL_WGRAD_EARLY else 2  # the last conv backward of the pass: its weight gradients may take
        # every CU, and the ones of the post layers go out before the edge backward instead of at the end
    a.ws, a.ws_bytes, a.acc_buf, a.dx = ws.data_ptr(), ws_bytes, acc.buf.data_ptr(), dx.data_ptr()
    etiles = pack.edge_tiles(D - 1) if (_FUSED_EDGE and pre_layers == 2) else None  # (the forward's table, cached on the pack)
    a.etile_info, a.etile_w = (etiles[0].data_ptr(), etiles[1]) if etiles is not None else (None, 0)
    ops.pna_conv_bwd(a, dev)
    # everything the side-stream launches touch stays alive until the join at the end of backward
    ops.keep_until_join(dev, [dout, x, BE, EE, A, *hs, *zs, *ctx.weffs, ctx.Wm, gbuf, gebuf, dA, pq, small, ws, acc.buf,
                              acc.slab, *params, *sinks])
    dBE = None
    if ctx.layer_index == 0:  # the last conv backward of the pass: the deferred small work, then the bond encoder's own
        acc.finish_deferred(dev)
        dBE = acc.handoff(dev)
    ops.finish_backward(dev, True, sinks)
    return (dx, dBE, None, None, *([None] * len(params)))


class PNAConvFn(torch.autograd.Function):
    """[3P] torch_geometric.nn.PNAConv(aggregators=[mean,min,max,std], scalers=[identity,amplification,attenuation],
    towers=T, pre_layers, post_layers, divide_input=True) as built at models.py:445-457.

    Restructured for the GPU (same sums, re-associated):
      * pre-layer 0, ``Linear(3F->F)`` on cat([x_i, x_j, e]), is split into node-level products P = x W_i^T,
        Q = x W_j^T and a 60-row bond table Te = (BondEmb W_enc^T + b_enc) W_e^T + b, combined per edge by a gather;
      * the 13F-wide post-layer-0 operand [x | A | amp*A | att*A] is never materialised: it is a 4-segment product.
    params: enc_w, enc_b, lin_w, lin_b, then per tower: pre (w, b) x pre_layers, post (w, b) x post_layers.
    """

    @staticmethod
    def forward(ctx, x, BE, pack: GraphPack, cfg, *params):
        T, F, pre_layers, post_layers, avg_deg_log, prep = cfg[:6]
        ctx.bond_acc, ctx.layer_index = (cfg[6], cfg[7]) if len(cfg) > 6 else (None, 0)
        x = x.contiguous()
        N, H = x.shape
        E, R = pack.E, BE.size(0)
        enc_w, enc_b, lin_w, lin_b = params[:4]
        per = 2 * (pre_layers + post_layers)
        pre = [[(params[4 + t * per + 2 * i], params[4 + t * per + 2 * i + 1]) for i in range(pre_layers)]
               for t in range(T)]
        post = [[(params[4 + t * per + 2 * (pre_layers + i)], params[4 + t * per + 2 * (pre_layers + i) + 1])
                 for i in range(post_layers)] for t in range(T)]
        sl = _tower_slices(T, F)
        # degree classes: amp/att depend on the in-degree only, so with rows grouped by degree class the 12F-wide scaled
        # operand of post-layer 0 collapses to A @ Weff(d)^T; hub-heavy batches (> 64 classes) keep 4 segments
        dc = pack.degree_classes(pack.max_degree_hint) if _USE_DEGREE_CLASSES else None
        # everything that depends on the WEIGHTS only (60-row bond-table chain, Weff(d), lin o last post layer): taken
        # from ``prep`` when the caller evaluated it ahead of time on the side stream (pna_weight_only_async)
        if prep is None:
            prep = _pna_weight_only(BE, T, F, pre_layers, post_layers, avg_deg_log, params, dc.D if dc is not None else 0)
        EE, Te, weffs, Wm, bm = prep
        if _NATIVE_LAYER_BWD and dc is not None and x.is_cuda and pre_layers <= 8 and post_layers <= 8 and T <= 8:
            return _pna_forward_native(ctx, x, BE, pack, cfg, params, dc, prep)
        P, Q = _empty(N, H, x), _empty(N, H, x)
        for t in range(T):
            W0 = pre[t][0][0]
            xt = _cols(x, sl[t])
            ops.gemm([(xt, None, W0[:, 0:F])], _cols(P, sl[t]))
            ops.gemm([(xt, None, W0[:, F:2 * F])], _cols(Q, sl[t]))
        h = ops.edge_combine_fwd(P, Q, Te, pack, relu=pre_layers > 1)
        hs = [h]
        for i in range(1, pre_layers):
            hn = _empty(E, H, x)
            for t in range(T):
                Wi, bi = pre[t][i]
                ops.gemm([(_cols(h, sl[t]), None, Wi)], _cols(hn, sl[t]), bias=bi, relu=i < pre_layers - 1)
            h = hn
            hs.append(h)
        A = ops.pna_aggregate_fwd(h, pack, T, F)
        amp, att = pack.degree_scalers(avg_deg_log)
        z = _empty(N, H, x)
        for t in range(T):
            Wp, bp = post[t][0]
            At = A[:, t * 4 * F:(t + 1) * 4 * F]
            if dc is not None:
                ops.gemm_grouped([(_cols(x, sl[t]), None, Wp[:, 0:F], 0), (At, None, weffs[t][0], 4 * F * F)], _cols(z, sl[t]), dc,
                                 bias=bp, relu=post_layers > 1)
            else:
                ops.gemm([(_cols(x, sl[t]), None, Wp[:, 0:F]), (At, None, Wp[:, F:5 * F]), (At, amp, Wp[:, 5 * F:9 * F]),
                          (At, att, Wp[:, 9 * F:13 * F])], _cols(z, sl[t]), bias=bp, relu=post_layers > 1)
        zs = [z]
        merged = Wm is not None
        for i in range(1, post_layers - 1 if merged else post_layers):
            zn = _empty(N, H, x)
            for t in range(T):
                Wi, bi = post[t][i]
                ops.gemm([(_cols(z, sl[t]), None, Wi)], _cols(zn, sl[t]), bias=bi, relu=i < post_layers - 1)
            z = zn
            zs.append(z)
        if merged:
            # The last post layer has no activation and feeds ``lin`` directly: lin(post_last(z)) = z Wm^T + bm with
            # Wm[:, tower t] = lin_w[:, tower t] @ W_last_t  and  bm = lin_w @ b_last + lin_b  (H x H, from the weights
            # alone) -- one [N, H] product and one [N, H] round trip less, forward and backward.
            out = ops.gemm([(z, None, Wm)], _empty(N, H, x), bias=bm)
        else:
            out = ops.gemm([(z, None, lin_w)], _empty(N, H, x), bias=lin_b)
        ctx.pack, ctx.cfg = pack, cfg[:5]
        ctx.n_h, ctx.n_z = len(hs), len(zs)
        ctx.sinks = grad_sinks(params)
        ctx.dc, ctx.weffs, ctx.Wm = dc, weffs, Wm
        ctx.save_for_backward(x, BE, EE, A, amp, att, *hs, *zs, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        T, F, pre_layers, post_layers, _ = ctx.cfg
        pack: GraphPack = ctx.pack
        saved = ctx.saved_tensors
        x, BE, EE, A, amp, att = saved[:6]
        hs = list(saved[6:6 + ctx.n_h])
        zs = list(saved[6 + ctx.n_h:6 + ctx.n_h + ctx.n_z])
        params = saved[6 + ctx.n_h + ctx.n_z:]
        N, H = x.shape
        E, R = pack.E, BE.size(0)
        enc_w, enc_b, lin_w, lin_b = params[:4]
        per = 2 * (pre_layers + post_layers)
        sinks = ctx.sinks
        if _NATIVE_LAYER_BWD and ctx.bond_acc is not None and ctx.dc is not None and x.is_cuda and \
                all(sk is not None for sk in sinks) and pre_layers <= 8 and post_layers <= 8 and T <= 8:
            code_pos = ops.bond_code_index(pack, R, H)
            if code_pos is not None or E == 0:
                return _pna_backward_native(ctx, dout, x, BE, EE, A, hs, zs, params, sinks, code_pos)
        grads = [sk if sk is not None else _zeros_like(p) for p, sk in zip(params, sinks)]
        d_enc_w, d_enc_b, d_lin_w, d_lin_b = grads[:4]

        def pidx(t, kind, i):  # index of (w) in params/grads
            return 4 + t * per + 2 * (i if kind == "pre" else pre_layers + i)

        sl = _tower_slices(T, F)
        dout = dout.contiguous()
        merged = None
        if ctx.Wm is not None:
            # lin o post_last as one product with Wm (see forward): dWm = dout^T z, dbm = sum dout, dz = dout Wm masked by
            # z > 0; the gradients of lin and of the last post layer follow from dWm / dbm by H x H products (queued
            # behind the weight-gradient launch that produces dWm)
            dWm, dbm = _zeros_like(ctx.Wm), ops.zeros(H, device=x.device)
            ops.queue_wgrad(dout, zs[-1], dWm, dbias=dbm)
            g = ops.gemm([(dout, None, ctx.Wm)], _empty(N, H, x), b_trans=False, mask=zs[-1])
            merged = (dWm, dbm)
            last_hidden = post_layers - 2
        else:
            ops.queue_wgrad(dout, zs[-1], d_lin_w, dbias=d_lin_b)
            g = ops.gemm([(dout, None, lin_w)], _empty(N, H, x), b_trans=False)
            last_hidden = post_layers - 1
        # hidden post layers last..1 : dgrad masked by the relu'd input activation
        for i in range(last_hidden, 0, -1):
            a_prev = zs[i - 1]
            gn = _empty(N, H, x)
            for t in range(T):
                k = pidx(t, "post", i)
                ops.queue_wgrad(_cols(g, sl[t]), _cols(a_prev, sl[t]), grads[k], dbias=grads[k + 1])
                ops.gemm([(_cols(g, sl[t]), None, params[k])], _cols(gn, sl[t]), b_trans=False, mask=_cols(a_prev, sl[t]))
            g = gn
        # post layer 0: 4-segment weight gradient, 3-segment dA
        dA = _empty(N, T * 4 * F, x)
        for t in range(T):
            k = pidx(t, "post", 0)
            Wp, dWp = params[k], grads[k]
            gt = _cols(g, sl[t])
            At = A[:, t * 4 * F:(t + 1) * 4 * F]
            ops.queue_wgrad(gt, _cols(x, sl[t]), dWp[:, 0:F], dbias=grads[k + 1])
            if ctx.dc is not None:
                ops.pna_post0_wgrad_classes(gt, At, ctx.dc, F, ctx.cfg[4], dWp)
                ops.gemm_grouped([(gt, None, ctx.weffs[t][0], 4 * F * F)], dA[:, t * 4 * F:(t + 1) * 4 * F], ctx.dc,
                                 b_trans=False)
            else:
                ops.queue_wgrad(gt, At, dWp[:, F:5 * F])
                ops.queue_wgrad(gt, At, dWp[:, 5 * F:9 * F], rowscale=amp)
                ops.queue_wgrad(gt, At, dWp[:, 9 * F:13 * F], rowscale=att)
                ops.gemm([(gt, None, Wp[:, F:5 * F]), (gt, amp, Wp[:, 5 * F:9 * F]), (gt, att, Wp[:, 9 * F:13 * F])],
                         dA[:, t * 4 * F:(t + 1) * 4 * F], b_trans=False)
        ge = ops.pna_aggregate_bwd(dA, hs[-1], A, pack, T, F)
        for i in range(pre_layers - 1, 0, -1):
            h_prev = hs[i - 1]
            gn = _empty(E, H, x)
            for t in range(T):
                k = pidx(t, "pre", i)
                ops.queue_wgrad(_cols(ge, sl[t]), _cols(h_prev, sl[t]), grads[k], dbias=grads[k + 1])
                ops.gemm([(_cols(ge, sl[t]), None, params[k])], _cols(gn, sl[t]), b_trans=False, mask=_cols(h_prev, sl[t]))
            ge = gn
        dP, dQ = ops.edge_combine_bwd_pq(ge, pack)
        dx = _empty(N, H, x)
        for t in range(T):
            k0 = pidx(t, "pre", 0)
            W0, dW0 = params[k0], grads[k0]
            kp = pidx(t, "post", 0)
            xt = _cols(x, sl[t])
            ops.queue_wgrad(_cols(dP, sl[t]), xt, dW0[:, 0:F])
            ops.queue_wgrad(_cols(dQ, sl[t]), xt, dW0[:, F:2 * F])
            ops.gemm([(_cols(g, sl[t]), None, params[kp][:, 0:F]), (_cols(dP, sl[t]), None, W0[:, 0:F]),
                      (_cols(dQ, sl[t]), None, W0[:, F:2 * F])], _cols(dx, sl[t]), b_trans=False)
        # Bond-table gradient chain: dTe (by-code segment sum over the 84 MB message gradient) -> the edge slice of
        # pre-layer 0 -> edge_encoder -> the bond-embedding gradient.  Nothing on the way to dx depends on it, so it runs
        # on side stream 1 (forked here, i.e. behind the kernel that produced ``ge``) and leaves the critical path.
        code_pos = ops.bond_code_index(pack, R, H)   # (may build the inverted index: on the main stream, before the fork)
        acc = ctx.bond_acc
        dBE_box = []
        keep = [ge, EE, BE] + ([acc.buf] if acc is not None else [])  # + the chain's temporaries (appended below)

        def bond_chain():
            dTe = ops.bond_table_grad(ge, pack, R, code_pos)
            dEE = _empty(R, F, x)
            for t in range(T):
                k0 = pidx(t, "pre", 0)
                ops.gemm_wgrad_inline(_cols(dTe, sl[t]), EE, grads[k0][:, 2 * F:3 * F], dbias=grads[k0 + 1])
                ops.gemm([(_cols(dTe, sl[t]), None, params[k0][:, 2 * F:3 * F])], dEE, b_trans=False, accumulate=t > 0)
            ops.gemm_wgrad_inline(dEE, BE, d_enc_w, dbias=d_enc_b)
            if acc is None:
                dBE_box.append(ops.gemm([(dEE, None, enc_w)], _empty(R, H, x), b_trans=False))
            else:  # the layers of one model share one accumulator; the layer whose backward runs first clears it
                if acc.first_in_backward(ctx.layer_index):
                    ops.zero_(acc.buf)
                ops.gemm([(dEE, None, enc_w)], acc.buf, b_trans=False, accumulate=True)
            keep.extend((dTe, dEE))

        if acc is None:
            bond_chain()  # stand-alone use of the layer: the caller gets dBE back at once, so no side stream
            dBE = dBE_box[0]
        else:
            ops.run_on_second_side_stream(ge, keep, bond_chain)
            dBE = None
            if ctx.layer_index == 0:  # the last conv backward of the pass hands the accumulated gradient to autograd
                dBE = acc.handoff(x.device)
        ops.flush_wgrads()  # the layer's weight gradients in batched launches on the weight-gradient stream
        if merged is not None:
            dWm, dbm = merged
            last = [(params[pidx(t, "post", post_layers - 1)], grads[pidx(t, "post", post_layers - 1)],
                     grads[pidx(t, "post", post_layers - 1) + 1]) for t in range(T)]
            last_bias = [params[pidx(t, "post", post_layers - 1) + 1] for t in range(T)]
            ops.run_after_wgrads(dout, (dWm, dbm, lin_w, d_lin_w, d_lin_b),
                                 lambda: _unmerge_last_post_and_lin(dWm, dbm, lin_w, d_lin_w, d_lin_b, last, last_bias, sl))
        ops.finish_backward(x.device, all(sk is not None for sk in sinks), sinks)
        return (dx, dBE, None, None, *[None if sk is not None else g_ for g_, sk in zip(grads, sinks)])


class GINEConvFn(torch.autograd.Function):
    """[3P] torch_geometric.nn.GINEConv(nn=Seq(Linear, ReLU, Linear), eps=0, train_eps=False, edge_dim=H) as built at
    models.py:529-538.  lin(edge_attr) is evaluated on the 60-row bond table; message, ReLU and
    sum-aggregation are one gather kernel (no [E,H] message tensor)."""

    @staticmethod
    def forward(ctx, x, BE, pack: GraphPack, eps, lin_w, lin_b, w0, b0, w2, b2, bond_acc=None, layer_index=0):
        ctx.bond_acc, ctx.layer_index = bond_acc, layer_index
        x = x.contiguous()
        N, H = x.shape
        R = BE.size(0)
        Le = ops.gemm([(BE, None, lin_w)], _empty(R, lin_w.size(0), x), bias=lin_b)
        agg = ops.gine_aggregate_fwd(x, Le, pack, eps)
        a1 = ops.gemm([(agg, None, w0)], _empty(N, w0.size(0), x), bias=b0, relu=True)
        out = ops.gemm([(a1, None, w2)], _empty(N, w2.size(0), x), bias=b2)
        ctx.pack, ctx.eps = pack, eps
        ctx.sinks = grad_sinks([lin_w, lin_b, w0, b0, w2, b2])
        ctx.save_for_backward(x, BE, Le, agg, a1, lin_w, w0, w2)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, BE, Le, agg, a1, lin_w, w0, w2 = ctx.saved_tensors
        pack: GraphPack = ctx.pack
        dout = dout.contiguous()
        N, H = x.shape
        dev = dict(dtype=torch.float32, device=x.device)
        sk = ctx.sinks
        dlw = sk[0] if sk[0] is not None else _zeros_like(lin_w)
        dlb = sk[1] if sk[1] is not None else ops.zeros(lin_w.size(0), device=x.device)
        dw0 = sk[2] if sk[2] is not None else _zeros_like(w0)
        db0 = sk[3] if sk[3] is not None else ops.zeros(w0.size(0), device=x.device)
        dw2 = sk[4] if sk[4] is not None else _zeros_like(w2)
        db2 = sk[5] if sk[5] is not None else ops.zeros(w2.size(0), device=x.device)
        ops.gemm_wgrad(dout, a1, dw2, dbias=db2)
        g1 = ops.gemm([(dout, None, w2)], _empty(N, a1.size(1), x), b_trans=False, mask=a1)
        ops.gemm_wgrad(g1, agg, dw0, dbias=db0)
        dagg = ops.gemm([(g1, None, w0)], _empty(N, H, x), b_trans=False)
        acc = ctx.bond_acc
        if acc is None:  # stand-alone layer: everything on the main stream, dBE handed back at once
            dx, dLe = ops.gine_aggregate_bwd(dagg, x, Le, pack, ctx.eps)
            ops.gemm_wgrad(dLe, BE, dlw, dbias=dlb)
            dBE = ops.gemm([(dLe, None, lin_w)], _empty(BE.size(0), BE.size(1), x), b_trans=False)
        else:
            # dx continues on the main stream; the bond-table gradient (a by-code segment sum over 2 x [E, H] of gathered
            # rows: 0.45 ms per layer at cfg-3) and what hangs off it run on side stream 1 into the shared accumulator
            dx, _ = ops.gine_aggregate_bwd(dagg, x, Le, pack, ctx.eps, want_dle=False)
            code_pos = pack.code_index(Le.size(0)) if pack.E > 0 else None  # built (once per batch) on the main stream
            keep = [dagg, x, Le, BE, acc.buf]

            def bond_chain():
                dLe_ = ops.gine_dle(dagg, x, Le, pack, code_pos)
                ops.gemm_wgrad_inline(dLe_, BE, dlw, dbias=dlb)
                if acc.first_in_backward(ctx.layer_index):
                    ops.zero_(acc.buf)
                ops.gemm([(dLe_, None, lin_w)], acc.buf, b_trans=False, accumulate=True)
                keep.append(dLe_)

            ops.run_on_second_side_stream(dagg, keep, bond_chain)
            dBE = None
            if ctx.layer_index == 0:
                dBE = acc.handoff(x.device)
        ops.finish_backward(x.device, all(k is not None for k in sk), sk)
        outs = [None if k is not None else g_ for g_, k in zip((dlw, dlb, dw0, db0, dw2, db2), sk)]
        return (dx, dBE, None, None, *outs, None, None)
