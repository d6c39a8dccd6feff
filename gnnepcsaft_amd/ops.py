"""Tensor-level wrappers over the C ABI (include/gnx.h): torch tensors in, ``data_ptr()`` + sizes across ctypes.

torch is used for device memory and streams only; every arithmetic op on the hot path is a gnx kernel.  All wrappers
require HIP-device fp32 / int64 / int32 tensors and raise ``GnxError`` otherwise — there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import GemmSeg, check, handle

_I32 = C.c_int32


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise _lib.GnxError(_lib.GNX_E_INVALID, f"{name}: expected float32, got {t.dtype}")
    return t


def _mat(t: torch.Tensor, name: str) -> Tuple[int, int]:
    """(data_ptr, leading dimension) of a 2-D fp32 view whose rows are contiguous."""
    if t.dtype is not torch.float32:
        raise _lib.GnxError(_lib.GNX_E_INVALID, f"{name}: expected float32, got {t.dtype}")
    shape, strides = t.shape, t.stride()
    if len(shape) != 2 or (shape[1] > 1 and strides[1] != 1):
        raise _lib.GnxError(_lib.GNX_E_INVALID, f"{name}: need a 2-D view with unit column stride, got "
                                                f"shape {tuple(shape)} strides {strides}")
    ld = strides[0]
    return t.data_ptr(), (ld if ld > shape[1] else shape[1])


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _carr(vals: Sequence[int]):
    return (_I32 * len(vals))(*vals)


def zeros(*shape: int, device: torch.device) -> torch.Tensor:
    """fp32 zeros through a fill launch: ``torch.zeros`` / ``zero_()`` go through hipMemsetAsync, which costs ~50 us of
    host time per call here (the launch path is on the critical path of the step)."""
    t = torch.empty(*shape, dtype=torch.float32, device=device)
    return zero_(t)


_ONES: dict = {}


def ones_vector(device: torch.device, n: int) -> torch.Tensor:
    """A cached fp32 vector of >= n ones on ``device`` (column sums as 1 x R products in gnx_gemm_small_batched)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    t = _ONES.get(key)
    if t is None or t.numel() < n:
        t = torch.empty(max(n, 256), dtype=torch.float32, device=device)
        check(_lib.load().gnx_fill(handle(device), t.data_ptr(), t.numel(), 1.0))
        _ONES[key] = t
    return t


def zero_(t: torch.Tensor) -> torch.Tensor:
    if t.is_cuda and t.dtype is torch.float32 and t.is_contiguous():
        if t.numel():
            check(_lib.load().gnx_fill(handle(t.device), t.data_ptr(), t.numel(), 0.0))
        return t
    return t.zero_()


# ------------------------------------------------------------------------------------------------------------------
# packer
# ------------------------------------------------------------------------------------------------------------------
class GraphPack:
    """Device-resident integer structure of one batch: dst-sorted CSR + by-source index + bond codes + graph_ptr.

    Everything the kernels need in place of ``edge_index`` / ``batch``.  int32 throughout.
    """

    __slots__ = ("N", "E", "B", "rowptr", "perm", "src", "dst", "colptr", "cpos", "code", "graph_ptr", "device",
                 "_scalers", "has_batch", "_classes", "_code_index", "max_degree_hint", "_edge_tiles")

    EDGE_TILE_ROWS = 64     # message rows of one tile of the fused edge kernels (gnx_pna_edge_fwd)
    EDGE_TILE_MAX_DEG = 16  # above this in-degree bound the tiles would be mostly padding: unfused kernels instead

    def edge_tiles(self, max_degree: int) -> Optional[Tuple[torch.Tensor, int]]:
        """(tile_info int32[2 (count + 1)], tile width) of gnx_edge_tiles for in-degrees <= ``max_degree`` (tile j = the
        nodes whose first CSR position lies in [w j, w (j + 1)), w = 65 - max_degree, so no tile exceeds 64 message
        rows); None when the bound is too large for the fused edge kernels or the batch has no edges.  Built once."""
        if max_degree > GraphPack.EDGE_TILE_MAX_DEG or self.E == 0 or self.N == 0:
            return None
        w = GraphPack.EDGE_TILE_ROWS + 1 - max(int(max_degree), 1)
        hit = self._edge_tiles
        if hit is None or hit[1] != w:
            lib = _lib.load()
            count = lib.gnx_edge_tiles_count(self.E, w)
            info = torch.empty(2 * (count + 1), dtype=torch.int32, device=self.device)
            check(lib.gnx_edge_tiles(handle(self.device), self.rowptr.data_ptr(), self.N, self.E, w, info.data_ptr()))
            hit = self._edge_tiles = (info, w)
        return hit

    def code_index(self, R: int) -> Optional[torch.Tensor]:
        """CSR positions stably grouped by bond code (inverted index for the bond-table gradient); None if R > 64."""
        if R > 64:
            return None
        if self._code_index is None or self._code_index[0] != R:
            lib, h = _lib.load(), handle(self.device)
            pos = torch.empty(max(self.E, 1), dtype=torch.int32, device=self.device)
            ptr = torch.empty(R + 1, dtype=torch.int32, device=self.device)
            nbytes = lib.gnx_degree_classes_workspace_bytes(self.E, R)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            check(lib.gnx_group_by_small_key(h, self.code.data_ptr(), self.E, R, pos.data_ptr(), ptr.data_ptr(),
                                             ws.data_ptr(), nbytes))
            self._code_index = (R, pos, ptr)
        return self._code_index[1]

    def degree_classes(self, max_degree_hint: Optional[int] = None) -> Optional["DegreeClasses"]:
        """Nodes grouped by in-degree (for PNA's per-degree effective post-layer-0 weight); ``None`` when the batch
        has more than 64 distinct degrees up to its maximum (hub-heavy graphs use the ungrouped 4-segment product).
        Built once per batch.  Without a hint the maximum in-degree is read back from the device (one sync, like
        PyG's ``batch.max()``); with ``max_degree_hint`` (e.g. ``len(config["deg"]) - 1``) nothing synchronises and a
        node above the hint sets the sticky range flag reported by ``check_range`` (needed under HIP-graph capture)."""
        if self._classes is None:
            self._classes = DegreeClasses.build(self, max_degree_hint)
        return self._classes if self._classes.D <= DegreeClasses.MAX_D else None

    def degree_scalers(self, avg_deg_log: float) -> Tuple[torch.Tensor, torch.Tensor]:
        key = float(avg_deg_log)
        hit = self._scalers.get(key)
        if hit is None:
            amp = torch.empty(self.N, dtype=torch.float32, device=self.device)
            att = torch.empty(self.N, dtype=torch.float32, device=self.device)
            check(_lib.load().gnx_degree_scalers(handle(self.device), self.rowptr.data_ptr(), self.N, key,
                                                 amp.data_ptr(), att.data_ptr()))
            hit = (amp, att)
            self._scalers[key] = hit
        return hit


class DegreeClasses:
    """dperm / cls_ptr / tile tables of gnx_degree_classes + gnx_class_tiles (see include/gnx.h)."""
    MAX_D = 64
    GEMM_ROWS = 128    # k_gemm's BM
    WGRAD_ROWS = 1024  # rows per weight-gradient chunk (512 -> 1024: -0.1 ms per cfg-2 step, fewer atomic flushes)

    __slots__ = ("D", "dperm", "cls_ptr", "tiles", "ntiles", "max_tiles", "chunks", "nchunks", "max_chunks",
                 "tiles_p", "ntiles_p", "max_tiles_p", "tile_rows_p")

    @staticmethod
    def build(g: "GraphPack", max_degree_hint: Optional[int] = None) -> "DegreeClasses":
        lib, h = _lib.load(), handle(g.device)
        dc = DegreeClasses()
        if max_degree_hint is None:
            mx = _I32(0)
            check(lib.gnx_degree_max(h, g.rowptr.data_ptr(), g.N, C.byref(mx)))
            dc.D = int(mx.value) + 1
        else:
            dc.D = int(max_degree_hint) + 1
        if dc.D > DegreeClasses.MAX_D:
            return dc
        i32 = dict(dtype=torch.int32, device=g.device)
        dc.dperm = torch.empty(max(g.N, 1), **i32)
        dc.cls_ptr = torch.empty(dc.D + 1, **i32)
        nbytes = lib.gnx_degree_classes_workspace_bytes(g.N, dc.D)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=g.device)
        check(lib.gnx_degree_classes(h, g.rowptr.data_ptr(), g.N, dc.D, dc.dperm.data_ptr(), dc.cls_ptr.data_ptr(),
                                     ws.data_ptr(), nbytes))
        dc.max_tiles = g.N // DegreeClasses.GEMM_ROWS + dc.D
        dc.max_chunks = g.N // DegreeClasses.WGRAD_ROWS + dc.D
        dc.tiles = torch.empty(3 * dc.max_tiles, **i32)
        dc.chunks = torch.empty(3 * dc.max_chunks, **i32)
        dc.ntiles = torch.empty(1, **i32)
        dc.nchunks = torch.empty(1, **i32)
        check(lib.gnx_class_tiles(h, dc.cls_ptr.data_ptr(), dc.D, DegreeClasses.GEMM_ROWS, dc.tiles.data_ptr(),
                                  dc.ntiles.data_ptr()))
        check(lib.gnx_class_tiles(h, dc.cls_ptr.data_ptr(), dc.D, DegreeClasses.WGRAD_ROWS, dc.chunks.data_ptr(),
                                  dc.nchunks.data_ptr()))
        # tile table of the pipelined forward product (post-layer 0): 96-row tiles where 128-row ones would leave the last
        # round of its persistent workgroups mostly idle (gnx_gemm_tile_rows); the input-gradient product keeps `tiles`
        dc.tile_rows_p = int(lib.gnx_gemm_tile_rows(h, g.N, 128))
        if dc.tile_rows_p == DegreeClasses.GEMM_ROWS:
            dc.tiles_p, dc.ntiles_p, dc.max_tiles_p = dc.tiles, dc.ntiles, dc.max_tiles
        else:
            dc.max_tiles_p = g.N // dc.tile_rows_p + dc.D
            dc.tiles_p = torch.empty(3 * dc.max_tiles_p, **i32)
            dc.ntiles_p = torch.empty(1, **i32)
            check(lib.gnx_class_tiles(h, dc.cls_ptr.data_ptr(), dc.D, dc.tile_rows_p, dc.tiles_p.data_ptr(),
                                      dc.ntiles_p.data_ptr()))
        return dc


def set_option(device: torch.device, opt: int, value: int) -> int:
    """Set an A/B switch of the device's handle (``_lib.OPT_*``; kernel selection only); returns the previous value."""
    lib, h = _lib.load(), handle(device)
    old = _I32(0)
    check(lib.gnx_get_option(h, opt, C.byref(old)))
    check(lib.gnx_set_option(h, opt, int(value)))
    return int(old.value)


def check_range(device: torch.device) -> None:
    """Synchronise and raise ``GnxError(GNX_E_RANGE)`` if any packer / embedding kernel saw an out-of-range integer."""
    check(_lib.load().gnx_check_range(handle(device)))


def pack_graph(edge_index: torch.Tensor, edge_attr: Optional[torch.Tensor], batch: Optional[torch.Tensor],
               num_nodes: int, num_graphs: Optional[int] = None, bond_dims: Sequence[int] = (5, 6, 2),
               validate: bool = True) -> GraphPack:
    """edge_index int64[2,E], edge_attr int64[E,K], batch int64[N]|None  ->  GraphPack (all on edge_index.device)."""
    dev = edge_index.device
    lib, h = _lib.load(), handle(dev)
    if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.size(0) != 2:
        raise _lib.GnxError(_lib.GNX_E_INVALID, f"edge_index must be int64[2,E], got {edge_index.dtype} "
                                                f"{tuple(edge_index.shape)}")
    edge_index = edge_index.contiguous()
    N, E = int(num_nodes), int(edge_index.size(1))
    i32 = dict(dtype=torch.int32, device=dev)
    g = GraphPack()
    g.N, g.E, g.device, g._scalers, g._classes, g._code_index = N, E, dev, {}, None, None
    g._edge_tiles = None
    g.max_degree_hint = None
    g.rowptr = torch.empty(N + 1, **i32)
    g.colptr = torch.empty(N + 1, **i32)
    g.perm = torch.empty(E, **i32)
    g.src = torch.empty(E, **i32)
    g.dst = torch.empty(E, **i32)
    g.cpos = torch.empty(E, **i32)
    ws_bytes = lib.gnx_pack_csr_workspace_bytes(N, E)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    check(lib.gnx_pack_csr(h, edge_index.data_ptr(), E, N, g.rowptr.data_ptr(), g.perm.data_ptr(), g.src.data_ptr(),
                           g.dst.data_ptr(), g.colptr.data_ptr(), g.cpos.data_ptr(), ws.data_ptr(), ws_bytes))
    g.code = torch.empty(E, **i32)
    if edge_attr is not None:
        if edge_attr.dtype != torch.int64 or edge_attr.dim() != 2 or edge_attr.size(0) != E or \
                edge_attr.size(1) != len(bond_dims):
            raise _lib.GnxError(_lib.GNX_E_INVALID, f"edge_attr must be int64[E,{len(bond_dims)}], got "
                                                    f"{edge_attr.dtype} {tuple(edge_attr.shape)}")
        edge_attr = edge_attr.contiguous()
        check(lib.gnx_feature_code(h, edge_attr.data_ptr(), E, len(bond_dims), _carr(list(bond_dims)),
                                   g.perm.data_ptr(), g.code.data_ptr(), None, 0))
    else:
        g.code.zero_()
    if batch is not None:
        if batch.dtype != torch.int64 or batch.dim() != 1 or batch.size(0) != N:
            raise _lib.GnxError(_lib.GNX_E_INVALID, f"batch must be int64[N], got {batch.dtype} {tuple(batch.shape)}")
        if num_graphs is None:
            # PyG: dim_size = batch.max() + 1 (device sync); pass num_graphs to avoid it
            num_graphs = int(batch.max()) + 1 if N > 0 else 0
        g.B = int(num_graphs)
        g.has_batch = True
        g.graph_ptr = torch.empty(g.B + 1, **i32)
        check(lib.gnx_graph_ptr(h, batch.contiguous().data_ptr(), N, g.B, g.graph_ptr.data_ptr(), None, 0))
    else:
        g.B = 1
        g.has_batch = False
        g.graph_ptr = torch.arange(2, **i32) * N  # [0, N] built on the device (capturable; no host-to-device copy)
    if validate:
        check_range(dev)
    return g


# ------------------------------------------------------------------------------------------------------------------
# embeddings
# ------------------------------------------------------------------------------------------------------------------
def embed_sum_fwd(idx: torch.Tensor, table: torch.Tensor, offsets: Sequence[int]) -> torch.Tensor:
    K = len(offsets) - 1
    if idx.dtype != torch.int64 or idx.dim() != 2 or idx.size(1) != K:
        raise _lib.GnxError(_lib.GNX_E_INVALID, f"embedding index must be int64[N,{K}], got {idx.dtype} "
                                                f"{tuple(idx.shape)}")
    idx = idx.contiguous()
    table = _f32(table, "table").contiguous()
    N, H = idx.size(0), table.size(1)
    out = torch.empty(N, H, dtype=torch.float32, device=table.device)
    check(_lib.load().gnx_embed_sum_fwd(handle(table.device), idx.data_ptr(), N, K, _carr(list(offsets)),
                                        table.data_ptr(), H, out.data_ptr()))
    return out


def embed_sum_bwd(idx: torch.Tensor, offsets: Sequence[int], dout: torch.Tensor,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Gradient of the concatenated tables; ``out`` (fp32 [R,H], contiguous) is accumulated into (+=) when given."""
    K, R = len(offsets) - 1, offsets[-1]
    dout = _f32(dout, "dout").contiguous()
    idx = idx.contiguous()
    H = dout.size(1)
    if out is not None:
        if out.shape != (R, H) or not out.is_contiguous() or out.dtype is not torch.float32:
            raise _lib.GnxError(_lib.GNX_E_INVALID, f"embed_sum_bwd: out must be contiguous fp32 [{R},{H}]")
        dtable = out
    else:
        dtable = zeros(R, H, device=dout.device)
    lib = _lib.load()
    nbytes = lib.gnx_table_scatter_workspace_bytes(idx.size(0), R, H)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dout.device)
    check(lib.gnx_embed_sum_bwd(handle(dout.device), idx.data_ptr(), idx.size(0), K, _carr(list(offsets)), R,
                                dout.data_ptr(), H, dtable.data_ptr(), ws.data_ptr(), nbytes))
    return dtable


# ------------------------------------------------------------------------------------------------------------------
# dense
# ------------------------------------------------------------------------------------------------------------------
Seg = Tuple[torch.Tensor, Optional[torch.Tensor], torch.Tensor]  # (A view [M,k], rowscale [M] | None, B view)


class _SegArrays(threading.local):
    """Per-thread reusable descriptor arrays (forward runs on the caller's thread, backward on autograd's)."""

    def __init__(self):
        self.by_len = {n: (GemmSeg * n)() for n in range(1, 5)}


_SEG_ARRAYS = _SegArrays()


def gemm(segs: Sequence[Seg], out: torch.Tensor, *, bias: Optional[torch.Tensor] = None,
         mask: Optional[torch.Tensor] = None, relu: bool = False, accumulate: bool = False,
         b_trans: bool = True) -> torch.Tensor:
    """out[M,N] (+)= act(sum_s (rs_s * A_s) @ B_s(^T) + bias), optionally * (mask > 0).  See gnx_gemm in gnx.h.

    b_trans=True: B_s is a [N,k] weight view (forward Linear); False: B_s is a [k,N] view (input gradient).
    """
    M, N = out.shape
    if M == 0:
        return out
    cptr, ldc = _mat(out, "out")
    arr = _SEG_ARRAYS.by_len[len(segs)]  # reused: the library copies the descriptors during the call
    for i, (a, rs, b) in enumerate(segs):
        ap, lda = _mat(a, "A")
        bp, ldb = _mat(b, "B")
        (am, k), (b0, b1) = a.shape, b.shape
        if am != M or (b_trans and (b0 != N or b1 != k)) or (not b_trans and (b0 != k or b1 != N)):
            raise _lib.GnxError(_lib.GNX_E_INVALID, f"gemm segment {i}: A {tuple(a.shape)} B {tuple(b.shape)} "
                                                    f"out {tuple(out.shape)} b_trans={b_trans}")
        e = arr[i]
        e.a, e.lda, e.rowscale = ap, lda, (None if rs is None else rs.data_ptr())
        e.b, e.ldb, e.k = bp, ldb, k
    flags = (_lib.GEMM_RELU if relu else 0) | (_lib.GEMM_ACCUMULATE if accumulate else 0) | \
            (_lib.GEMM_B_TRANS if b_trans else 0)
    mp, ldm = (None, 0) if mask is None else _mat(mask, "mask")
    lib, h = _lib.load(), handle(out.device)
    # split weight images of the tiled split-operand kernel live in caller-owned scratch (0 bytes for most calls; the
    # split path needs >= 4096 rows, so small products skip the query: host time matters in tiny-kernel phases)
    nbytes = lib.gnx_gemm_workspace_bytes(h, len(segs), arr, None, 1, M, N, mp, flags, 0) if M >= 4096 else 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=out.device) if nbytes else None
    check(lib.gnx_gemm(h, len(segs), arr, M, N, _ptr(bias), mp, ldm, cptr, ldc, flags, _ptr(ws), nbytes))
    return out


GSeg = Tuple[torch.Tensor, Optional[torch.Tensor], torch.Tensor, int]  # (A view, rowscale|None, B of class 0, stride)


def gemm_grouped(segs: Sequence[GSeg], out: torch.Tensor, dc: DegreeClasses, *, bias: Optional[torch.Tensor] = None,
                 mask: Optional[torch.Tensor] = None, relu: bool = False, b_trans: bool = True,
                 forward_tiles: bool = False) -> torch.Tensor:
    """gemm() over the degree-class tiles of ``dc``: A rows gathered / out rows scattered through ``dc.dperm``; segment
    s reads its weight at ``B_s + class * stride_s`` elements (stride 0 = the same weight for every class).
    ``forward_tiles``: walk ``dc.tiles_p`` (the 96- or 128-row table of the pipelined forward product) instead of the
    128-row table."""
    M, N = out.shape
    if M == 0:
        return out
    cptr, ldc = _mat(out, "out")
    arr = (GemmSeg * len(segs))()
    strides = (C.c_int64 * len(segs))()
    for i, (a, rs, b, stride) in enumerate(segs):
        ap, lda = _mat(a, f"A[{i}]")
        bp, ldb = _mat(b, f"B[{i}]")
        arr[i].a, arr[i].lda, arr[i].rowscale = ap, lda, _ptr(rs)
        arr[i].b, arr[i].ldb, arr[i].k = bp, ldb, a.size(1)
        strides[i] = stride
    flags = (_lib.GEMM_RELU if relu else 0) | (_lib.GEMM_B_TRANS if b_trans else 0)
    mp, ldm = (None, 0) if mask is None else _mat(mask, "mask")
    lib, h = _lib.load(), handle(out.device)
    nbytes = lib.gnx_gemm_workspace_bytes(h, len(segs), arr, strides, dc.D, M, N, mp, flags, 1)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=out.device) if nbytes else None
    if forward_tiles:
        check(lib.gnx_gemm_grouped_rows(h, len(segs), arr, strides, dc.D, M, N, _ptr(bias), mp, ldm, cptr, ldc, flags,
                                        dc.dperm.data_ptr(), dc.tiles_p.data_ptr(), dc.ntiles_p.data_ptr(), dc.max_tiles_p,
                                        _ptr(ws), nbytes, dc.tile_rows_p))
        return out
    check(lib.gnx_gemm_grouped(h, len(segs), arr, strides, dc.D, M, N, _ptr(bias), mp, ldm, cptr, ldc, flags,
                               dc.dperm.data_ptr(), dc.tiles.data_ptr(), dc.ntiles.data_ptr(), dc.max_tiles, _ptr(ws),
                               nbytes))
    return out


def gemm_wgrad_grouped(dC: torch.Tensor, A: torch.Tensor, dW_cls: torch.Tensor, dc: DegreeClasses) -> None:
    """dW_cls[c] (fp32[D, N, K], contiguous) += sum over rows of class c of dC[row]^T A[row]."""
    M, N = dC.shape
    if M == 0:
        return
    xp, ldx = _mat(dC, "dC")
    ap, lda = _mat(A, "A")
    K = A.size(1)
    if dW_cls.shape != (dc.D, N, K) or not dW_cls.is_contiguous():
        raise _lib.GnxError(_lib.GNX_E_INVALID, f"dW_cls must be contiguous [{dc.D},{N},{K}], got {tuple(dW_cls.shape)}")
    check(_lib.load().gnx_gemm_wgrad_grouped(handle(dC.device), xp, ldx, ap, lda, M, N, K, dW_cls.data_ptr(), K, N * K,
                                             dc.dperm.data_ptr(), dc.chunks.data_ptr(), dc.nchunks.data_ptr(),
                                             dc.max_chunks))


def pna_weff(W: torch.Tensor, F: int, D: int, avg_deg_log: float) -> torch.Tensor:
    """Weff fp32[D, F, 4F] from post_nns[t][0].weight ([F, 13F])."""
    wp, ldw = _mat(W, "W")
    out = torch.empty(D, F, 4 * F, dtype=torch.float32, device=W.device)
    check(_lib.load().gnx_pna_weff(handle(W.device), wp, ldw, F, D, float(avg_deg_log), out.data_ptr()))
    return out


def pna_weff_bwd(dWeff: torch.Tensor, F: int, D: int, avg_deg_log: float, dW: torch.Tensor) -> None:
    wp, ldw = _mat(dW, "dW")
    check(_lib.load().gnx_pna_weff_bwd(handle(dW.device), dWeff.data_ptr(), F, D, float(avg_deg_log), wp, ldw))


_SIDE_ENABLED = False
_SIDE_PENDING = set()   # device indices with weight-gradient launches not yet joined
_SIDE_KEEP: dict = {}   # device index -> buffers those launches touch, kept alive until the join


def set_wgrad_side_stream(enabled: bool) -> None:
    """Run weight-gradient kernels on a second HIP stream so they overlap the input-gradient chain (the two are
    independent inside a layer's backward).  A Function whose weight gradients all go into persistent in-place sinks
    (``functional.set_grad_in_place``) defers the join to the end of the whole backward pass
    (``join_side_stream_at_end_of_backward``); a Function that hands a freshly allocated gradient back to autograd joins
    before it returns (``finish_backward``), because AccumulateGrad consumes that tensor on the main stream at once."""
    global _SIDE_ENABLED
    _SIDE_ENABLED = bool(enabled)


_JOIN_QUEUED = False


_SIDE_USED: dict = {}    # device index -> set of side-stream indices with launches not yet joined


def _join(idx: int, which: Optional[int] = None) -> None:
    """Main stream waits for side stream ``which`` of device ``idx`` (None = every side stream in use)."""
    dev = torch.device("cuda", idx)
    used = _SIDE_USED.get(idx, set())
    for w in sorted(used if which is None else (used & {which})):
        check(_lib.load().gnx_side_join_n(handle(dev), w))
        used.discard(w)
    if not used:
        _SIDE_PENDING.discard(idx)
        _SIDE_KEEP.pop(idx, None)  # later users of these blocks are ordered behind the join on the main stream


def _join_all_side_streams() -> None:
    global _JOIN_QUEUED
    _JOIN_QUEUED = False
    for idx in list(_SIDE_PENDING):
        _join(idx)


def join_side_stream_at_end_of_backward() -> None:
    """Called from inside an autograd backward: the main stream waits for the weight-gradient stream ONCE, when the
    autograd engine finishes this backward pass (gradients are only consumed after it), so weight-gradient kernels of
    one layer overlap the input-gradient chain of the next layers too."""
    global _JOIN_QUEUED
    if _SIDE_PENDING and not _JOIN_QUEUED:
        _JOIN_QUEUED = True
        torch.autograd.Variable._execution_engine.queue_callback(_join_all_side_streams)


_WGRAD_DONE_HOOK = None


def set_wgrad_done_hook(fn) -> None:
    """``fn(sinks)`` is called at the end of every conv / Linear backward whose weight gradients all went into
    persistent in-place sinks, right after their launches were issued (on the side stream when enabled): the
    data-parallel exchange uses it to start a layer's slice of the all-reduce while backward continues."""
    global _WGRAD_DONE_HOOK
    _WGRAD_DONE_HOOK = fn


def side_stream(device: torch.device, which: int = 0) -> "torch.cuda.Stream":
    """One of the library's side streams (0 = weight gradients, 1 = bond-table chain) as a torch stream (no ownership)."""
    out = C.c_void_p()
    check(_lib.load().gnx_side_stream_n(handle(device), which, C.byref(out)))
    return torch.cuda.ExternalStream(out.value, device=device)


def second_side_stream_in_use(device: torch.device) -> bool:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return 1 in _SIDE_USED.get(idx, set())


def wgrad_stream_enabled() -> bool:
    return _SIDE_ENABLED


def scale_(t: torch.Tensor, v: float) -> torch.Tensor:
    """t *= v through a gnx launch (flat fp32 HIP buffers; anything else falls back to torch's mul_)."""
    if t.is_cuda and t.dtype is torch.float32 and t.is_contiguous():
        if t.numel():
            check(_lib.load().gnx_scale(handle(t.device), t.data_ptr(), t.numel(), float(v)))
        return t
    return t.mul_(v)


def finish_backward(device: torch.device, all_in_place: bool, sinks=None) -> None:
    """Last call of an autograd backward that issued weight-gradient launches.  ``all_in_place``: every destination of
    those launches is a persistent gradient buffer nobody reads before the backward pass ends -> one deferred join for
    the whole pass.  Otherwise the Function is about to return fresh tensors that autograd accumulates on the main
    stream immediately -> the main stream waits for the side stream now."""
    if all_in_place:
        if _WGRAD_DONE_HOOK is not None and sinks:
            _WGRAD_DONE_HOOK(sinks)
        join_side_stream_at_end_of_backward()
    else:
        join_side_stream(device)


_WGRAD_QUEUE: List[tuple] = []
_WGRAD_BATCHING = True


def set_wgrad_batching(enabled: bool) -> None:
    """A/B switch: queue a layer's weight gradients and launch them as one batched kernel (default on)."""
    global _WGRAD_BATCHING
    _WGRAD_BATCHING = bool(enabled)


def queue_wgrad(dC: torch.Tensor, A: torch.Tensor, dW: torch.Tensor, *, rowscale: Optional[torch.Tensor] = None,
                dbias: Optional[torch.Tensor] = None) -> None:
    """Like gemm_wgrad, but deferred until flush_wgrads() so that independent weight gradients share one launch."""
    if dC.size(0) == 0:
        return
    if not _WGRAD_BATCHING:
        gemm_wgrad(dC, A, dW, rowscale=rowscale, dbias=dbias)
        return
    _WGRAD_QUEUE.append((dC, A, dW, rowscale, dbias))


def flush_wgrads() -> None:
    """Launch every queued weight gradient: up to 8 problems per kernel (gnx_gemm_wgrad_batched), on the side stream
    when enabled.  Problems that do not meet the batched kernel's alignment rules go through gemm_wgrad."""
    global _WGRAD_QUEUE
    jobs, _WGRAD_QUEUE = _WGRAD_QUEUE, []
    if not jobs:
        return
    batchable, rest = [], []
    for j in jobs:
        dC, A, dW, rs, db = j
        ok = all(t is None or t.data_ptr() % 16 == 0 for t in (dC, A)) and dC.stride(0) % 4 == 0 and \
            A.stride(0) % 4 == 0 and dC.size(1) % 4 == 0 and A.size(1) % 4 == 0 and dC.size(0) > 1 and \
            (dC.size(1) == 1 or dC.stride(1) == 1) and (A.size(1) == 1 or A.stride(1) == 1)
        (batchable if ok else rest).append(j)
    for dC, A, dW, rs, db in rest:
        gemm_wgrad(dC, A, dW, rowscale=rs, dbias=db)
    # largest row counts first so that problems of similar size share a launch
    batchable.sort(key=lambda j: -j[0].size(0))
    for i in range(0, len(batchable), 8):
        chunk = batchable[i:i + 8]
        arr = (_lib.WgradProb * len(chunk))()
        keep = []
        for k, (dC, A, dW, rs, db) in enumerate(chunk):
            xp, ldx = _mat(dC, "dC")
            ap, lda = _mat(A, "A")
            wp, ldw = _mat(dW, "dW")
            M, N = dC.shape
            K = A.size(1)
            if A.size(0) != M or dW.size(0) != N or dW.size(1) != K:
                raise _lib.GnxError(_lib.GNX_E_INVALID, f"wgrad shapes dC {tuple(dC.shape)} A {tuple(A.shape)} "
                                                        f"dW {tuple(dW.shape)}")
            arr[k].dC, arr[k].lddc, arr[k].A, arr[k].lda = xp, ldx, ap, lda
            arr[k].rowscale, arr[k].dW, arr[k].lddw, arr[k].dbias = _ptr(rs), wp, ldw, _ptr(db)
            arr[k].M, arr[k].N, arr[k].K = M, N, K
            keep += [dC, A, rs, dW, db]
        ref = chunk[0][0]
        n = len(chunk)
        _run_on_side(ref, keep, lambda arr=arr, n=n, ref=ref: check(
            _lib.load().gnx_gemm_wgrad_batched(handle(ref.device), n, arr)))


def join_side_stream(device: torch.device, which: Optional[int] = None) -> None:
    """Make the current stream wait for the launches issued on the side stream(s) (``which`` = one of them)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx in _SIDE_PENDING:
        _join(idx, which)


def _run_on_side(ref: torch.Tensor, tensors, fn, which: int = 0) -> None:
    """Run ``fn`` (weight-gradient launches) on the library's side stream when enabled, else inline.  ``tensors`` are
    the device buffers the launches read or write; they are kept referenced until the join so the caching allocator
    cannot hand them out again under the side kernels (fork / join are two HIP event calls inside the library:
    gnx_side_begin / gnx_side_join)."""
    if _SIDE_ENABLED and ref.is_cuda:
        idx = ref.device.index
        lib, h = _lib.load(), handle(ref.device)
        check(lib.gnx_side_begin_n(h, which))
        try:
            fn()
        finally:
            check(lib.gnx_side_end(h))
        _SIDE_KEEP.setdefault(idx, []).extend(t for t in tensors if t is not None)
        _SIDE_PENDING.add(idx)
        _SIDE_USED.setdefault(idx, set()).add(which)
        return
    fn()


def run_after_wgrads(ref: torch.Tensor, tensors, fn) -> None:
    """Run ``fn`` (launches that consume a freshly computed weight gradient) on the stream the weight gradients run
    on, i.e. ordered behind them; ``tensors`` are kept alive until the join like every side-stream operand."""
    _run_on_side(ref, tensors, fn)


def keep_until_join(device: torch.device, tensors, whichs=(0, 1)) -> None:
    """Register buffers that launches issued on the side streams by a native composite call (gnx_pna_conv_bwd) read or
    write: they stay referenced until the join, exactly like the operands of ``_run_on_side``."""
    if not _SIDE_ENABLED:
        return
    idx = device.index if device.index is not None else torch.cuda.current_device()
    _SIDE_KEEP.setdefault(idx, []).extend(t for t in tensors if t is not None)
    _SIDE_PENDING.add(idx)
    _SIDE_USED.setdefault(idx, set()).update(whichs)


def pna_conv_bwd(args: "_lib.PnaBwdArgs", device: torch.device) -> None:
    check(_lib.load().gnx_pna_conv_bwd(handle(device), C.byref(args)))


def run_on_second_side_stream(ref: torch.Tensor, tensors, fn) -> None:
    """Run ``fn`` on side stream 1 (inline when side streams are off): the bond-table gradient chain of a conv layer's
    backward -- it forks from the main stream here, feeds parameter gradients and the bond-embedding gradient only, and
    is joined with the rest at the end of backward (or by ``join_side_stream(device, 1)``)."""
    _run_on_side(ref, tensors, fn, which=1)


def gemm_wgrad_inline(dC: torch.Tensor, A: torch.Tensor, dW: torch.Tensor, dbias: Optional[torch.Tensor] = None) -> None:
    """dW += dC^T A (dbias += column sums of dC) on the CURRENT stream of the library handle -- inside
    ``run_after_wgrads`` / ``run_on_second_side_stream`` that is the side stream."""
    _gemm_wgrad_launch(dC, A, dW, None, dbias)


def axpy_(y: torch.Tensor, x: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """y += alpha * x for contiguous fp32 HIP vectors of equal size."""
    if y.numel() != x.numel() or not (y.is_contiguous() and x.is_contiguous()):
        raise _lib.GnxError(_lib.GNX_E_INVALID, "axpy_: need contiguous tensors of equal size")
    check(_lib.load().gnx_axpy(handle(y.device), _f32(y, "y").data_ptr(), _f32(x, "x").data_ptr(), y.numel(),
                               float(alpha)))
    return y


def gemm_wgrad(dC: torch.Tensor, A: torch.Tensor, dW: torch.Tensor, *, rowscale: Optional[torch.Tensor] = None,
               dbias: Optional[torch.Tensor] = None) -> None:
    """dW[N,K] += dC[M,N]^T @ (rowscale * A[M,K]);  dbias[N] += column sums of dC."""
    if dC.size(0) == 0:
        return
    _run_on_side(dC, (dC, A, rowscale, dW, dbias), lambda: _gemm_wgrad_launch(dC, A, dW, rowscale, dbias))


def pna_post0_wgrad_classes(g: torch.Tensor, A: torch.Tensor, dc: "DegreeClasses", F: int, avg_deg_log: float,
                            dW: torch.Tensor) -> None:
    """dW[:, F:13F] += the three A-blocks of post-layer 0's weight gradient, through per-degree-class partial sums."""
    if g.size(0) == 0:
        return
    dWeff = zeros(dc.D, F, 4 * F, device=g.device)

    def run():
        gemm_wgrad_grouped(g, A, dWeff, dc)
        pna_weff_bwd(dWeff, F, dc.D, avg_deg_log, dW)

    _run_on_side(g, (g, A, dWeff, dW), run)


def _gemm_wgrad_launch(dC, A, dW, rowscale, dbias) -> None:
    M, N = dC.shape
    xp, ldx = _mat(dC, "dC")
    ap, lda = _mat(A, "A")
    wp, ldw = _mat(dW, "dW")
    K = A.size(1)
    if A.size(0) != M or dW.size(0) != N or dW.size(1) != K:
        raise _lib.GnxError(_lib.GNX_E_INVALID, f"wgrad shapes dC {tuple(dC.shape)} A {tuple(A.shape)} "
                                                f"dW {tuple(dW.shape)}")
    check(_lib.load().gnx_gemm_wgrad(handle(dW.device), xp, ldx, ap, lda, _ptr(rowscale), M, N, K, wp, ldw,
                                     _ptr(dbias)))


# ------------------------------------------------------------------------------------------------------------------
# message passing
# ------------------------------------------------------------------------------------------------------------------
def edge_combine_fwd(P: torch.Tensor, Q: torch.Tensor, Te: torch.Tensor, g: GraphPack, relu: bool) -> torch.Tensor:
    H = P.size(1)
    h1 = torch.empty(g.E, H, dtype=torch.float32, device=P.device)
    check(_lib.load().gnx_edge_combine_fwd(handle(P.device), P.data_ptr(), Q.data_ptr(), Te.data_ptr(),
                                           g.src.data_ptr(), g.dst.data_ptr(), g.code.data_ptr(), g.E, H, int(relu),
                                           h1.data_ptr()))
    return h1


def edge_combine_bwd_pq(gr: torch.Tensor, g: GraphPack) -> Tuple[torch.Tensor, torch.Tensor]:
    """dP[i] = sum of g over CSR row i, dQ[j] = sum of g over the edges leaving j (no bond-table gradient)."""
    H = gr.size(1)
    dP = torch.empty(g.N, H, dtype=torch.float32, device=gr.device)
    dQ = torch.empty(g.N, H, dtype=torch.float32, device=gr.device)
    check(_lib.load().gnx_edge_combine_bwd(handle(gr.device), gr.data_ptr(), g.rowptr.data_ptr(), g.colptr.data_ptr(),
                                           g.cpos.data_ptr(), g.code.data_ptr(), g.N, g.E, H, 0, dP.data_ptr(),
                                           dQ.data_ptr(), None, None, 0))
    return dP, dQ


def bond_table_grad(gr: torch.Tensor, g: GraphPack, R: int, pos: Optional[torch.Tensor]) -> torch.Tensor:
    """dTe[r] = sum of g over the edges with bond code r; ``pos`` = ``g.code_index(R)`` fetched by the caller (it may
    build the index, which must not happen on a side stream)."""
    H = gr.size(1)
    dTe = zeros(R, H, device=gr.device)
    lib = _lib.load()
    if g.E == 0:
        return dTe
    if pos is not None:
        check(lib.gnx_key_segment_sum(handle(gr.device), gr.data_ptr(), pos.data_ptr(), g.code.data_ptr(), g.E, H,
                                      dTe.data_ptr()))
        return dTe
    # more than 64 codes (never the 60-row bond table): the LDS-privatised scatter inside gnx_edge_combine_bwd
    nbytes = lib.gnx_table_scatter_workspace_bytes(g.E, R, H)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=gr.device)
    tmp = torch.empty(2, g.N, H, dtype=torch.float32, device=gr.device)
    check(lib.gnx_edge_combine_bwd(handle(gr.device), gr.data_ptr(), g.rowptr.data_ptr(), g.colptr.data_ptr(),
                                   g.cpos.data_ptr(), g.code.data_ptr(), g.N, g.E, H, R, tmp[0].data_ptr(),
                                   tmp[1].data_ptr(), dTe.data_ptr(), ws.data_ptr(), nbytes))
    return dTe


def bond_code_index(g: GraphPack, R: int, H: int) -> Optional[torch.Tensor]:
    return g.code_index(R) if ((H % 4 == 0 and H <= 1024) or H <= 256) and g.E > 0 else None


def edge_combine_bwd(gr: torch.Tensor, g: GraphPack, R: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    dP, dQ = edge_combine_bwd_pq(gr, g)
    return dP, dQ, bond_table_grad(gr, g, R, bond_code_index(g, R, gr.size(1)))


def pna_edge_fwd(P: torch.Tensor, Q: torch.Tensor, Te: torch.Tensor, g: GraphPack, T: int, F: int,
                 weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], max_degree: int,
                 keep: bool = True) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor], torch.Tensor]:
    """Fused message assembly -> pre-layer 1 -> aggregate (gnx_pna_edge_fwd): returns (h1 [E,H], m [E,H], A [N,T*4F]);
    ``keep=False`` does not materialise h1 / m (inference).  ``weights`` / ``biases``: pre-layer 1 of every tower."""
    H = T * F
    tiles = g.edge_tiles(max_degree)
    if tiles is None and g.E > 0:
        raise _lib.GnxError(_lib.GNX_E_INVALID, f"pna_edge_fwd: in-degree bound {max_degree} too large for edge tiles")
    dev = P.device
    h1 = torch.empty(g.E, H, dtype=torch.float32, device=dev) if keep else None
    m = torch.empty(g.E, H, dtype=torch.float32, device=dev) if keep else None
    A = torch.empty(g.N, T * 4 * F, dtype=torch.float32, device=dev)
    warr = (C.c_void_p * T)(*[_f32(w, "W1").data_ptr() for w in weights])
    barr = (C.c_void_p * T)(*[_f32(b, "b1").data_ptr() for b in biases])
    for w in weights:
        if tuple(w.shape) != (F, F) or not w.is_contiguous():
            raise _lib.GnxError(_lib.GNX_E_INVALID, f"pna_edge_fwd: pre-layer 1 weight must be contiguous [{F},{F}]")
    info, w_ = tiles if tiles is not None else (None, 1)
    check(_lib.load().gnx_pna_edge_fwd(handle(dev), P.data_ptr(), Q.data_ptr(), Te.data_ptr(), g.src.data_ptr(),
                                       g.dst.data_ptr(), g.code.data_ptr(), g.rowptr.data_ptr(), _ptr(info), w_, g.N, g.E,
                                       T, F, C.cast(warr, C.POINTER(C.c_void_p)), C.cast(barr, C.POINTER(C.c_void_p)),
                                       _ptr(h1), _ptr(m), A.data_ptr()))
    return h1, m, A


def pna_edge_bwd(ge: torch.Tensor, h1: torch.Tensor, g: GraphPack, T: int, F: int, R: int,
                 weights: Sequence[torch.Tensor], max_degree: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Fused masked input gradient of pre-layer 1 + destination sums + bond-table sums (gnx_pna_edge_bwd): returns
    (gh1 [E,H], dP [N,H], dTe [R,H]).  ``weights``: pre-layer 1 ([F,F]) of every tower."""
    H = T * F
    tiles = g.edge_tiles(max_degree)
    if tiles is None and g.E > 0:
        raise _lib.GnxError(_lib.GNX_E_INVALID, f"pna_edge_bwd: in-degree bound {max_degree} too large for edge tiles")
    dev = ge.device
    gh1 = torch.empty(g.E, H, dtype=torch.float32, device=dev)
    dP = torch.empty(g.N, H, dtype=torch.float32, device=dev)
    dTe = zeros(R, H, device=dev)
    warr = (C.c_void_p * T)(*[_f32(w, "W1").data_ptr() for w in weights])
    info, w_ = tiles if tiles is not None else (None, 1)
    check(_lib.load().gnx_pna_edge_bwd(handle(dev), _f32(ge, "ge").contiguous().data_ptr(), _f32(h1, "h1").contiguous().data_ptr(),
                                       g.code.data_ptr(), g.rowptr.data_ptr(), _ptr(info), w_, g.N, g.E, T, F, R,
                                       C.cast(warr, C.POINTER(C.c_void_p)), gh1.data_ptr(), dP.data_ptr(), dTe.data_ptr()))
    return gh1, dP, dTe


def pna_aggregate_fwd(m: torch.Tensor, g: GraphPack, T: int, F: int) -> torch.Tensor:
    A = torch.empty(g.N, T * 4 * F, dtype=torch.float32, device=m.device)
    check(_lib.load().gnx_pna_aggregate_fwd(handle(m.device), m.data_ptr(), g.rowptr.data_ptr(), g.N, g.E, T, F,
                                            A.data_ptr()))
    return A


def pna_aggregate_bwd(dA: torch.Tensor, m: torch.Tensor, A: torch.Tensor, g: GraphPack, T: int, F: int) -> torch.Tensor:
    dm = torch.empty_like(m)
    check(_lib.load().gnx_pna_aggregate_bwd(handle(m.device), dA.data_ptr(), m.data_ptr(), A.data_ptr(),
                                            g.rowptr.data_ptr(), g.N, g.E, T, F, dm.data_ptr()))
    return dm


def gine_aggregate_fwd(x: torch.Tensor, Le: torch.Tensor, g: GraphPack, eps: float) -> torch.Tensor:
    out = torch.empty_like(x)
    check(_lib.load().gnx_gine_aggregate_fwd(handle(x.device), x.data_ptr(), Le.data_ptr(), g.rowptr.data_ptr(),
                                             g.src.data_ptr(), g.code.data_ptr(), g.N, g.E, x.size(1), float(eps),
                                             out.data_ptr()))
    return out


def gine_aggregate_bwd(dout: torch.Tensor, x: torch.Tensor, Le: torch.Tensor, g: GraphPack,
                       eps: float, want_dle: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """dx (and, with ``want_dle``, the bond-table gradient dLe from the same call)."""
    dx = torch.empty_like(x)
    dLe = zeros(*Le.shape, device=Le.device) if want_dle else None
    pos = g.code_index(Le.size(0)) if (g.E > 0 and want_dle) else None  # inverted index by bond code (None above 64 codes)
    check(_lib.load().gnx_gine_aggregate_bwd(handle(x.device), dout.data_ptr(), x.data_ptr(), Le.data_ptr(),
                                             g.colptr.data_ptr(), g.cpos.data_ptr(), g.src.data_ptr(),
                                             g.dst.data_ptr(), g.code.data_ptr(), _ptr(pos), g.N, g.E, x.size(1),
                                             Le.size(0), float(eps), dx.data_ptr(), _ptr(dLe)))
    return dx, dLe


def gine_dle(dout: torch.Tensor, x: torch.Tensor, Le: torch.Tensor, g: GraphPack, pos: Optional[torch.Tensor]) -> torch.Tensor:
    """dLe[r] = sum over the edges with bond code r of dout[dst] * (x[src] + Le[r] > 0), on the handle's current stream."""
    dLe = zeros(*Le.shape, device=Le.device)
    check(_lib.load().gnx_gine_dle(handle(x.device), dout.data_ptr(), x.data_ptr(), Le.data_ptr(), g.src.data_ptr(),
                                   g.dst.data_ptr(), g.code.data_ptr(), _ptr(pos), g.E, x.size(1), Le.size(0),
                                   dLe.data_ptr()))
    return dLe


_POOL = {"add": _lib.POOL_ADD, "sum": _lib.POOL_ADD, "mean": _lib.POOL_MEAN, "max": _lib.POOL_MAX}


def segment_pool_fwd(x: torch.Tensor, ptr: torch.Tensor, B: int, mode: str) -> torch.Tensor:
    out = torch.empty(B, x.size(1), dtype=torch.float32, device=x.device)
    check(_lib.load().gnx_segment_pool_fwd(handle(x.device), x.data_ptr(), ptr.data_ptr(), B, x.size(1), _POOL[mode],
                                           out.data_ptr()))
    return out


def segment_pool_bwd(dout: torch.Tensor, x: torch.Tensor, out: torch.Tensor, ptr: torch.Tensor, B: int,
                     mode: str) -> torch.Tensor:
    dx = torch.empty_like(x)
    check(_lib.load().gnx_segment_pool_bwd(handle(x.device), dout.data_ptr(), x.data_ptr(), out.data_ptr(),
                                           ptr.data_ptr(), B, x.size(1), _POOL[mode], dx.data_ptr()))
    return dx


# ------------------------------------------------------------------------------------------------------------------
# normalisation / loss
# ------------------------------------------------------------------------------------------------------------------
def _bn_ws(M: int, H: int, dev) -> Tuple[torch.Tensor, int]:
    nbytes = _lib.load().gnx_batchnorm_workspace_bytes(M, H)
    return torch.empty(nbytes, dtype=torch.uint8, device=dev), nbytes


def batchnorm_fwd(x, gamma, beta, running_mean, running_var, momentum: float, eps: float, training: bool, relu: bool,
                  num_batches_tracked: Optional[torch.Tensor] = None):
    """``num_batches_tracked`` (int64[1] device tensor) is incremented by the kernel in training mode."""
    M, H = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(H, dtype=torch.float32, device=x.device)
    rstd = torch.empty(H, dtype=torch.float32, device=x.device)
    ws, nbytes = _bn_ws(M, H, x.device)
    check(_lib.load().gnx_batchnorm_fwd(handle(x.device), x.data_ptr(), M, H, _ptr(gamma), _ptr(beta),
                                        _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked),
                                        float(momentum), float(eps),
                                        int(training), int(relu), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                        ws.data_ptr(), nbytes))
    return y, mean, rstd


def batchnorm_bwd(dy, x, y, gamma, mean, rstd, relu: bool, dgamma=None, dbeta=None):
    """dgamma / dbeta: optional existing buffers to accumulate (+=) into; fresh zero buffers otherwise."""
    M, H = x.shape
    dx = torch.empty_like(x)
    if dgamma is None:
        dgamma = zeros(H, device=x.device)
    if dbeta is None:
        dbeta = zeros(H, device=x.device)
    ws, nbytes = _bn_ws(M, H, x.device)
    check(_lib.load().gnx_batchnorm_bwd(handle(x.device), dy.data_ptr(), x.data_ptr(), y.data_ptr(), M, H, _ptr(gamma),
                                        mean.data_ptr(), rstd.data_ptr(), int(relu), dx.data_ptr(), dgamma.data_ptr(),
                                        dbeta.data_ptr(), ws.data_ptr(), nbytes))
    return dx, dgamma, dbeta


def dropout(x: torch.Tensor, p: float, seed: int, offset: int) -> torch.Tensor:
    """x * mask / (1 - p), mask from Philox keyed by (seed, offset); the same call on a gradient is the backward."""
    x = _f32(x, "x").contiguous()
    y = torch.empty_like(x)
    check(_lib.load().gnx_dropout(handle(x.device), x.data_ptr(), x.numel(), float(p), int(seed) & (2 ** 64 - 1),
                                  int(offset) & (2 ** 64 - 1), y.data_ptr()))
    return y


def huber_ape(pred: torch.Tensor, target: torch.Tensor, delta: float, need_grad: bool):
    pred = _f32(pred, "pred").contiguous()
    target = _f32(target, "target").contiguous()
    if pred.shape != target.shape:
        raise _lib.GnxError(_lib.GNX_E_INVALID, f"pred {tuple(pred.shape)} vs target {tuple(target.shape)}")
    out2 = torch.empty(2, dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred) if need_grad else None
    check(_lib.load().gnx_huber_ape(handle(pred.device), pred.data_ptr(), target.data_ptr(), pred.numel(), float(delta),
                                    out2.data_ptr(), _ptr(dpred)))
    return out2, dpred


def clip_rows(x: torch.Tensor, lo: torch.Tensor, hi: torch.Tensor) -> torch.Tensor:
    x = _f32(x, "x").contiguous()
    y = torch.empty_like(x)
    check(_lib.load().gnx_clip_rows(handle(x.device), x.data_ptr(), x.size(0), x.size(1), lo.data_ptr(), hi.data_ptr(),
                                    y.data_ptr()))
    return y


def prof_begin(device: torch.device, kernel_ids: Sequence[int]) -> None:
    """Record HIP event pairs (on the stream the kernels run on) around every launch of the given GNX_K_* kernels."""
    mask = 0
    for k in kernel_ids:
        mask |= 1 << k
    check(_lib.load().gnx_prof_begin(handle(device), mask))


def prof_read(device: torch.device, kernel_id: int) -> Tuple[int, float]:
    """(launches, total milliseconds) of one kernel group since prof_begin; synchronises."""
    d = prof_read_work(device, kernel_id)
    return d["launches"], d["ms"]


def prof_read_work(device: torch.device, kernel_id: int) -> dict:
    """launches, total ms, algorithmic bytes, algorithmic FLOPs and executed bf16-MFMA FLOPs of one kernel group."""
    n, ms = C.c_int64(0), C.c_double(0.0)
    by, fl, mf = C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
    check(_lib.load().gnx_prof_read(handle(device), kernel_id, C.byref(n), C.byref(ms), C.byref(by), C.byref(fl),
                                    C.byref(mf)))
    return {"launches": n.value, "ms": ms.value, "bytes": by.value, "flops": fl.value, "mfma_bf16_flops": mf.value}


def prof_end(device: torch.device) -> None:
    check(_lib.load().gnx_prof_end(handle(device)))
