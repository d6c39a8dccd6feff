"""Native (HIP-backed) stand-ins for the third-party modules the reference wires in
``/root/reference/gnnepcsaft/train/models.py``: ogb ``AtomEncoder``/``BondEncoder`` (:12, :175-176), PyG ``PNAConv``
(:445-457), ``GINEConv`` (:529-538), ``BatchNorm`` (:17, :184), ``aggr.Sum/Mean/MaxAggregation`` (:587-595).

Attribute and state-dict names follow upstream so that reference checkpoints' ``state_dict``s load unchanged
(SURVEY.md §8b).  Where the reference passes ``edge_index`` / ``edge_attr[E,H]`` these modules take a ``GraphPack``
(dst-sorted CSR with per-edge bond codes) and the 60-row encoded bond table instead — see INTEGRATION.md.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
from torch.nn import Module, ModuleList

from . import functional as Fn
from .data.batching import ATOM_FEATURE_DIMS, BOND_FEATURE_DIMS
from .ops import GraphPack


class Linear(torch.nn.Linear):
    """torch.nn.Linear / PyG Linear: weight [out,in], y = x W^T + b, default (kaiming_uniform a=sqrt 5) init."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # pylint: disable=arguments-renamed
        return Fn.LinearFn.apply(x, self.weight, self.bias)


class ReLU(Module):
    """Placeholder kept for state-dict index compatibility inside Sequential containers; the containers in this
    package fuse the activation into the preceding kernel and never call it."""

    def forward(self, x):  # pragma: no cover - never on the product path
        raise RuntimeError("standalone ReLU is fused into the producing kernel on this path")


class Dropout(Module):
    """torch.nn.Dropout(p) on the HIP path (reference models.py:177, 209).  Identity in eval mode and for p = 0 (no
    launch).  In training mode every call draws a fresh mask from a counter-based generator: key = ``seed`` (taken from
    torch's global seed at construction unless given, so ``torch.manual_seed`` makes runs repeatable), counter = an
    offset that advances by one per call; checkpointing ``(seed, calls)`` resumes the stream exactly.  No parameters
    or buffers: state-dict compatible with the reference's ``dropout`` attribute."""

    def __init__(self, p: float = 0.5, seed: Optional[int] = None):
        super().__init__()
        if p < 0.0 or p > 1.0:
            raise ValueError(f"dropout probability has to be between 0 and 1, but got {p}")
        if p == 1.0:
            raise ValueError("native Dropout implements p in [0, 1) (the reference's configs use 0 and 0.25)")
        self.p = float(p)
        self.seed = int(torch.initial_seed() if seed is None else seed) & (2 ** 63 - 1)
        self.calls = 0

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.training or self.p == 0.0:
            return x
        if x.is_cuda and torch.cuda.is_current_stream_capturing():
            # the mask counter is a host integer passed to the kernel by value: a captured step would replay ONE mask for
            # ever (and the checkpointed counter would stop advancing) -- refuse instead of silently training wrong
            raise RuntimeError("Dropout(p > 0) cannot be captured into a HIP graph: its per-call mask counter lives on the "
                               "host; run the training step eagerly (bench.py --launch eager)")
        self.calls += 1
        return Fn.DropoutFn.apply(x, self.p, self.seed, self.calls)

    def extra_repr(self) -> str:
        return f"p={self.p}"


class _Encoder(Module):
    dims: Sequence[int] = ()
    list_name = ""

    def __init__(self, emb_dim: int):
        super().__init__()
        embs = ModuleList()
        for dim in self.dims:
            emb = torch.nn.Embedding(dim, emb_dim)
            torch.nn.init.xavier_uniform_(emb.weight.data)
            embs.append(emb)
        setattr(self, self.list_name, embs)
        offs = [0]
        for d in self.dims:
            offs.append(offs[-1] + d)
        self.offsets = tuple(offs)

    def _weights(self):
        """The K tables, kept as consecutive row blocks of one buffer (re-packed after ``.to()`` / when something
        re-pointed them) so that the concatenated table is a view; values and state-dict entries are unchanged."""
        ws = [e.weight for e in getattr(self, self.list_name)]
        if ws[0].is_cuda and Fn.adjacent_rows(ws) is None:
            with torch.no_grad():
                big = torch.cat([w.data for w in ws], dim=0)
                for w, o0, o1 in zip(ws, self.offsets[:-1], self.offsets[1:]):
                    w.data = big[o0:o1]
        return ws

    def forward(self, idx: torch.Tensor) -> torch.Tensor:
        return Fn.EmbedSumFn.apply(idx, self.offsets, *self._weights())


class AtomEncoder(_Encoder):
    """[3P] ogb.graphproppred.mol_encoder.AtomEncoder — 9 tables (119,5,12,12,10,6,6,2,2), xavier-uniform."""
    dims = ATOM_FEATURE_DIMS
    list_name = "atom_embedding_list"


class BondEncoder(_Encoder):
    """[3P] ogb BondEncoder — 3 tables (5,6,2).  ``table()`` evaluates all 60 feature combinations once; edges refer
    to rows of it through ``GraphPack.code`` (mixed-radix code of edge_attr)."""
    dims = BOND_FEATURE_DIMS
    list_name = "bond_embedding_list"

    def __init__(self, emb_dim: int):
        super().__init__(emb_dim)
        combos = torch.cartesian_prod(*[torch.arange(d) for d in self.dims]).to(torch.long)
        self.register_buffer("combos", combos, persistent=False)

    def table(self) -> torch.Tensor:
        return Fn.EmbedSumFn.apply(self.combos, self.offsets, *self._weights())


class DegreeScalerAggregation(Module):
    """Holds the ``avg_deg_lin`` / ``avg_deg_log`` buffers of [3P] DegreeScalerAggregation (state-dict names
    ``aggr_module.avg_deg_*``); the aggregation itself is gnx_pna_aggregate_* inside ``PNAConv``."""

    def __init__(self, deg: torch.Tensor):
        super().__init__()
        deg = deg.to(torch.float)
        N = int(deg.sum())
        bin_degree = torch.arange(deg.numel())
        self.init_avg_deg_lin = float((bin_degree * deg).sum()) / N
        self.init_avg_deg_log = float(((bin_degree + 1).log() * deg).sum()) / N
        self.register_buffer("avg_deg_lin", torch.full((1,), self.init_avg_deg_lin))
        self.register_buffer("avg_deg_log", torch.full((1,), self.init_avg_deg_log))
        self._log_cache = None

    def avg_log(self) -> float:
        # buffers only change through load_state_dict / .to(); cache the host copy to keep the hot loop sync-free
        if self._log_cache is None or self._log_cache[0] is not self.avg_deg_log or \
                self._log_cache[1] != self.avg_deg_log._version:
            self._log_cache = (self.avg_deg_log, self.avg_deg_log._version, float(self.avg_deg_log.item()))
        return self._log_cache[2]


class _Seq(torch.nn.Sequential):
    """Sequential of Linear / ReLU used only as a parameter container with upstream indices (0, 2, 4, ...)."""

    def linears(self) -> List[Linear]:
        return [m for m in self if isinstance(m, torch.nn.Linear)]


class PNAConv(Module):
    """[3P] torch_geometric.nn.PNAConv as constructed at models.py:445-457 (aggregators mean/min/max/std, scalers
    identity/amplification/attenuation, divide_input=True)."""

    def __init__(self, in_channels: int, out_channels: int, aggregators: Sequence[str], scalers: Sequence[str],
                 deg: torch.Tensor, edge_dim: int, towers: int = 1, pre_layers: int = 1, post_layers: int = 1,
                 divide_input: bool = False):
        super().__init__()
        if list(aggregators) != ["mean", "min", "max", "std"] or \
                list(scalers) != ["identity", "amplification", "attenuation"]:
            raise ValueError("native PNAConv implements the reference's aggregators=[mean,min,max,std] and "
                             "scalers=[identity,amplification,attenuation] (models.py:443-444)")
        if not divide_input:
            raise ValueError("native PNAConv implements divide_input=True (models.py:456)")
        if in_channels != out_channels:
            raise ValueError("native PNAConv needs in_channels == out_channels (models.py:447-448)")
        if in_channels % towers != 0:
            raise ValueError("in_channels must be divisible by towers")
        if pre_layers < 1 or post_layers < 1:
            raise ValueError("pre_layers and post_layers must be >= 1")
        self.in_channels, self.out_channels, self.towers, self.edge_dim = in_channels, out_channels, towers, edge_dim
        self.pre_layers, self.post_layers = pre_layers, post_layers
        self.F_in = in_channels // towers
        self.F_out = out_channels // towers
        self.aggr_module = DegreeScalerAggregation(deg)
        self.edge_encoder = Linear(edge_dim, self.F_in)
        self.pre_nns = ModuleList()
        self.post_nns = ModuleList()
        for _ in range(towers):
            mods = [Linear(3 * self.F_in, self.F_in)]
            for _ in range(pre_layers - 1):
                mods += [ReLU(), Linear(self.F_in, self.F_in)]
            self.pre_nns.append(_Seq(*mods))
            mods = [Linear(13 * self.F_in, self.F_out)]
            for _ in range(post_layers - 1):
                mods += [ReLU(), Linear(self.F_out, self.F_out)]
            self.post_nns.append(_Seq(*mods))
        self.lin = Linear(out_channels, out_channels)

    def _params(self):
        """The layer's parameters in PNAConvFn's order.  Cached: Parameter objects keep their identity across ``.to()``
        / ``load_state_dict`` (only ``.data`` changes), and Module.__getattr__ is slow enough to matter here (the hot
        loop is host-bound at small batches)."""
        cached = self.__dict__.get("_params_cache")
        if cached is not None and all(a is b for a, b in zip(cached[1], self._parameter_ids())):
            return cached[0]
        params = [self.edge_encoder.weight, self.edge_encoder.bias, self.lin.weight, self.lin.bias]
        for t in range(self.towers):
            for lin in self.pre_nns[t].linears():
                params += [lin.weight, lin.bias]
            for lin in self.post_nns[t].linears():
                params += [lin.weight, lin.bias]
        self.__dict__["_params_cache"] = (params, self._parameter_ids())
        return params

    def _parameter_ids(self):
        # two cheap identity probes: a re-registered parameter (rare: module surgery) invalidates the cache
        return (self.edge_encoder._parameters["weight"], self.lin._parameters["weight"])

    def prepare_ahead(self, edge_index: GraphPack, edge_attr: torch.Tensor) -> "Fn.WeightOnlyAhead":
        """Issue this layer's weight-only work (bond-table chain, Weff(d), merged lin o last post layer) on the side
        stream now; pass ``.wait()`` of the result to ``forward(prepared=...)``.  The model does this one layer ahead."""
        dc = edge_index.degree_classes(edge_index.max_degree_hint) if Fn._USE_DEGREE_CLASSES else None
        return Fn.WeightOnlyAhead(edge_attr, self.towers, self.F_in, self.pre_layers, self.post_layers,
                                  self.aggr_module.avg_log(), self._params(), dc.D if dc is not None else 0)

    @staticmethod
    def prepare_all(convs: Sequence["PNAConv"], edge_index: GraphPack, edge_attr: torch.Tensor) -> "Fn.WeightOnlyAll":
        """The weight-only work of every layer of a stack of identically shaped PNAConv layers in three batched launches
        (on the side stream when enabled); pass ``.get(l)`` to layer l's ``forward(prepared=...)``."""
        c0 = convs[0]
        if any((c.towers, c.F_in, c.pre_layers, c.post_layers) != (c0.towers, c0.F_in, c0.pre_layers, c0.post_layers)
               for c in convs):
            raise ValueError("prepare_all needs identically shaped PNAConv layers")
        dc = edge_index.degree_classes(edge_index.max_degree_hint) if Fn._USE_DEGREE_CLASSES else None
        layers = [(c.aggr_module.avg_log(), c._params()) for c in convs]
        return Fn.WeightOnlyAll(edge_attr, c0.towers, c0.F_in, c0.pre_layers, c0.post_layers, layers,
                                dc.D if dc is not None else 0)

    def forward(self, x: torch.Tensor, edge_index: GraphPack, edge_attr: torch.Tensor, prepared=None, bond_acc=None,
                layer_index: int = 0) -> torch.Tensor:
        """x fp32[N,H]; edge_index: GraphPack of the batch; edge_attr: fp32[60,H] encoded bond table.  ``bond_acc``
        (optional, ``Fn.BondGradAccumulator`` shared by the ``depth`` layers of a model, this one being number
        ``layer_index``): the bond-table gradient is accumulated there off the critical path."""
        cfg = (self.towers, self.F_in, self.pre_layers, self.post_layers, self.aggr_module.avg_log(), prepared,
               bond_acc, layer_index)
        return Fn.PNAConvFn.apply(x, edge_attr, edge_index, cfg, *self._params())


class GINEConv(Module):
    """[3P] torch_geometric.nn.GINEConv(nn, eps=0, train_eps=False, edge_dim) as constructed at models.py:529-538."""

    def __init__(self, nn: torch.nn.Sequential, eps: float = 0.0, train_eps: bool = False,
                 edge_dim: Optional[int] = None):
        super().__init__()
        if train_eps:
            raise ValueError("native GINEConv implements train_eps=False (models.py:536)")
        lins = [m for m in nn if isinstance(m, torch.nn.Linear)]
        if len(nn) != 3 or len(lins) != 2:
            raise ValueError("native GINEConv implements nn = Sequential(Linear, ReLU, Linear) (models.py:531-535)")
        self.nn = nn
        self.initial_eps = eps
        self.register_buffer("eps", torch.full((1,), float(eps)))
        self.lin = Linear(edge_dim, lins[0].in_features)
        self._eps_cache = None

    def eps_value(self) -> float:
        """Host copy of the ``eps`` buffer (a loaded checkpoint may carry a non-default value); cached by buffer
        identity and version so the hot loop stays sync-free, like ``DegreeScalerAggregation.avg_log``."""
        c = self._eps_cache
        if c is None or c[0] is not self.eps or c[1] != self.eps._version:
            self._eps_cache = c = (self.eps, self.eps._version, float(self.eps.item()))
        return c[2]

    def forward(self, x: torch.Tensor, edge_index: GraphPack, edge_attr: torch.Tensor, bond_acc=None,
                layer_index: int = 0) -> torch.Tensor:
        l0, l2 = self.nn[0], self.nn[2]
        return Fn.GINEConvFn.apply(x, edge_attr, edge_index, self.eps_value(), self.lin.weight, self.lin.bias,
                                   l0.weight, l0.bias, l2.weight, l2.bias, bond_acc, layer_index)


class BatchNorm1d(torch.nn.BatchNorm1d):
    """torch.nn.BatchNorm1d over rows (parameters/buffers and their names inherited); ``relu=True`` fuses F.relu."""

    def forward(self, x: torch.Tensor, relu: bool = False) -> torch.Tensor:  # pylint: disable=arguments-differ
        if self.training and x.size(0) <= 1:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")
        use_batch_stats = self.training or not self.track_running_stats
        momentum = 0.0 if self.momentum is None else self.momentum
        nbt = None
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            if self.momentum is None:  # cumulative average: the count is needed on the host (one sync, as in torch)
                self.num_batches_tracked.add_(1)
                momentum = 1.0 / float(self.num_batches_tracked)
            elif self.num_batches_tracked.is_cuda and self.num_batches_tracked.dtype == torch.int64:
                nbt = self.num_batches_tracked  # incremented by the kernel: no separate launch per layer
            else:
                self.num_batches_tracked.add_(1)
        return Fn.BatchNormFn.apply(x, self.weight, self.bias,
                                    self.running_mean if self.track_running_stats else None,
                                    self.running_var if self.track_running_stats else None, momentum, self.eps,
                                    use_batch_stats, relu, nbt)


class BatchNorm(Module):
    """[3P] torch_geometric.nn.BatchNorm: wrapper whose state-dict prefix is ``module.``."""

    def __init__(self, in_channels: int, eps: float = 1e-5, momentum: float = 0.1):
        super().__init__()
        self.module = BatchNorm1d(in_channels, eps, momentum, True, True)

    def forward(self, x: torch.Tensor, relu: bool = False) -> torch.Tensor:
        return self.module(x, relu=relu)


class _SegmentAggregation(Module):
    mode = ""

    def forward(self, x: torch.Tensor, index=None, ptr: Optional[torch.Tensor] = None,
                dim_size: Optional[int] = None, dim: int = -2) -> torch.Tensor:
        """``index`` may be a GraphPack (uses its graph_ptr) — the reference passes ``batch`` here (models.py:219)."""
        if isinstance(index, GraphPack):
            ptr, dim_size = index.graph_ptr, index.B
        if ptr is None or dim_size is None or ptr.dtype != torch.int32:
            raise ValueError("native global pool needs graph_ptr (int32[B+1]) and dim_size; pass the GraphPack")
        return Fn.SegmentPoolFn.apply(x, ptr, int(dim_size), self.mode)


class SumAggregation(_SegmentAggregation):
    mode = "add"


class MeanAggregation(_SegmentAggregation):
    mode = "mean"


class MaxAggregation(_SegmentAggregation):
    mode = "max"
