"""Pure-torch CPU restatement of the reference's GNN hot path (TEST INFRASTRUCTURE, parity unpinned —
see ``oracle/__init__.py``).

Every block cites what it follows.  ``ref:`` paths are relative to ``/root/reference``; ``[3P]`` marks
third-party semantics (torch_geometric 2.x / ogb 1.3.6, not vendored in the reference) restated from their
published source as catalogued in SURVEY.md Appendix A.

Written op-for-op like PyG's CPU path so that, short of PyG itself, it is the closest available proxy:
index_select / cat / F.linear / scatter_add_ / scatter_reduce_(amin|amax, include_self=False) /
F.batch_norm / F.embedding.
"""
from __future__ import annotations

import inspect
import math
from typing import Any, List, Optional, Sequence, Union

import torch
import torch.nn.functional as F
from torch.nn import BatchNorm1d, Dropout, Linear, ModuleList, ReLU, Sequential

# ref: gnnepcsaft/data/ogb_utils.py:8-23 (atom vocab sizes) and :24-33 (bond vocab sizes)
ATOM_FEATURE_DIMS = [119, 5, 12, 12, 10, 6, 6, 2, 2]
BOND_FEATURE_DIMS = [5, 6, 2]


# --------------------------------------------------------------------------------------
# [3P] torch_geometric.utils.scatter / degree (CPU path, no torch_scatter)  — SURVEY A.2
# --------------------------------------------------------------------------------------
def _broadcast(index: torch.Tensor, ref: torch.Tensor, dim: int) -> torch.Tensor:
    size = [1] * ref.dim()
    size[dim] = -1
    return index.view(size).expand_as(ref)


def scatter(src: torch.Tensor, index: torch.Tensor, dim: int, dim_size: int, reduce: str) -> torch.Tensor:
    dim = src.dim() + dim if dim < 0 else dim
    size = list(src.size())
    size[dim] = dim_size
    if reduce in ("sum", "add"):
        return src.new_zeros(size).scatter_add_(dim, _broadcast(index, src, dim), src)
    if reduce == "mean":
        count = src.new_zeros(dim_size)
        count.scatter_add_(0, index, src.new_ones(src.size(dim)))
        count = count.clamp(min=1)
        out = src.new_zeros(size).scatter_add_(dim, _broadcast(index, src, dim), src)
        return out / _broadcast(count, out, dim)
    if reduce in ("min", "max"):
        return src.new_zeros(size).scatter_reduce_(
            dim, _broadcast(index, src, dim), src, reduce=f"a{reduce}", include_self=False
        )
    raise ValueError(reduce)


def degree(index: torch.Tensor, num_nodes: int, dtype=None) -> torch.Tensor:
    out = torch.zeros((num_nodes,), dtype=dtype, device=index.device)
    one = torch.ones((index.size(0),), dtype=out.dtype, device=out.device)
    return out.scatter_add_(0, index, one)


# --------------------------------------------------------------------------------------
# [3P] torch_geometric.nn.aggr  — SURVEY A.2 / A.4
# --------------------------------------------------------------------------------------
class _Aggregation(torch.nn.Module):
    reduce_name = ""

    def forward(self, x, index=None, ptr=None, dim_size=None, dim=-2):
        if dim_size is None:
            dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
        return scatter(x, index, dim, dim_size, self.reduce_name)


class SumAggregation(_Aggregation):
    reduce_name = "sum"


class MeanAggregation(_Aggregation):
    reduce_name = "mean"


class MaxAggregation(_Aggregation):
    reduce_name = "max"


class MinAggregation(_Aggregation):
    reduce_name = "min"


class StdAggregation(torch.nn.Module):
    """var = mean(x*x) - mean(x)^2 (semi_grad=False); std = sqrt(clamp(var,1e-5)), masked to 0 at <= sqrt(1e-5)."""

    def forward(self, x, index=None, ptr=None, dim_size=None, dim=-2):
        mean = scatter(x, index, dim, dim_size, "mean")
        mean2 = scatter(x * x, index, dim, dim_size, "mean")
        var = mean2 - mean * mean
        out = var.clamp(min=1e-5).sqrt()
        out = out.masked_fill(out <= math.sqrt(1e-5), 0.0)
        return out


class DegreeScalerAggregation(torch.nn.Module):
    """[3P] torch_geometric.nn.aggr.DegreeScalerAggregation, aggr/scaler lists from ref: train/models.py:443-444."""

    def __init__(self, aggr: Sequence[str], scaler: Sequence[str], deg: torch.Tensor):
        super().__init__()
        table = {"mean": MeanAggregation, "min": MinAggregation, "max": MaxAggregation, "std": StdAggregation,
                 "sum": SumAggregation}
        self.aggrs = ModuleList([table[a]() for a in aggr])
        self.scaler = list(scaler)
        deg = deg.to(torch.float)
        N = int(deg.sum())
        bin_degree = torch.arange(deg.numel(), device=deg.device)
        self.init_avg_deg_lin = float((bin_degree * deg).sum()) / N
        self.init_avg_deg_log = float(((bin_degree + 1).log() * deg).sum()) / N
        self.register_buffer("avg_deg_lin", torch.empty(1))
        self.register_buffer("avg_deg_log", torch.empty(1))
        self.avg_deg_lin.data.fill_(self.init_avg_deg_lin)
        self.avg_deg_log.data.fill_(self.init_avg_deg_log)

    def forward(self, x, index, dim_size, dim=0):
        out = torch.cat([a(x, index, dim_size=dim_size, dim=dim) for a in self.aggrs], dim=-1)  # mode='cat'
        deg = degree(index, num_nodes=dim_size, dtype=out.dtype)
        size = [1] * out.dim()
        size[dim] = -1
        deg = deg.view(size)
        outs = []
        for s in self.scaler:
            if s == "identity":
                o = out
            elif s == "amplification":
                o = out * (torch.log(deg + 1) / self.avg_deg_log)
            elif s == "attenuation":
                o = out * (self.avg_deg_log / torch.log(deg.clamp(min=1) + 1))
            else:
                raise ValueError(s)
            outs.append(o)
        return torch.cat(outs, dim=-1) if len(outs) > 1 else outs[0]


# --------------------------------------------------------------------------------------
# [3P] ogb.graphproppred.mol_encoder  — SURVEY A.1 ; built at ref: train/models.py:175-176
# --------------------------------------------------------------------------------------
class AtomEncoder(torch.nn.Module):
    def __init__(self, emb_dim: int):
        super().__init__()
        self.atom_embedding_list = ModuleList()
        for dim in ATOM_FEATURE_DIMS:
            emb = torch.nn.Embedding(dim, emb_dim)
            torch.nn.init.xavier_uniform_(emb.weight.data)
            self.atom_embedding_list.append(emb)

    def forward(self, x):
        x_embedding = 0
        for i in range(x.shape[1]):
            x_embedding += self.atom_embedding_list[i](x[:, i])
        return x_embedding


class BondEncoder(torch.nn.Module):
    def __init__(self, emb_dim: int):
        super().__init__()
        self.bond_embedding_list = ModuleList()
        for dim in BOND_FEATURE_DIMS:
            emb = torch.nn.Embedding(dim, emb_dim)
            torch.nn.init.xavier_uniform_(emb.weight.data)
            self.bond_embedding_list.append(emb)

    def forward(self, edge_attr):
        bond_embedding = 0
        for i in range(edge_attr.shape[1]):
            bond_embedding += self.bond_embedding_list[i](edge_attr[:, i])
        return bond_embedding


# --------------------------------------------------------------------------------------
# [3P] torch_geometric.nn.PNAConv — SURVEY A.2 ; constructed at ref: train/models.py:445-457
# --------------------------------------------------------------------------------------
class PNAConv(torch.nn.Module):
    def __init__(self, in_channels, out_channels, aggregators, scalers, deg, edge_dim, towers=1, pre_layers=1,
                 post_layers=1, divide_input=False):
        super().__init__()
        assert divide_input and in_channels % towers == 0 and out_channels % towers == 0
        self.in_channels, self.out_channels, self.towers, self.edge_dim = in_channels, out_channels, towers, edge_dim
        self.F_in = in_channels // towers
        self.F_out = out_channels // towers
        self.aggr_module = DegreeScalerAggregation(aggregators, scalers, deg)
        self.edge_encoder = Linear(edge_dim, self.F_in)
        self.pre_nns = ModuleList()
        self.post_nns = ModuleList()
        for _ in range(towers):
            modules = [Linear(3 * self.F_in, self.F_in)]
            for _ in range(pre_layers - 1):
                modules += [ReLU(), Linear(self.F_in, self.F_in)]
            self.pre_nns.append(Sequential(*modules))
            in_ch = (len(aggregators) * len(scalers) + 1) * self.F_in
            modules = [Linear(in_ch, self.F_out)]
            for _ in range(post_layers - 1):
                modules += [ReLU(), Linear(self.F_out, self.F_out)]
            self.post_nns.append(Sequential(*modules))
        self.lin = Linear(out_channels, out_channels)

    def forward(self, x, edge_index, edge_attr):
        N = x.size(0)
        x = x.view(-1, self.towers, self.F_in)  # divide_input=True
        j, i = edge_index[0], edge_index[1]  # flow source_to_target: j = source, i = target (aggregation index)
        x_i = x.index_select(0, i)
        x_j = x.index_select(0, j)
        e = self.edge_encoder(edge_attr)
        e = e.view(-1, 1, self.F_in).repeat(1, self.towers, 1)
        h = torch.cat([x_i, x_j, e], dim=-1)
        hs = [nn(h[:, t]) for t, nn in enumerate(self.pre_nns)]
        m = torch.stack(hs, dim=1)
        out = self.aggr_module(m, i, dim_size=N, dim=0)
        out = torch.cat([x, out], dim=-1)
        outs = [nn(out[:, t]) for t, nn in enumerate(self.post_nns)]
        out = torch.cat(outs, dim=1)
        return self.lin(out)


# --------------------------------------------------------------------------------------
# [3P] torch_geometric.nn.GINEConv — SURVEY A.3 ; constructed at ref: train/models.py:529-538
# --------------------------------------------------------------------------------------
class GINEConv(torch.nn.Module):
    def __init__(self, nn: torch.nn.Module, eps: float = 0.0, train_eps: bool = False, edge_dim: Optional[int] = None):
        super().__init__()
        assert not train_eps
        self.nn = nn
        self.register_buffer("eps", torch.empty(1))
        self.eps.data.fill_(eps)
        in_channels = nn[0].in_features
        self.lin = Linear(edge_dim, in_channels)

    def forward(self, x, edge_index, edge_attr):
        j, i = edge_index[0], edge_index[1]
        e = self.lin(edge_attr)
        m = (x.index_select(0, j) + e).relu()
        out = scatter(m, i, 0, x.size(0), "sum")
        out = out + (1 + self.eps) * x
        return self.nn(out)


class BatchNorm(torch.nn.Module):
    """[3P] torch_geometric.nn.BatchNorm: wrapper around BatchNorm1d, state-dict prefix ``module.`` — SURVEY A.4."""

    def __init__(self, in_channels, eps=1e-5, momentum=0.1):
        super().__init__()
        self.module = BatchNorm1d(in_channels, eps, momentum, True, True)

    def forward(self, x):
        return self.module(x)


# --------------------------------------------------------------------------------------
# ref: gnnepcsaft/train/models.py:441-606 (factories), :159-254 (model), :77-92 (loss)
# --------------------------------------------------------------------------------------
def get_conv(config):
    aggregators = ["mean", "min", "max", "std"]  # ref: models.py:443
    scalers = ["identity", "amplification", "attenuation"]  # ref: models.py:444
    if config["conv"] == "PNA":
        return PNAConv(config["hidden_dim"], config["hidden_dim"], aggregators, scalers,
                       torch.tensor(config["deg"], dtype=torch.long), config["hidden_dim"], config["towers"],
                       config["pre_layers"], config["post_layers"], divide_input=True)
    if config["conv"] == "GINE":
        h = config["hidden_dim"]
        return GINEConv(Sequential(Linear(h, h), ReLU(), Linear(h, h)), train_eps=False, edge_dim=h)
    raise ValueError(f"Unsupported convolution: {config['conv']}.")


def get_global_pool(config):
    if config["global_pool"] == "mean":
        return MeanAggregation()
    if config["global_pool"] == "max":
        return MaxAggregation()
    if config["global_pool"] == "add":
        return SumAggregation()
    raise ValueError(f"Unsupported global pooling: {config['global_pool']}.")


class GNNePCSAFT(torch.nn.Module):
    """ref: gnnepcsaft/train/models.py:159-254."""

    def __init__(self, config):
        super().__init__()
        self.convs = ModuleList()
        self.batch_norms = ModuleList()
        self.lower_bounds = torch.tensor([1.0, 1.9, 50.0, -1 * math.log10(0.9), math.log10(200.0)])
        self.upper_bounds = torch.tensor([25.0, 4.5, 550.0, -1 * math.log10(0.0001), math.log10(5000.0)])
        self.num_para = config["num_para"]
        self.node_embed = AtomEncoder(config["hidden_dim"])
        self.edge_embed = BondEncoder(config["hidden_dim"])
        self.dropout = Dropout(p=config["dropout"])
        self.global_pool = get_global_pool(config)
        self.global_pool_type = config["global_pool"]
        for _ in range(config["propagation_depth"]):
            self.convs.append(get_conv(config))
            self.batch_norms.append(BatchNorm(config["hidden_dim"]))
        h = config["hidden_dim"]
        self.mlp = Sequential(Linear(h, h // 2), BatchNorm1d(h // 2), ReLU(), Linear(h // 2, h // 4),
                              BatchNorm1d(h // 4), ReLU(), Linear(h // 4, config["num_para"]))

    def forward(self, x, edge_index, edge_attr, batch):
        x = self.node_embed(x)
        edge_attr = self.edge_embed(edge_attr)
        for conv, batch_norm in zip(self.convs, self.batch_norms):
            x = self.dropout(x)
            if "edge_attr" in inspect.signature(conv.forward).parameters:
                x = F.relu(batch_norm(conv(x=x, edge_index=edge_index, edge_attr=edge_attr)))
            else:
                x = F.relu(batch_norm(conv(x=x, edge_index=edge_index)))
        if batch is not None:
            x = self.global_pool(x, batch)
        elif self.global_pool_type == "mean":
            x = x.mean(dim=0, keepdim=True)
        elif self.global_pool_type == "max":
            x = x.max(dim=0, keepdim=True).values
        elif self.global_pool_type == "add":
            x = x.sum(dim=0, keepdim=True)
        return self.mlp(x)

    def pred_with_bounds(self, data):
        x, edge_index, edge_attr, batch = data.x, data.edge_index, data.edge_attr, data.batch
        if isinstance(x, torch.Tensor) and isinstance(edge_index, torch.Tensor) and isinstance(edge_attr, torch.Tensor):
            params = self.forward(x, edge_index, edge_attr, batch)
            upper = (self.upper_bounds[:3] if self.num_para == 3 else self.upper_bounds[3:]).to(device=x.device)
            lower = (self.lower_bounds[:3] if self.num_para == 3 else self.lower_bounds[3:]).to(device=x.device)
            return params.clip(lower, upper)
        raise ValueError("Invalid input data")


def ape_huber_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """ref: gnnepcsaft/train/models.py:89-91."""
    ape = (pred - target) / target
    return F.huber_loss(ape, torch.zeros_like(ape), delta=0.01)


def mape(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """[3P] torchmetrics.functional.mean_absolute_percentage_error (eps = 1.17e-06); ref: models.py:92."""
    return ((pred - target).abs() / target.abs().clamp(min=1.17e-06)).mean()


def training_loss(model: GNNePCSAFT, graphs, dataset: str = "esper") -> torch.Tensor:
    """ref: gnnepcsaft/train/models.py:77-91 (loss part of ``training_step``)."""
    target = graphs.assoc if dataset in ("esper_assoc", "esper_assoc_only") else graphs.para
    pred = model(graphs.x, graphs.edge_index, graphs.edge_attr, graphs.batch)
    return ape_huber_loss(pred, target)


# --------------------------------------------------------------------------------------
# [3P] torch_geometric Batch.from_data_list — SURVEY A.6 ; call site ref: train/train.py:59-75
# --------------------------------------------------------------------------------------
class Data:
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def num_nodes(self):
        return int(self.x.shape[0])


def collate(data_list: List[Data]) -> Data:
    """Concatenate graphs; offset edge_index by the cumulative node count; build batch and ptr (all int64)."""
    xs, eis, eas, batches, paras, assocs = [], [], [], [], [], []
    ptr = [0]
    for g, d in enumerate(data_list):
        n = d.num_nodes
        xs.append(d.x)
        eis.append(d.edge_index + ptr[-1])
        eas.append(d.edge_attr)
        batches.append(torch.full((n,), g, dtype=torch.long))
        if hasattr(d, "para"):
            paras.append(d.para)
        if hasattr(d, "assoc"):
            assocs.append(d.assoc)
        ptr.append(ptr[-1] + n)
    out = Data(x=torch.cat(xs, 0), edge_index=torch.cat(eis, 1), edge_attr=torch.cat(eas, 0),
               batch=torch.cat(batches, 0), ptr=torch.tensor(ptr, dtype=torch.long))
    if paras:
        out.para = torch.cat(paras, 0)
    if assocs:
        out.assoc = torch.cat(assocs, 0)
    out.num_graphs = len(data_list)
    return out


def calc_deg(data_list: List[Data]) -> List[int]:
    """ref: gnnepcsaft/train/utils.py:48-60 (the histogram part; dataset loading is out of scope)."""
    max_degree = -1
    for data in data_list:
        d = degree(data.edge_index[1], num_nodes=data.num_nodes, dtype=torch.long)
        max_degree = max(max_degree, int(d.max()) if d.numel() else 0)
    deg = torch.zeros(max_degree + 1, dtype=torch.long)
    for data in data_list:
        d = degree(data.edge_index[1], num_nodes=data.num_nodes, dtype=torch.long)
        deg += torch.bincount(d, minlength=deg.numel())
    return deg.tolist()
