"""Host side of the input packer: ``Data`` / ``Batch.from_data_list`` / ``calc_deg``.

Replaces, for the hot path only, what the reference gets from torch_geometric's ``DataLoader`` →
``Batch.from_data_list`` (call sites ``/root/reference/gnnepcsaft/train/train.py:19,59-75``) and
``calc_deg`` (``/root/reference/gnnepcsaft/train/utils.py:37-60``).  Integer work only; results are bit-exact
with the oracle's collate (tests/test_host_cpu.py::test_collate_bit_exact_with_oracle).  The device-side CSR packing lives in ``csrc/gnx_pack.hip``.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence

import torch

# vocabulary sizes = embedding-table rows (/root/reference/gnnepcsaft/data/ogb_utils.py:8-34)
ATOM_FEATURE_DIMS = (119, 5, 12, 12, 10, 6, 6, 2, 2)
BOND_FEATURE_DIMS = (5, 6, 2)


class Data:
    """Minimal stand-in for ``torch_geometric.data.Data``: the fields the hot path reads
    (``/root/reference/gnnepcsaft/train/models.py:78-87``): x, edge_index, edge_attr, batch, para, assoc."""

    def __init__(self, x=None, edge_index=None, edge_attr=None, **kwargs):
        self.x, self.edge_index, self.edge_attr = x, edge_index, edge_attr
        self.batch = None
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])

    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None and not k.startswith("_")]

    def to(self, device, non_blocking: bool = False):
        out = self.__class__.__new__(self.__class__)
        for k, v in self.__dict__.items():
            setattr(out, k, v.to(device, non_blocking=non_blocking) if isinstance(v, torch.Tensor) else v)
        return out

    def __repr__(self):
        items = ", ".join(f"{k}={list(v.shape) if isinstance(v, torch.Tensor) else v}" for k, v in self.__dict__.items()
                          if v is not None and not k.startswith("_"))
        return f"{self.__class__.__name__}({items})"


class Batch(Data):
    """Disjoint union of graphs (PyG ``Batch`` layout, all index tensors int64)."""

    @classmethod
    def from_data_list(cls, data_list: Sequence[Data]) -> "Batch":
        if len(data_list) == 0:
            raise ValueError("empty data_list")
        n = torch.tensor([d.num_nodes for d in data_list], dtype=torch.long)
        ptr = torch.zeros(len(data_list) + 1, dtype=torch.long)
        torch.cumsum(n, 0, out=ptr[1:])
        e = torch.tensor([d.num_edges for d in data_list], dtype=torch.long)
        edge_index = torch.cat([d.edge_index for d in data_list], dim=1)
        edge_index = edge_index + torch.repeat_interleave(ptr[:-1], e).unsqueeze(0)
        out = cls(
            x=torch.cat([d.x for d in data_list], dim=0),
            edge_index=edge_index,
            edge_attr=torch.cat([d.edge_attr for d in data_list], dim=0),
        )
        out.batch = torch.repeat_interleave(torch.arange(len(data_list), dtype=torch.long), n)
        out.ptr = ptr
        out.num_graphs = len(data_list)
        skip = {"x", "edge_index", "edge_attr", "batch", "ptr", "num_graphs"}
        for key in data_list[0].keys():
            if key in skip:
                continue
            vals = [getattr(d, key) for d in data_list]
            if all(isinstance(v, torch.Tensor) for v in vals):
                setattr(out, key, torch.cat(vals, dim=0))
            else:
                setattr(out, key, vals)
        return out

    def to_data_list(self) -> List[Data]:
        out = []
        ptr = self.ptr.tolist()
        src_graph = self.batch[self.edge_index[0]]
        for g in range(self.num_graphs):
            emask = src_graph == g
            d = Data(x=self.x[ptr[g]:ptr[g + 1]], edge_index=self.edge_index[:, emask] - ptr[g],
                     edge_attr=self.edge_attr[emask])
            for key in ("para", "assoc"):
                if hasattr(self, key):
                    setattr(d, key, getattr(self, key)[g:g + 1])
            out.append(d)
        return out


def in_degree(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """int64 in-degree of every node (PyG ``degree(edge_index[1], num_nodes, dtype=long)``)."""
    return torch.bincount(edge_index[1], minlength=num_nodes)[:num_nodes] if num_nodes > 0 else \
        torch.zeros(0, dtype=torch.long)


def calc_deg(dataset: Iterable[Data]) -> List[int]:
    """In-degree histogram over a dataset → ``config["deg"]``.

    Same algorithm as ``/root/reference/gnnepcsaft/train/utils.py:48-60`` (max in-degree, then summed
    ``bincount``); the reference's dataset-loading prologue (lines 39-47) is out of scope, so this takes the
    iterable of graphs directly.  A ``Batch`` may be passed as a one-element dataset: the histogram of a
    disjoint union equals the sum of the per-graph histograms.
    """
    if isinstance(dataset, Data):
        dataset = [dataset]
    dataset = list(dataset)
    max_degree = -1
    degs = []
    for data in dataset:
        d = in_degree(data.edge_index, data.num_nodes)
        degs.append(d)
        max_degree = max(max_degree, int(d.max()) if d.numel() else 0)
    deg = torch.zeros(max_degree + 1, dtype=torch.long)
    for d in degs:
        deg += torch.bincount(d, minlength=deg.numel())
    return deg.tolist()


def shard_by_graph(batch: Batch, world_size: int, rank: int) -> Batch:
    """Contiguous range of graphs for ``rank`` with Σ(nodes+edges) balanced across ranks.

    No data-path collective: every rank computes the same split from the same integer arrays.
    """
    B = batch.num_graphs
    ptr = batch.ptr
    n = ptr[1:] - ptr[:-1]
    e_per_graph = torch.bincount(batch.batch[batch.edge_index[1]], minlength=B)
    w = torch.cumsum(n + e_per_graph, 0)
    total = int(w[-1]) if B else 0
    bounds = [0]
    for r in range(1, world_size):
        # number of leading graphs whose cumulative weight is <= r/world of the total (integer arithmetic)
        g = int(torch.searchsorted(w * world_size, torch.tensor(total * r, dtype=w.dtype), right=True))
        bounds.append(max(bounds[-1], min(B, g)))
    bounds.append(B)
    g0, g1 = bounds[rank], bounds[rank + 1]
    n0, n1 = int(ptr[g0]), int(ptr[g1])
    emask = (batch.edge_index[1] >= n0) & (batch.edge_index[1] < n1)
    out = Batch(x=batch.x[n0:n1], edge_index=batch.edge_index[:, emask] - n0, edge_attr=batch.edge_attr[emask])
    out.batch = batch.batch[n0:n1] - g0
    out.ptr = ptr[g0:g1 + 1] - n0
    out.num_graphs = g1 - g0
    for key in ("para", "assoc"):
        if hasattr(batch, key):
            setattr(out, key, getattr(batch, key)[g0:g1])
    return out
