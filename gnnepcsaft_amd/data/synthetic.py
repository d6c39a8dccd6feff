"""Deterministic synthetic molecular graphs for the benchmark / parity configs.

There is no network for the reference's datasets (DVC blobs), so every config runs on random molecule-shaped
graphs whose integer domain is the reference's (``/root/reference/gnnepcsaft/data/ogb_utils.py:8-34`` vocab
sizes; both bond directions adjacent, ``ogb_utils.py:125-129``; empty graphs have ``edge_index[2,0]``,
``ogb_utils.py:137-139``).  ``numpy.random.Generator(PCG64(seed))``, seed = 20260130 + cfg_index.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from .batching import ATOM_FEATURE_DIMS, BOND_FEATURE_DIMS, Batch

BASE_SEED = 20260130
# parameter bounds (/root/reference/gnnepcsaft/train/models.py:167-172)
LOWER_BOUNDS = np.array([1.0, 1.9, 50.0, -1 * math.log10(0.9), math.log10(200.0)])
UPPER_BOUNDS = np.array([25.0, 4.5, 550.0, -1 * math.log10(0.0001), math.log10(5000.0)])


def _grow_fixed(rng: np.random.Generator, G: int, n: int, extra: np.ndarray, cap: int, pref: bool):
    """Vectorised over G graphs of n nodes: random recursive tree + ``extra[g]`` ring-closing bonds.

    Returns (bond_u, bond_v, bond_valid) with shape [G, n-1+max_extra]; bonds in creation order.
    """
    deg = np.zeros((G, n), dtype=np.int64)
    adj = np.zeros((G, n, n), dtype=bool)
    max_extra = int(extra.max()) if G else 0
    nb = n - 1 + max_extra
    bu = np.zeros((G, max(nb, 0)), dtype=np.int64)
    bv = np.zeros((G, max(nb, 0)), dtype=np.int64)
    valid = np.zeros((G, max(nb, 0)), dtype=bool)
    ar = np.arange(G)
    for i in range(1, n):
        w = ((deg[:, :i] + 1.0) ** 2) if pref else np.ones((G, i))
        w = w * (deg[:, :i] < cap)
        # fall back to "any earlier node" if every earlier node is saturated (cannot happen for cap >= 2 trees)
        dead = w.sum(1) == 0
        w[dead] = 1.0
        # weighted choice via exponential race (vectorised, deterministic given rng)
        key = rng.exponential(size=(G, i)) / np.maximum(w, 1e-300)
        key[w == 0] = np.inf
        j = key.argmin(1)
        bu[:, i - 1], bv[:, i - 1], valid[:, i - 1] = j, i, True
        deg[ar, j] += 1
        deg[ar, i] += 1
        adj[ar, j, i] = adj[ar, i, j] = True
    iu = np.triu_indices(n, 1)
    for k in range(max_extra):
        want = extra > k
        ok = (~adj[:, iu[0], iu[1]]) & (deg[:, iu[0]] < cap) & (deg[:, iu[1]] < cap) & want[:, None]
        key = rng.random(size=ok.shape)
        key[~ok] = np.inf
        pick = key.argmin(1)
        has = ok.any(1)
        a, b = iu[0][pick], iu[1][pick]
        col = n - 1 + k
        bu[:, col], bv[:, col], valid[:, col] = a, b, has
        g = ar[has]
        deg[g, a[has]] += 1
        deg[g, b[has]] += 1
        adj[g, a[has], b[has]] = adj[g, b[has], a[has]] = True
    return bu, bv, valid


def synthetic_batch(num_graphs: int, cfg_index: int = 2, *, skewed: Optional[bool] = None, num_para: int = 3,
                    seed: Optional[int] = None, n_atoms: int = 20, n_bonds: int = 20,
                    molecule_like: bool = False) -> Batch:
    """Build one ``Batch`` of ``num_graphs`` synthetic molecules (CPU int64 tensors, PyG collate order).

    cfg 1-4: every graph ``n_atoms`` atoms / ``n_bonds`` undirected bonds (2·n_bonds ``edge_index`` columns),
    degree cap 4.  cfg 5 (``skewed``): n = clip(round(exp(N(ln 18, 0.6))), 5, 80), u = n-1+Poisson(0.08 n),
    cap 12, preferential attachment ∝ (deg+1)².  ``molecule_like`` draws features from a few combinations
    (≈70 % "carbon") so that exactly tied messages occur — used by the tie-handling tests.
    """
    if skewed is None:
        skewed = cfg_index == 5
    rng = np.random.Generator(np.random.PCG64(BASE_SEED + cfg_index if seed is None else seed))
    B = num_graphs
    if skewed:
        n = np.clip(np.rint(np.exp(rng.normal(math.log(18.0), 0.6, size=B))), 5, 80).astype(np.int64)
        extra = rng.poisson(0.08 * n)
        cap, pref = 12, True
    else:
        n = np.full(B, n_atoms, dtype=np.int64)
        extra = np.full(B, max(n_bonds - (n_atoms - 1), 0), dtype=np.int64)
        cap, pref = 4, False
    node_ptr = np.zeros(B + 1, dtype=np.int64)
    np.cumsum(n, out=node_ptr[1:])
    src_parts, dst_parts, gid_parts = [], [], []
    order = np.arange(B)
    for size in np.unique(n):
        sel = order[n == size]
        chunk = max(1, 4_000_000 // int(size * size))
        for c0 in range(0, len(sel), chunk):
            gs = sel[c0:c0 + chunk]
            if size == 1:
                continue
            bu, bv, valid = _grow_fixed(rng, len(gs), int(size), extra[gs], cap, pref)
            gg = np.broadcast_to(gs[:, None], bu.shape)[valid]
            src_parts.append(bu[valid] + node_ptr[gg])
            dst_parts.append(bv[valid] + node_ptr[gg])
            gid_parts.append(gg)
    if src_parts:
        u = np.concatenate(src_parts)
        v = np.concatenate(dst_parts)
        gid = np.concatenate(gid_parts)
        # restore dataset order (graphs concatenated in order; bonds in creation order inside a graph)
        o = np.argsort(gid, kind="stable")
        u, v = u[o], v[o]
    else:
        u = v = np.zeros(0, dtype=np.int64)
    nb = len(u)
    edge_index = np.empty((2, 2 * nb), dtype=np.int64)
    edge_index[0, 0::2], edge_index[1, 0::2] = u, v
    edge_index[0, 1::2], edge_index[1, 1::2] = v, u
    N = int(node_ptr[-1])
    if molecule_like:
        palette = np.array([[5, 0, 4, 5, 3, 0, 2, 0, 0], [5, 0, 3, 5, 2, 0, 2, 0, 0], [7, 0, 2, 5, 0, 0, 2, 0, 0],
                            [6, 0, 3, 5, 1, 0, 2, 0, 0]], dtype=np.int64)
        x = palette[rng.choice(4, size=N, p=[0.7, 0.1, 0.1, 0.1])]
        bond = np.zeros((nb, 3), dtype=np.int64)
        bond[:, 0] = rng.choice(2, size=nb, p=[0.9, 0.1])
    else:
        x = np.stack([rng.integers(0, d, size=N) for d in ATOM_FEATURE_DIMS], axis=1).astype(np.int64)
        bond = np.stack([rng.integers(0, d, size=nb) for d in BOND_FEATURE_DIMS], axis=1).astype(np.int64)
    edge_attr = np.repeat(bond, 2, axis=0)
    lo, hi = (LOWER_BOUNDS[:3], UPPER_BOUNDS[:3]) if num_para == 3 else (LOWER_BOUNDS[3:], UPPER_BOUNDS[3:])
    para = rng.uniform(LOWER_BOUNDS[:3], UPPER_BOUNDS[:3], size=(B, 3)).astype(np.float32)
    assoc = rng.uniform(LOWER_BOUNDS[3:], UPPER_BOUNDS[3:], size=(B, 2)).astype(np.float32)
    out = Batch(x=torch.from_numpy(x), edge_index=torch.from_numpy(edge_index), edge_attr=torch.from_numpy(edge_attr))
    out.batch = torch.from_numpy(np.repeat(np.arange(B, dtype=np.int64), n))
    out.ptr = torch.from_numpy(node_ptr)
    out.num_graphs = B
    out.para = torch.from_numpy(para)
    out.assoc = torch.from_numpy(assoc)
    return out


def default_config(cfg_index: int = 2) -> dict:
    """Model hyper-parameters of BASELINE.json's configs (defaults from
    ``/root/reference/gnnepcsaft/configs/default.py:37-47``: L=6, pre=2, post=4, T=1, pool=add, dropout=0, P=3)."""
    cfg = dict(model="gnn", conv="PNA", global_pool="add", propagation_depth=6, hidden_dim=256, dropout=0.0,
               add_self_loops=True, num_para=3, post_layers=4, pre_layers=2, towers=1, deg=[], num_layers=2,
               num_stacks=2, heads=2, optimizer="adam", learning_rate=1e-3, weight_decay=1e-2, warmup_steps=2,
               dataset="esper", batch_size=512)
    if cfg_index == 1:
        cfg.update(hidden_dim=256, batch_size=32)
    elif cfg_index in (2, 4):
        cfg.update(hidden_dim=128, batch_size=4096 if cfg_index == 2 else 131072)
    elif cfg_index == 3:
        cfg.update(conv="GINE", hidden_dim=256, batch_size=16384)
    elif cfg_index == 5:
        cfg.update(hidden_dim=512, towers=4, batch_size=65536)
    else:
        raise ValueError(f"unknown BASELINE config index {cfg_index}")
    return cfg
