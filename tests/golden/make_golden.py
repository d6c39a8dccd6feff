"""Generates the committed golden vectors ``tests/golden/*.npz``.

The reference itself cannot produce vectors here (torch_geometric / ogb / lightning are not importable in this image
and the reference ships no fixtures: SURVEY.md §8c) — PARITY UNPINNED.  These vectors are therefore outputs of the
repository's own oracle (``oracle/pyg_restatement.py``) evaluated in **fp64** on seeded synthetic molecules, i.e. the
"exact" answer of the restated algorithm, reproducible across machines to ~1e-12.  They pin (a) the oracle against
silent drift and (b) the HIP path against a value that does not depend on any fp32 summation order.

Run:  python -m tests.golden.make_golden
"""
from __future__ import annotations

import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CASES = {
    "pna_h32_l2_t2": dict(cfg=dict(conv="PNA", hidden_dim=32, propagation_depth=2, towers=2), graphs=12, gen=2, seed=11),
    "gine_h32_l2": dict(cfg=dict(conv="GINE", hidden_dim=32, propagation_depth=2), graphs=12, gen=2, seed=12),
}
GRAD_KEYS = ["mlp.6.weight", "mlp.0.weight", "convs.0.lin.weight", "convs.1.lin.weight",
             "node_embed.atom_embedding_list.0.weight", "edge_embed.bond_embedding_list.0.weight",
             "batch_norms.0.module.weight"]


def build(case):
    from gnnepcsaft_amd.data import calc_deg, default_config, synthetic_batch
    from oracle import pyg_restatement as O
    cfg = default_config(2)
    cfg.update(case["cfg"])
    batch = synthetic_batch(case["graphs"], case["gen"], seed=20260130 + case["seed"])
    cfg["deg"] = calc_deg(batch)
    torch.manual_seed(case["seed"])
    model = O.GNNePCSAFT(cfg)
    return cfg, batch, model


def run_case(case, dtype=torch.float64):
    from oracle import pyg_restatement as O
    cfg, batch, model = build(case)
    model = copy.deepcopy(model).to(dtype)
    model.train()
    pred = model(batch.x, batch.edge_index, batch.edge_attr, batch.batch)
    loss = O.ape_huber_loss(pred, batch.para.to(dtype))
    loss.backward()
    out = {"pred": pred.detach().numpy().astype(np.float64), "loss": np.array(float(loss))}
    params = dict(model.named_parameters())
    for k in GRAD_KEYS:
        if k in params:
            out["grad." + k] = params[k].grad.numpy().astype(np.float64)
    out["running_mean.0"] = model.batch_norms[0].module.running_mean.numpy().astype(np.float64)
    return out


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    for name, case in CASES.items():
        out = run_case(case)
        np.savez_compressed(os.path.join(here, f"{name}.npz"), **out)
        print(name, {k: v.shape for k, v in out.items()}, "loss", float(out["loss"]))


if __name__ == "__main__":
    main()
