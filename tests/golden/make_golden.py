"""Generates the committed golden fixtures ``tests/golden/*.npz``.

The reference itself cannot produce vectors here (torch_geometric / ogb / lightning are not importable in this image
and the reference ships no fixtures: SURVEY.md §8c) — PARITY UNPINNED.  The fixtures are therefore outputs of the
repository's own oracle (``oracle/pyg_restatement.py``) evaluated in **fp64** on seeded synthetic molecules, i.e. the
"exact" answer of the restated algorithm, reproducible across machines to ~1e-12.  Each file is self-contained:

  in.x / in.edge_index / in.edge_attr / in.batch / in.para      the batch (int64 / fp32), PyG collate layout
  w.<state-dict key>                                            every parameter and buffer BEFORE the step (fp32)
  inter.embed / inter.conv{l} / inter.act{l} / inter.pool       per-layer forward intermediates
  dinter.d_embed / d_conv{l} / d_pool                           d loss / d intermediate
  pred, loss, grad.<parameter>                                  outputs and ALL parameter gradients
  after.<buffer>                                                BatchNorm running statistics after the step
                                                                (all computed in fp64; pred / loss stored as fp64, the
                                                                bulky arrays rounded to fp32 for storage: 6e-8 relative,
                                                                far inside the 1e-5 they are compared at)
  env.<metric>                                                  the reference's fp32 reproducibility envelope on this
                                                                batch (tests/golden/make_conditioning.py)

They pin (a) the oracle against silent drift (tests/test_host_cpu.py) and (b) the HIP path, layer by layer, against
values that do not depend on any fp32 summation order (tests/test_model_gpu.py).

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

GOLDEN = {
    "pna_h32_l2_t2": dict(cfg=dict(conv="PNA", hidden_dim=32, propagation_depth=2, towers=2), graphs=8, gen=2, seed=11),
    "gine_h32_l2": dict(cfg=dict(conv="GINE", hidden_dim=32, propagation_depth=2), graphs=8, gen=2, seed=12),
}


def build(case):
    """(cfg, batch, freshly initialised fp32 oracle model) from the recipe -- used to CREATE a fixture."""
    from gnnepcsaft_amd.data import calc_deg, default_config, synthetic_batch
    from oracle import pyg_restatement as O
    cfg = default_config(2)
    cfg.update(case["cfg"])
    batch = synthetic_batch(case["graphs"], case["gen"], seed=20260130 + case["seed"])
    cfg["deg"] = calc_deg(batch)
    torch.manual_seed(case["seed"])
    model = O.GNNePCSAFT(cfg)
    return cfg, batch, model


def load_fixture(name):
    """(cfg, batch, state dict of fp32 tensors, the npz) from a committed file -- nothing is regenerated from seeds."""
    from gnnepcsaft_amd.data import Batch, default_config
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"{name}.npz"))
    cfg = default_config(2)
    cfg.update(GOLDEN[name]["cfg"])
    cfg["deg"] = [int(v) for v in gold["in.deg"]]
    batch = Batch(x=torch.from_numpy(gold["in.x"]), edge_index=torch.from_numpy(gold["in.edge_index"]),
                  edge_attr=torch.from_numpy(gold["in.edge_attr"]))
    batch.batch = torch.from_numpy(gold["in.batch"])
    batch.ptr = torch.from_numpy(gold["in.ptr"])
    batch.num_graphs = int(batch.ptr.numel() - 1)
    batch.para = torch.from_numpy(gold["in.para"])
    state = {k[2:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("w.")}
    return cfg, batch, state, gold


def evaluate(cfg, batch, state, dtype=torch.float64):
    """The oracle on (batch, weights) in ``dtype``: every array a fixture holds, as numpy fp64."""
    from oracle import pyg_restatement as O
    from tests.parity_util import capture_intermediates
    model = O.GNNePCSAFT(cfg)
    model.load_state_dict(state, strict=True)
    model = model.to(dtype).train()
    cap = capture_intermediates(model)
    with cap as inter:
        pred = model(batch.x, batch.edge_index, batch.edge_attr, batch.batch)
    loss = O.ape_huber_loss(pred, batch.para.to(dtype))
    loss.backward()
    out = {"pred": pred.detach().double().numpy(), "loss": np.array(float(loss))}
    for k, v in inter.items():
        out["inter." + k] = v.numpy()
    for k, v in cap.grads().items():
        out["dinter." + k] = v.numpy()
    for n, p in model.named_parameters():
        out["grad." + n] = p.grad.detach().double().numpy()
    for n, b in model.named_buffers():
        if b.dtype.is_floating_point and "running" in n:
            out["after." + n] = b.detach().double().numpy()
    return out


def main():
    from tests.golden.make_conditioning import distances, jitter_weights, run_model
    from tests.model_cases import permuted_copy
    from oracle import pyg_restatement as O
    here = os.path.dirname(os.path.abspath(__file__))
    for name, case in GOLDEN.items():
        cfg, batch, model = build(case)
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        out = evaluate(cfg, batch, state)
        out.update({"in.x": batch.x.numpy(), "in.edge_index": batch.edge_index.numpy(),
                    "in.edge_attr": batch.edge_attr.numpy(), "in.batch": batch.batch.numpy(), "in.ptr": batch.ptr.numpy(),
                    "in.para": batch.para.numpy(), "in.deg": np.array(cfg["deg"], dtype=np.int64)})
        for k, v in state.items():
            out["w." + k] = v.numpy()
        # the reference's fp32 reproducibility envelope on this very batch and these very weights
        m64 = O.GNNePCSAFT(cfg)
        m64.load_state_dict(state)
        ref = run_model(m64.double().train(), batch, "para", torch.float64)
        env = {}
        for d in range(64):
            m = O.GNNePCSAFT(cfg).train()
            m.load_state_dict(state)
            if d >= 32:
                jitter_weights(m, seed=d)
            if d == 0:
                dist = distances(run_model(m, batch, "para", torch.float32), ref)
            else:
                pb, gp, nm = permuted_copy(batch, 1000 * d + 17)
                dist = distances(run_model(m, pb, "para", torch.float32), ref, gp, nm)
            for k, v in dist.items():
                env[k] = max(env.get(k, 0.0), v)
        for k, v in env.items():
            out["env." + k] = np.array(v)
        for k in list(out):
            if k.split(".")[0] in ("inter", "dinter", "grad", "after"):
                out[k] = out[k].astype(np.float32)
        np.savez_compressed(os.path.join(here, f"{name}.npz"), **out)
        size = os.path.getsize(os.path.join(here, f"{name}.npz"))
        print(name, f"{size / 1024:.0f} KiB", "loss", float(out["loss"]), "envelope", {k: f"{v:.1e}" for k, v in env.items()})


if __name__ == "__main__":
    main()
