"""Measures how far the reference algorithm's own fp32 evaluation (the CPU oracle) sits from its fp64 evaluation on the
model cases of tests/test_model_gpu.py -> tests/golden/conditioning.json.  CPU only; evidence for the tolerances in
tests/parity_util.py.  Run:  python -m tests.golden.make_conditioning"""
import copy
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from gnnepcsaft_amd.data import calc_deg, default_config, synthetic_batch  # noqa: E402
from oracle import pyg_restatement as O  # noqa: E402
from tests.parity_util import grad_errors, rel_err  # noqa: E402

CASES = {
    "pna_small (H64 L2, 32 graphs)": (dict(hidden_dim=64, propagation_depth=2), 32, 1),
    "pna_cfg1_shape (H256 L6, 32 graphs)": (dict(hidden_dim=256, propagation_depth=6), 32, 1),
    "pna_towers4 (H128 T4 L2, 32 graphs)": (dict(hidden_dim=128, towers=4, propagation_depth=2), 32, 1),
    "pna_cfg2_shape (H128 L6, 256 graphs)": (dict(hidden_dim=128, propagation_depth=6), 256, 2),
    "pna_skewed (H64 T2 L3, 64 graphs, cfg-5 sizes)": (dict(hidden_dim=64, towers=2, propagation_depth=3), 64, 5),
    "gine_small (H64 L3, 32 graphs)": (dict(conv="GINE", hidden_dim=64, propagation_depth=3), 32, 1),
    "gine_h256 (H256 L6, 32 graphs)": (dict(conv="GINE", hidden_dim=256, propagation_depth=6), 32, 1),
}


def main():
    out = {}
    for name, (kw, graphs, gen) in CASES.items():
        cfg = default_config(2)
        cfg.update(kw)
        b = synthetic_batch(graphs, gen)
        cfg["deg"] = calc_deg(b)
        torch.manual_seed(0)
        m32 = O.GNNePCSAFT(cfg).train()
        m64 = copy.deepcopy(m32).double()
        p32 = m32(b.x, b.edge_index, b.edge_attr, b.batch)
        p64 = m64(b.x, b.edge_index, b.edge_attr, b.batch)
        l32, l64 = O.ape_huber_loss(p32, b.para), O.ape_huber_loss(p64, b.para.double())
        l32.backward()
        l64.backward()
        g32 = {n: p.grad.double() for n, p in m32.named_parameters()}
        g64 = {n: p.grad for n, p in m64.named_parameters()}
        ge = grad_errors(g32, g64)
        out[name] = {"pred_rel": rel_err(p32, p64), "loss_rel": rel_err(l32, l64), "grad_rel_l2": ge["l2"],
                     "grad_rel_max": ge["max"], "grad_rel_argmax": ge["argmax"]}
        print(name, out[name], flush=True)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conditioning.json")
    json.dump({"what": "CPU oracle fp32 vs the same oracle in fp64 (norm-wise relative errors), torch threads = "
                       f"{torch.get_num_threads()}", "cases": out}, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
