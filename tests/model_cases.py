"""Registry of the whole-model parity cases: name -> (config overrides, batch recipe).  Shared by
tests/test_model_gpu.py (the GPU parity tests), tests/golden/make_conditioning.py (the reference's own fp32
reproducibility envelope per case, CPU) and tools/model_parity_survey.py.

Coverage of BASELINE.json's configs: configs[0] ``pna_cfg1_shape`` (H=256, L=6, batch 32, the case as written);
configs[1] ``pna_cfg2_full_1024`` (H=128, L=6, every large-batch kernel active) and ``pna_cfg2_shape_256``;
configs[2] ``gine_cfg3_full_1024`` (H=256, L=6) and ``gine_large_l2``; configs[4] ``pna_cfg5_shaped`` (H=512, T=4,
skewed 5..80 atoms, >= 8192 atoms, L=2) and ``pna_skewed``; configs[3] is configs[1] on 8 GPUs (tests/test_dp_*).
"""
from __future__ import annotations

import copy
from typing import Dict, Tuple


def _lone():
    import torch
    from gnnepcsaft_amd.data import Data
    return Data(x=torch.tensor([[5, 0, 4, 5, 3, 0, 2, 0, 0]]), edge_index=torch.empty(2, 0, dtype=torch.long),
                edge_attr=torch.empty(0, 3, dtype=torch.long), para=torch.tensor([[2.0, 3.0, 200.0]]),
                assoc=torch.tensor([[1.0, 3.0]]))


# name: (cfg overrides, recipe).  recipe: graphs, gen (synthetic generator config index), optional seed /
# molecule_like / special ("lone": single-atom graphs interleaved, "hubs": star graphs with > 64 in-degrees,
# "esper": real molecules from tests/golden/esper_smiles_sample.tsv through gnnepcsaft_amd.data.featurize)
MODEL_CASES: Dict[str, Tuple[dict, dict]] = {
    # --- the 32-graph shape sweep
    "pna_small": (dict(hidden_dim=64, propagation_depth=2), dict(graphs=32, gen=1)),
    "pna_cfg1_shape": (dict(hidden_dim=256, propagation_depth=6), dict(graphs=32, gen=1)),
    "pna_towers4": (dict(hidden_dim=128, towers=4, propagation_depth=2), dict(graphs=32, gen=1)),
    "pna_pre1_post1": (dict(hidden_dim=32, pre_layers=1, post_layers=1, propagation_depth=2), dict(graphs=32, gen=1)),
    "pna_pre3_post2_mean": (dict(hidden_dim=48, pre_layers=3, post_layers=2, propagation_depth=2, global_pool="mean"),
                            dict(graphs=32, gen=1)),
    "pna_max_pool_assoc": (dict(hidden_dim=32, propagation_depth=2, global_pool="max", num_para=2),
                           dict(graphs=32, gen=1)),
    "gine_small": (dict(conv="GINE", hidden_dim=64, propagation_depth=3), dict(graphs=32, gen=1)),
    "gine_h256": (dict(conv="GINE", hidden_dim=256, propagation_depth=6), dict(graphs=32, gen=1)),
    # --- seeded small batches (other depths / towers / pools / tie-rich features)
    "pna_h64_l2_s": (dict(hidden_dim=64, propagation_depth=2), dict(graphs=32, gen=2, seed=7047)),
    "pna_h128_l3_s": (dict(hidden_dim=128, propagation_depth=3), dict(graphs=12, gen=2, seed=7007)),
    "pna_t2_h32_l3_skewed_s": (dict(hidden_dim=32, towers=2, propagation_depth=3), dict(graphs=32, gen=5, seed=7002)),
    "pna_t4_h64_l2_mean_s": (dict(hidden_dim=64, towers=4, propagation_depth=2, global_pool="mean"),
                             dict(graphs=24, gen=2, seed=7242)),
    "pna_pre1_post1_max_assoc_s": (dict(hidden_dim=32, pre_layers=1, post_layers=1, propagation_depth=2,
                                        global_pool="max", num_para=2), dict(graphs=32, gen=2, seed=7013)),
    "pna_ties_h32_l2_s": (dict(hidden_dim=32, propagation_depth=2), dict(graphs=32, gen=2, seed=7034, molecule_like=True)),
    "pna_h64_l6_s": (dict(hidden_dim=64, propagation_depth=6), dict(graphs=8, gen=2, seed=7004)),
    "gine_h64_l3_s": (dict(conv="GINE", hidden_dim=64, propagation_depth=3), dict(graphs=32, gen=3, seed=7001)),
    "gine_h256_l6_s": (dict(conv="GINE", hidden_dim=256, propagation_depth=6), dict(graphs=32, gen=3, seed=7007)),
    # --- skewed degrees, exact ties, degenerate graphs
    "pna_skewed": (dict(hidden_dim=64, towers=2, propagation_depth=3), dict(graphs=64, gen=5)),
    "pna_ties": (dict(hidden_dim=64, towers=2, propagation_depth=3), dict(graphs=64, gen=2, molecule_like=True)),
    "pna_lone_atoms": (dict(hidden_dim=32, propagation_depth=2), dict(special="lone")),
    "gine_lone_atoms": (dict(conv="GINE", hidden_dim=32, propagation_depth=2), dict(special="lone")),
    "pna_hubs": (dict(hidden_dim=64, propagation_depth=2), dict(special="hubs")),
    # --- real molecules: 94 rows of the reference's Esper table through the SMILES featuriser (1..40 heavy atoms,
    #     aromatic rings, single-atom molecules, labels = the table's m / sigma / epsilon)
    "pna_esper_molecules": (dict(hidden_dim=64, propagation_depth=3), dict(special="esper")),
    "gine_esper_molecules": (dict(conv="GINE", hidden_dim=64, propagation_depth=3), dict(special="esper")),
    # --- batches that select the large-batch kernels (>= 4096 / 8192 rows)
    "pna_cfg2_shape_256": (dict(hidden_dim=128, propagation_depth=6), dict(graphs=256, gen=2)),
    "pna_large_l2": (dict(hidden_dim=128, propagation_depth=2), dict(graphs=640, gen=2)),
    "gine_large_l2": (dict(conv="GINE", hidden_dim=256, propagation_depth=2), dict(graphs=512, gen=3)),
    "pna_cfg2_full_1024": (dict(hidden_dim=128, propagation_depth=6), dict(graphs=1024, gen=2)),
    "gine_cfg3_full_1024": (dict(conv="GINE", hidden_dim=256, propagation_depth=6), dict(graphs=1024, gen=3)),
    "pna_cfg5_shaped": (dict(hidden_dim=512, towers=4, propagation_depth=2), dict(graphs=448, gen=5)),
    # BASELINE configs[4] at its full depth (H = 512, T = 4, L = 6) on 448 skewed graphs: the yardstick of the full-size
    # train-mode property test (tests/test_full_size_gpu.py)
    "pna_cfg5_l6_shaped": (dict(hidden_dim=512, towers=4, propagation_depth=6), dict(graphs=448, gen=5)),
}


def build_case(name: str):
    """(cfg with ``deg`` filled in, Batch on the CPU, name of the label field)."""
    from gnnepcsaft_amd.data import Batch, calc_deg, default_config, synthetic_batch
    kw, recipe = MODEL_CASES[name]
    cfg = default_config(2)
    cfg.update(copy.deepcopy(kw))
    special = recipe.get("special")
    if special == "lone":
        base = synthetic_batch(6, 2).to_data_list()
        lone = _lone()
        batch = Batch.from_data_list([lone, base[0], lone, base[1], base[2], lone])
    elif special == "esper":
        import os
        import torch
        from gnnepcsaft_amd.data.featurize import from_smiles
        rows = [ln.rstrip("\n").split("\t") for ln in open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                                                       "esper_smiles_sample.tsv")) if not ln.startswith("#")]
        batch = Batch.from_data_list([from_smiles(s, para=torch.tensor([[float(m), float(sg), float(ep)]]),
                                                  assoc=torch.tensor([[1.0, 3.0]])) for s, m, sg, ep in rows])
    elif special == "hubs":
        from tests.conv_cases import hub_batch
        batch = hub_batch(graphs=24)
    else:
        batch = synthetic_batch(recipe["graphs"], recipe["gen"], seed=recipe.get("seed"),
                                molecule_like=recipe.get("molecule_like", False))
    cfg["deg"] = calc_deg(batch)
    return cfg, batch, ("assoc" if cfg["num_para"] == 2 else "para")


def permuted_copy(batch, seed: int):
    """An EQUIVALENT presentation of the same batch: graphs in another order, nodes relabelled inside every graph,
    edge columns shuffled.  Every quantity the model defines is invariant (predictions up to the graph order returned
    as ``graph_perm``: row g of the new prediction belongs to graph ``graph_perm[g]`` of the original; node row i of a new
    node-level tensor is node ``old_of_new[i]`` of the original), but every fp32
    summation (scatter order, BatchNorm statistics, GEMM row blocks) runs in another order -- a fresh draw of the
    reference's own fp32 rounding."""
    import numpy as np
    import torch
    from gnnepcsaft_amd.data import Batch
    rng = np.random.Generator(np.random.PCG64(seed))
    B = int(batch.num_graphs)
    ptr = batch.ptr.numpy()
    graph_perm = rng.permutation(B)
    new_of_old = np.empty(int(ptr[-1]), dtype=np.int64)
    pieces, pos = [], 0
    for g in graph_perm:
        n = int(ptr[g + 1] - ptr[g])
        local = rng.permutation(n)                      # new local position -> old local node
        pieces.append(ptr[g] + local)
        new_of_old[ptr[g] + local] = pos + np.arange(n)
        pos += n
    old_of_new = np.concatenate(pieces) if pieces else np.zeros(0, dtype=np.int64)
    ecol = rng.permutation(batch.edge_index.size(1))
    ei = torch.from_numpy(new_of_old[batch.edge_index.numpy()[:, ecol]])
    out = Batch(x=batch.x[torch.from_numpy(old_of_new)], edge_index=ei, edge_attr=batch.edge_attr[torch.from_numpy(ecol)])
    sizes = (ptr[1:] - ptr[:-1])[graph_perm]
    out.batch = torch.from_numpy(np.repeat(np.arange(B, dtype=np.int64), sizes))
    out.ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64))
    out.num_graphs = B
    gp = torch.from_numpy(graph_perm)
    for key in ("para", "assoc"):
        if hasattr(batch, key):
            setattr(out, key, getattr(batch, key)[gp])
    return out, gp, torch.from_numpy(old_of_new)
