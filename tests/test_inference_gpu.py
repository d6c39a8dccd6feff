"""Inference engine: BatchNorm folded into the preceding Linear, weight-only tables precomputed.
Against the CPU oracle in eval mode and against the unfolded native model, PNA (with towers) and GINE, batched and
``batch=None`` single-molecule form, ``pred_with_bounds`` clipping, hub-heavy fallback.

Tolerance against the oracle: the fp64 oracle is the arbiter and the bound is max(1e-5, 1.5 x the reference's own fp32
reproducibility on the same inputs) -- the eval-mode forward still contains StdAggregation's hard mask, so the CPU fp32
oracle on permuted-equivalent presentations of the batch is evaluated alongside (tests/parity_util.py explains)."""
import copy

import pytest
import torch

from tests.parity_util import make_models, rel_err

pytestmark = pytest.mark.gpu


def _bound_vs_fp64(oracle, batch, draws: int = 6):
    """(fp64 prediction, max(1e-5, 1.5 x max over fp32 draws of the oracle's distance to it))."""
    from tests.model_cases import permuted_copy
    o64 = copy.deepcopy(oracle).double().eval()
    with torch.no_grad():
        ref64 = o64(batch.x, batch.edge_index, batch.edge_attr, batch.batch)
        worst = rel_err(oracle(batch.x, batch.edge_index, batch.edge_attr, batch.batch), ref64)
        for d in range(1, draws):
            pb, gp, _ = permuted_copy(batch, 77 + d)
            worst = max(worst, rel_err(oracle(pb.x, pb.edge_index, pb.edge_attr, pb.batch), ref64[gp]))
    return ref64, max(1e-5, 1.5 * worst)


def _trained_like(cfg, batch):
    """Oracle / native pair with non-trivial running statistics and BN affine parameters."""
    oracle, native = make_models(cfg)
    torch.manual_seed(3)
    for m in oracle.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.3, 0.3)
    oracle.train()
    for _ in range(2):
        oracle(batch.x, batch.edge_index, batch.edge_attr, batch.batch)
    native.load_state_dict(oracle.state_dict())
    oracle.eval()
    native.eval().to("cuda:0")
    return oracle, native


@pytest.mark.parametrize("kw", [dict(hidden_dim=64, propagation_depth=3),
                                dict(hidden_dim=128, towers=4, propagation_depth=2, pre_layers=1, post_layers=2),
                                dict(conv="GINE", hidden_dim=64, propagation_depth=3, global_pool="mean"),
                                dict(hidden_dim=32, propagation_depth=2, global_pool="max", num_para=2)])
def test_engine_matches_eval_model_and_oracle(gpu_device, kw):
    from gnnepcsaft_amd.data import calc_deg, default_config, synthetic_batch
    from gnnepcsaft_amd.inference import InferenceEngine
    cfg = default_config(2)
    cfg.update(kw)
    batch = synthetic_batch(48, 2)
    cfg["deg"] = calc_deg(batch)
    oracle, native = _trained_like(cfg, batch)
    eng = InferenceEngine(native)
    bd = batch.to("cuda:0")
    with torch.no_grad():
        ref = oracle(batch.x, batch.edge_index, batch.edge_attr, batch.batch)
        nat = native(bd.x, bd.edge_index, bd.edge_attr, bd.batch)
    out = eng(bd.x, bd.edge_index, bd.edge_attr, bd.batch)
    assert out.shape == ref.shape
    assert rel_err(out, nat) <= 1e-5, "folded engine vs unfolded native eval model"
    ref64, bound = _bound_vs_fp64(oracle, batch)
    assert rel_err(out, ref64) <= bound, ("folded engine vs fp64 oracle in eval mode", rel_err(out, ref64), bound)
    assert rel_err(nat, ref64) <= bound, ("native eval model vs fp64 oracle", rel_err(nat, ref64), bound)
    # single molecule, batch=None (demo/utils.py:950), and the clipped form
    one = batch.to_data_list()[5].to("cuda:0")
    with torch.no_grad():
        ref1 = oracle(one.x.cpu(), one.edge_index.cpu(), one.edge_attr.cpu(), None)
        refb = oracle.pred_with_bounds(batch)
    with torch.no_grad():
        o64 = copy.deepcopy(oracle).double()
        ref1_64 = o64(one.x.cpu(), one.edge_index.cpu(), one.edge_attr.cpu(), None)
    assert rel_err(eng(one.x, one.edge_index, one.edge_attr, None), ref1_64) <= max(1e-5, 1.5 * rel_err(ref1, ref1_64))
    assert rel_err(eng.pred_with_bounds(bd), refb) <= bound
    with pytest.raises(ValueError):
        bad = copy.copy(bd)
        bad.x = None
        eng.pred_with_bounds(bad)


def test_engine_large_batch_and_hub_fallback(gpu_device):
    """>= 8192 atoms (large-batch kernels) and a batch with > 64 distinct in-degrees (ungrouped post-layer 0)."""
    from gnnepcsaft_amd.data import Batch, Data, calc_deg, default_config, synthetic_batch
    from gnnepcsaft_amd.inference import InferenceEngine
    cfg = default_config(2)
    cfg.update(hidden_dim=128, propagation_depth=2)
    batch = synthetic_batch(512, 2)
    cfg["deg"] = calc_deg(batch)
    oracle, native = _trained_like(cfg, synthetic_batch(64, 2))
    eng = InferenceEngine(native)
    bd = batch.to("cuda:0")
    with torch.no_grad():
        ref = oracle(batch.x, batch.edge_index, batch.edge_attr, batch.batch)
    ref64, bound = _bound_vs_fp64(oracle, batch, draws=4)
    got = eng(bd.x, bd.edge_index, bd.edge_attr, bd.batch)
    assert rel_err(got, ref64) <= bound, (rel_err(got, ref64), bound)
    # a star graph with 100 leaves: in-degree 100 on the hub -> no degree classes
    n = 101
    src = torch.arange(1, n)
    ei = torch.stack([torch.cat([src, torch.zeros(n - 1, dtype=torch.long)]),
                      torch.cat([torch.zeros(n - 1, dtype=torch.long), src])])
    star = Data(x=batch.x[:n].clone(), edge_index=ei, edge_attr=batch.edge_attr[:2 * (n - 1)].clone(),
                para=torch.ones(1, 3), assoc=torch.ones(1, 2))
    mixed = Batch.from_data_list([star] + batch.to_data_list()[:3])
    with torch.no_grad():
        ref_m = oracle(mixed.x, mixed.edge_index, mixed.edge_attr, mixed.batch)
    md = mixed.to("cuda:0")
    mixed.ptr = torch.tensor([0, n] + [n + 20 * (k + 1) for k in range(3)])
    ref64_m, bound_m = _bound_vs_fp64(oracle, mixed, draws=4)
    got_m = eng(md.x, md.edge_index, md.edge_attr, md.batch)
    assert rel_err(got_m, ref64_m) <= bound_m, (rel_err(got_m, ref64_m), bound_m)


def test_single_molecule_graph_replay(gpu_device):
    """``engine.single``: per-shape HIP graph; equals the eager engine bit for bit, across shapes and repeated calls."""
    from gnnepcsaft_amd import ops
    from gnnepcsaft_amd.data import calc_deg, default_config, synthetic_batch
    from gnnepcsaft_amd.inference import InferenceEngine
    cfg = default_config(2)
    cfg.update(hidden_dim=64, propagation_depth=3)
    batch = synthetic_batch(24, 5)  # 5..80 atoms: several shapes
    cfg["deg"] = calc_deg(batch)
    _, native = _trained_like(cfg, synthetic_batch(32, 2))
    eng = InferenceEngine(native, max_degree=len(cfg["deg"]) - 1)
    mols = [m.to("cuda:0") for m in batch.to_data_list()]
    for rep in range(2):
        for m in mols[:10]:
            eager = eng(m.x, m.edge_index, m.edge_attr, None, validate=False).clone()
            replay = eng.single(m.x, m.edge_index, m.edge_attr).clone()
            assert torch.equal(eager, replay)
    ops.check_range(torch.device("cuda:0"))
