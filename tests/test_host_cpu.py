"""CPU tests of the host logic (no GPU, no compute calls): integer batching vs the oracle (bit-exact), synthetic
generator invariants, C-ABI library loads and exports every symbol include/gnx.h declares, the model mirror's
construction / state-dict / error surface, and that the product path refuses to run without a HIP device."""
import os
import re
import sys

import numpy as np
import pytest
import torch

from oracle import pyg_restatement as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------------------------------
# batching: integer work, bit-exact against the oracle's collate restatement
# ---------------------------------------------------------------------------------------------------------------
def _data_lists(seed=0, sizes=(1, 5, 3, 1, 8)):
    from gnnepcsaft_amd.data import Data
    rng = np.random.default_rng(seed)
    ours, theirs = [], []
    for n in sizes:
        e = 0 if n == 1 else int(rng.integers(1, 3 * n))
        ei = torch.from_numpy(rng.integers(0, n, size=(2, e))).long()
        x = torch.from_numpy(np.stack([rng.integers(0, d, size=n) for d in O.ATOM_FEATURE_DIMS], 1)).long()
        ea = torch.from_numpy(np.stack([rng.integers(0, d, size=e) for d in O.BOND_FEATURE_DIMS], 1)).long().reshape(e, 3)
        para = torch.rand(1, 3)
        ours.append(Data(x=x, edge_index=ei, edge_attr=ea, para=para))
        theirs.append(O.Data(x=x, edge_index=ei, edge_attr=ea, para=para))
    return ours, theirs


def test_collate_bit_exact_with_oracle():
    from gnnepcsaft_amd.data import Batch
    ours, theirs = _data_lists()
    a, b = Batch.from_data_list(ours), O.collate(theirs)
    for key in ("x", "edge_index", "edge_attr", "batch", "ptr", "para"):
        ta, tb = getattr(a, key), getattr(b, key)
        assert ta.dtype == tb.dtype and torch.equal(ta, tb), key
    assert a.num_graphs == b.num_graphs == 5
    back = a.to_data_list()
    for d0, d1 in zip(ours, back):
        assert torch.equal(d0.x, d1.x) and torch.equal(d0.edge_index, d1.edge_index)
        assert torch.equal(d0.edge_attr, d1.edge_attr)


def test_calc_deg_bit_exact_with_oracle():
    from gnnepcsaft_amd.data import Batch, calc_deg
    ours, theirs = _data_lists(seed=3, sizes=(4, 9, 1, 6))
    assert calc_deg(ours) == O.calc_deg(theirs)
    assert calc_deg(Batch.from_data_list(ours)) == O.calc_deg(theirs)  # histogram of a disjoint union


def test_shard_by_graph_partitions_the_batch():
    from gnnepcsaft_amd.data import shard_by_graph, synthetic_batch
    b = synthetic_batch(37, 5)
    for world in (1, 2, 3, 8):
        shards = [shard_by_graph(b, world, r) for r in range(world)]
        assert sum(s.num_graphs for s in shards) == 37
        assert torch.equal(torch.cat([s.x for s in shards]), b.x)
        assert torch.equal(torch.cat([s.para for s in shards]), b.para)
        off, e_tot = 0, 0
        for s in shards:
            assert int(s.ptr[-1]) == s.x.size(0) and (s.num_edges == 0 or int(s.edge_index.max()) < s.x.size(0))
            e_tot += s.num_edges
            off += s.x.size(0)
        assert e_tot == b.num_edges
        if world > 1:  # balanced by nodes + edges
            w = [s.x.size(0) + s.num_edges for s in shards]
            assert max(w) - min(w) <= 2 * 250


def test_synthetic_generator_invariants():
    from gnnepcsaft_amd.data import synthetic_batch
    a, b = synthetic_batch(50, 2), synthetic_batch(50, 2)
    assert torch.equal(a.edge_index, b.edge_index) and torch.equal(a.x, b.x) and torch.equal(a.para, b.para)
    assert a.x.shape == (1000, 9) and a.edge_index.shape == (2, 2000) and a.edge_attr.shape == (2000, 3)
    ei = a.edge_index
    assert torch.equal(ei[:, 0::2], ei[:, 1::2].flip(0)), "both directions of a bond are adjacent columns"
    assert torch.equal(a.edge_attr[0::2], a.edge_attr[1::2])
    deg = torch.bincount(ei[1], minlength=1000)
    assert int(deg.max()) <= 4 and int(deg.min()) >= 1
    assert torch.equal(a.batch[ei[0]], a.batch[ei[1]]), "edges stay inside their graph"
    for k, d in enumerate(O.ATOM_FEATURE_DIMS):
        assert 0 <= int(a.x[:, k].min()) and int(a.x[:, k].max()) < d
    s = synthetic_batch(200, 5)
    n = s.ptr[1:] - s.ptr[:-1]
    assert int(n.min()) >= 5 and int(n.max()) <= 80 and int(torch.bincount(s.edge_index[1]).max()) <= 12
    lo, hi = torch.tensor([1.0, 1.9, 50.0]), torch.tensor([25.0, 4.5, 550.0])
    assert bool(((a.para >= lo) & (a.para <= hi)).all())


# ---------------------------------------------------------------------------------------------------------------
# C ABI: the library loads and exports exactly what include/gnx.h declares
# ---------------------------------------------------------------------------------------------------------------
def test_abi_library_exports_every_declared_symbol():
    from gnnepcsaft_amd import _lib
    header = open(os.path.join(ROOT, "include", "gnx.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(gnx_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    lib = _lib.load()  # binds every symbol; AttributeError if one is missing
    assert lib.gnx_abi_version() == _lib.ABI_VERSION
    assert lib.gnx_pack_csr_workspace_bytes(100, 300) >= 4 * (2 * 300 + 100)
    assert lib.gnx_batchnorm_workspace_bytes(1000, 128) >= 4 * 128 * 2 * 4


def test_product_path_fails_loudly_without_gpu():
    from gnnepcsaft_amd import ops
    from gnnepcsaft_amd._lib import GnxError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(GnxError):
        ops.pack_graph(torch.zeros(2, 0, dtype=torch.long), None, None, 1)
    with pytest.raises(GnxError):
        ops.gemm([(torch.zeros(2, 4), None, torch.zeros(3, 4))], torch.zeros(2, 3))


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "gnnepcsaft_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


# ---------------------------------------------------------------------------------------------------------------
# model mirror: names, state dict, error behaviour (reference models.py:441-606)
# ---------------------------------------------------------------------------------------------------------------
def _cfg(**kw):
    from gnnepcsaft_amd.data import default_config
    cfg = default_config(2)
    cfg.update(hidden_dim=16, propagation_depth=2, deg=[0, 3, 2, 1])
    cfg.update(kw)
    return cfg


def test_state_dict_keys_match_upstream_naming():
    from gnnepcsaft_amd.train.models import GNNePCSAFTL
    sd = GNNePCSAFTL(_cfg()).state_dict()
    expect = ["model.node_embed.atom_embedding_list.8.weight", "model.edge_embed.bond_embedding_list.2.weight",
              "model.convs.0.aggr_module.avg_deg_lin", "model.convs.0.aggr_module.avg_deg_log",
              "model.convs.1.edge_encoder.weight", "model.convs.1.edge_encoder.bias",
              "model.convs.0.pre_nns.0.0.weight", "model.convs.0.pre_nns.0.2.bias",
              "model.convs.0.post_nns.0.0.weight", "model.convs.0.post_nns.0.6.bias", "model.convs.0.lin.weight",
              "model.batch_norms.1.module.running_var", "model.batch_norms.1.module.num_batches_tracked",
              "model.mlp.0.weight", "model.mlp.1.running_mean", "model.mlp.3.bias", "model.mlp.4.weight",
              "model.mlp.6.weight"]
    for k in expect:
        assert k in sd, k
    assert sd["model.convs.0.post_nns.0.0.weight"].shape == (16, 13 * 16)
    assert sd["model.convs.0.pre_nns.0.0.weight"].shape == (16, 3 * 16)
    assert "model.lower_bounds" not in sd
    sd_g = GNNePCSAFTL(_cfg(conv="GINE")).state_dict()
    for k in ("model.convs.0.nn.0.weight", "model.convs.0.nn.2.bias", "model.convs.0.lin.weight", "model.convs.0.eps"):
        assert k in sd_g, k
    # interchangeable with the oracle's (= upstream's) state dict, both ways
    torch.manual_seed(0)
    o = O.GNNePCSAFT(_cfg(towers=2))
    from gnnepcsaft_amd.train.models import GNNePCSAFT
    n = GNNePCSAFT(_cfg(towers=2))
    n.load_state_dict(o.state_dict(), strict=True)
    o.load_state_dict(n.state_dict(), strict=True)


def test_factories_and_errors():
    from gnnepcsaft_amd.train import models as M
    cfg = _cfg()
    m = M.create_model(cfg, [0, 5, 4])
    assert cfg["deg"] == [0, 5, 4] and isinstance(m, M.GNNePCSAFTL) and m.hparams["config"] is cfg
    assert len(m.model.convs) == 2 and m.model.num_para == 3 and m.model.global_pool_type == "add"
    assert torch.equal(m.model.lower_bounds[:3], torch.tensor([1.0, 1.9, 50.0]))
    assert torch.equal(m.model.upper_bounds[:3], torch.tensor([25.0, 4.5, 550.0]))
    with pytest.raises(ValueError, match="Unsupported model"):
        M.create_model(_cfg(model="foo"), [1])
    with pytest.raises(ValueError, match="Unsupported convolution"):
        M.get_conv(_cfg(conv="nope"))
    with pytest.raises(NotImplementedError):
        M.get_conv(_cfg(conv="GATv2"))
    with pytest.raises(ValueError, match="Unsupported global pooling"):
        M.get_global_pool(_cfg(global_pool="sum"))
    with pytest.raises(ValueError, match="Unsupported optimizer"):
        M.GNNePCSAFTL(_cfg(optimizer="lion")).configure_optimizers()
    oc = M.GNNePCSAFTL(_cfg(optimizer="sgd")).configure_optimizers()
    assert isinstance(oc["optimizer"], torch.optim.SGD) and oc["lr_scheduler"]["frequency"] == 10
    sched = M.GNNePCSAFTL(_cfg()).configure_optimizers()["lr_scheduler"]["scheduler"]
    assert sched.T_0 == 2 and sched.T_mult == 2 and sched.eta_min == 1e-6


def test_avg_deg_buffers_match_oracle():
    from gnnepcsaft_amd import nn as gnn
    deg = torch.tensor([0, 34432, 23864, 12816, 10808])
    a = gnn.DegreeScalerAggregation(deg)
    o = O.DegreeScalerAggregation(["mean"], ["identity"], deg)
    assert torch.equal(a.avg_deg_log, o.avg_deg_log) and torch.equal(a.avg_deg_lin, o.avg_deg_lin)
    assert a.avg_log() == float(o.avg_deg_log)


# ---------------------------------------------------------------------------------------------------------------
# golden fixtures: the oracle must keep reproducing the committed vectors (guards the checker against drift)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["pna_h32_l2_t2", "gine_h32_l2"])
def test_oracle_reproduces_golden(name):
    """The committed fixture is (a) what its recipe generates (inputs and weights bit for bit: guards the synthetic
    generator and the initialisers) and (b) what the oracle computes from the stored inputs and weights in fp64 (to the
    fp32 rounding the bulky arrays are stored with); the oracle in fp32 stays inside the stored envelope."""
    from tests.golden.make_golden import GOLDEN, build, evaluate, load_fixture
    cfg, batch, state, gold = load_fixture(name)
    cfg2, batch2, model2 = build(GOLDEN[name])
    assert cfg2["deg"] == cfg["deg"]
    for f in ("x", "edge_index", "edge_attr", "batch", "para"):
        assert torch.equal(getattr(batch, f), getattr(batch2, f)), f
    for k, v in model2.state_dict().items():
        assert torch.equal(v, state[k]), k
    got = evaluate(cfg, batch, state)  # fp64: machine-independent to ~1e-12
    for k in got:
        scale = max(float(np.abs(gold[k]).max()), 1e-30)
        tol = 1e-9 if gold[k].dtype == np.float64 else 2.0 ** -23
        assert float(np.abs(got[k] - gold[k]).max()) <= tol * scale, k
    got32 = evaluate(cfg, batch, state, dtype=torch.float32)  # the reference's working precision
    assert abs(float(got32["loss"]) - float(gold["loss"])) <= 1e-5 * abs(float(gold["loss"]))
    pred_err = float(np.abs(got32["pred"] - gold["pred"]).max()) / float(np.abs(gold["pred"]).max())
    assert pred_err <= float(gold["env.pred"]) * 1.0001


def test_reference_envelope_covers_every_model_case():
    """tests/golden/conditioning.json holds an envelope for every case of tests/model_cases.py (the GPU tests read it)."""
    import json
    from tests.model_cases import MODEL_CASES, build_case, permuted_copy
    env = json.load(open(os.path.join(ROOT, "tests", "golden", "conditioning.json")))["cases"]
    assert set(env) == set(MODEL_CASES), set(env) ^ set(MODEL_CASES)
    for name, e in env.items():
        assert e["draws"] >= 32 and set(e["max"]) == {"pred", "loss", "grad_l2", "grad_max", "inter", "dinter"}, name
    # permuted_copy is an equivalent presentation: the fp64 oracle cannot tell the difference
    cfg, batch, target = build_case("pna_lone_atoms")
    pb, gp, nm = permuted_copy(batch, 5)
    assert torch.equal(pb.x, batch.x[nm]) and torch.equal(pb.para, batch.para[gp])
    torch.manual_seed(0)
    m = O.GNNePCSAFT(cfg).double().train()
    p0 = m(batch.x, batch.edge_index, batch.edge_attr, batch.batch)
    p1 = m(pb.x, pb.edge_index, pb.edge_attr, pb.batch)
    assert float((p1 - p0[gp]).abs().max()) <= 1e-12 * float(p0.abs().max())


def test_lightning_style_checkpoint_with_configdict_is_read_without_executing_it(tmp_path):
    """a Lightning ``.ckpt`` carries ``hyper_parameters`` as an ``ml_collections.ConfigDict`` (not
    installed).  A file pickled against a throw-away module of that import path must load through the safe loader once
    the module is gone again, with the hyper-parameters back as plain dicts and the tensors intact."""
    import sys
    import types
    from gnnepcsaft_amd.train.trainer import read_checkpoint

    names = ["ml_collections", "ml_collections.config_dict", "ml_collections.config_dict.config_dict"]
    assert not any(n in sys.modules for n in names), "ml_collections unexpectedly importable: adapt this test"
    mods = {n: types.ModuleType(n) for n in names}

    class ConfigDict:  # default object pickling (NEWOBJ + BUILD), entries under _fields like the real class
        def __init__(self, d):
            self._fields = {k: (ConfigDict(v) if isinstance(v, dict) else v) for k, v in d.items()}
            self._locked, self._type_safe = False, True

    ConfigDict.__module__ = names[2]
    ConfigDict.__qualname__ = "ConfigDict"
    mods[names[2]].ConfigDict = ConfigDict
    sys.modules.update(mods)
    try:
        sd = {"model.mlp.6.weight": torch.arange(12.0).reshape(3, 4), "model.mlp.6.bias": torch.ones(3)}
        cfg = {"conv": "PNA", "hidden_dim": 128, "nested": {"lr": 1e-3, "deg": [1, 2, 3]}}
        path = tmp_path / "last.ckpt"
        torch.save({"state_dict": sd, "global_step": 7, "epoch": 2, "pytorch-lightning_version": "2.5.0",
                    "hyper_parameters": {"config": ConfigDict(cfg)}, "hparams_name": "kwargs"}, path)
    finally:
        for n in names:
            sys.modules.pop(n, None)
    with pytest.raises(Exception):  # the plain safe loader refuses the unknown class
        torch.load(path, map_location="cpu", weights_only=True)
    ckpt = read_checkpoint(str(path))
    assert ckpt["global_step"] == 7 and ckpt["hyper_parameters"]["config"] == cfg
    assert torch.equal(ckpt["state_dict"]["model.mlp.6.weight"], sd["model.mlp.6.weight"])
    assert not any(n in sys.modules for n in names)


# ---------------------------------------------------------------------------------------------------------------
# bench.py launcher: `--gpus N` must really be N ranks (it used to run one rank silently)
# ---------------------------------------------------------------------------------------------------------------
def _run_bench(argv, env_extra=None, timeout=180):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_gpus_flag_starts_that_many_ranks():
    import json
    r = _run_bench(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["steps"] == 2 and out["warmup"] == 1


def test_bench_under_torchrun_is_one_of_the_ranks():
    """The driver's launch line: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N."""
    import json
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"], env=env, capture_output=True,
                       text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2, r.stdout


def test_bench_refuses_a_rank_count_it_cannot_honour():
    r = _run_bench(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr
    if torch.cuda.device_count() < 3:
        r = _run_bench(["--gpus", "3", "--steps", "1"])  # real run: needs 3 visible devices
        assert r.returncode == 2 and "HIP device(s) visible" in r.stderr
