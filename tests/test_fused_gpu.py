"""GPU tests of the fused PNA edge pipeline (gnx_edge_tiles, gnx_pna_edge_fwd: message assembly -> pre-layer 1 ->
mean|min|max|std aggregate in one kernel; reference semantics: PyG PNAConv.message + DegreeScalerAggregation as built at
/root/reference/gnnepcsaft/train/models.py:445-457 with the default pre_layers = 2).

Bars: the tile table is integer work -> bit-exact against numpy; h1 bit-exact against gnx_edge_combine_fwd; the messages
bit-exact against the three-launch sequence whenever that sequence takes the same split-operand product (>= 8192 message
rows; below that it runs the exact-fp32 MFMA kernel, and the two fp32-faithful products are compared at 2e-6); the
aggregate bit-exact against gnx_pna_aggregate_fwd of the fused kernel's own messages (same operation order); whole
models with the fused kernel on and off: identical predictions and loss.
"""
import copy

import numpy as np
import pytest
import torch

from tests.parity_util import rel_err

pytestmark = pytest.mark.gpu


def _mol_graph(rng, N, maxdeg, isolated=0.0, tail_empty=0):
    """Random directed edges with in-degree <= maxdeg (a fraction of nodes with no incoming edge, optionally a run of
    edge-less nodes at the end)."""
    deg = rng.integers(1, maxdeg + 1, size=N)
    deg[rng.random(N) < isolated] = 0
    if tail_empty:
        deg[-tail_empty:] = 0
    dst = np.repeat(np.arange(N), deg)
    src = rng.integers(0, N, size=dst.size)
    order = rng.permutation(dst.size)  # PyG order is not sorted by destination
    return torch.from_numpy(np.stack([src[order], dst[order]])).long()


def _tiles_numpy(rowptr, N, E, w):
    count = E // w + 1
    info = np.zeros((count + 1, 2), dtype=np.int64)
    for j in range(count):
        n = int(np.searchsorted(rowptr, w * j, side="left"))
        info[j] = (n, rowptr[n])
    info[count] = (N, E)
    return info


@pytest.mark.parametrize("N,maxdeg,isolated,tail", [(1, 1, 0.0, 0), (50, 4, 0.3, 5), (5000, 4, 0.05, 0), (3000, 12, 0.1, 40),
                                                    (200, 16, 0.5, 100)])
def test_edge_tiles_bit_exact(gpu_device, N, maxdeg, isolated, tail):
    from gnnepcsaft_amd import ops
    rng = np.random.default_rng(N + maxdeg)
    ei = _mol_graph(rng, N, maxdeg, isolated, tail)
    g = ops.pack_graph(ei.to(gpu_device), None, None, N)
    if g.E == 0:
        assert g.edge_tiles(maxdeg) is None
        return
    info, w = g.edge_tiles(maxdeg)
    assert w == 65 - maxdeg
    ref = _tiles_numpy(g.rowptr.cpu().numpy().astype(np.int64), N, g.E, w)
    got = info.cpu().numpy().reshape(-1, 2)
    assert np.array_equal(got, ref)
    assert (np.diff(ref[:, 1]) <= 64).all() and (np.diff(ref[:, 1]) >= 0).all()  # no tile above 64 message rows
    assert g.edge_tiles(17) is None  # above the bound the fused kernels are not offered


CASES = [  # N, maxdeg, T, F, isolated, tail_empty
    (3, 2, 1, 32, 0.0, 0),
    (700, 4, 1, 128, 0.1, 9),
    (6000, 4, 1, 128, 0.02, 0),       # >= 8192 message rows: the unfused product is k_gemm_ws3 -> bit-exact messages
    (5000, 4, 4, 32, 0.05, 0),
    (4000, 4, 2, 64, 0.05, 3),
    (2500, 12, 4, 128, 0.05, 0),      # cfg-5's layer shape: H = 512, T = 4, skewed degrees
    (900, 16, 1, 36, 0.3, 50),        # partial last k-slab, many empty rows
    (40000, 4, 1, 128, 0.0, 0),       # more tiles than workgroups: the persistent pipeline over many tiles
]


@pytest.mark.parametrize("N,maxdeg,T,F,isolated,tail", CASES)
def test_pna_edge_fwd_equals_three_launch_sequence(gpu_device, N, maxdeg, T, F, isolated, tail):
    from gnnepcsaft_amd import ops
    dev = gpu_device
    rng = np.random.default_rng(N * 31 + F)
    H = T * F
    ei = _mol_graph(rng, N, maxdeg, isolated, tail)
    E = ei.size(1)
    ea = torch.from_numpy(np.stack([rng.integers(0, d, size=E) for d in (5, 6, 2)], 1)).long()
    g = ops.pack_graph(ei.to(dev), ea.to(dev), None, N)
    gen = torch.Generator().manual_seed(N + F)
    P, Q = (torch.randn(N, H, generator=gen).to(dev) for _ in range(2))
    Te = torch.randn(60, H, generator=gen).to(dev)
    Ws = [(torch.randn(F, F, generator=gen) / F ** 0.5).to(dev) for _ in range(T)]
    bs = [torch.randn(F, generator=gen).to(dev) for _ in range(T)]
    h1, m, A = ops.pna_edge_fwd(P, Q, Te, g, T, F, Ws, bs, maxdeg)
    # the three-launch sequence
    h1_ref = ops.edge_combine_fwd(P, Q, Te, g, relu=True)
    m_ref = torch.empty(E, H, device=dev)
    for t in range(T):
        ops.gemm([(h1_ref[:, t * F:(t + 1) * F], None, Ws[t])], m_ref[:, t * F:(t + 1) * F], bias=bs[t])
    ops.check_range(dev)
    torch.cuda.synchronize()
    assert torch.equal(h1, h1_ref)
    if E >= 8192 and F >= 32:
        assert torch.equal(m, m_ref), float((m - m_ref).abs().max())
    else:
        assert rel_err(m, m_ref) <= 2e-6
    A_ref = ops.pna_aggregate_fwd(m, g, T, F)
    assert torch.equal(A, A_ref)
    # messages not kept (inference form): same aggregate
    _, _, A2 = ops.pna_edge_fwd(P, Q, Te, g, T, F, Ws, bs, maxdeg, keep=False)
    assert torch.equal(A2, A)


def test_pna_edge_fwd_degree_above_the_bound_sets_the_range_flag(gpu_device):
    """A tile table built for in-degrees <= 2 on a graph with in-degree 40: rows are dropped, nothing is read or written
    out of bounds, and the sticky range flag reports it."""
    from gnnepcsaft_amd import ops
    from gnnepcsaft_amd._lib import GnxError
    dev = gpu_device
    N, F = 300, 32
    rng = np.random.default_rng(5)
    dst = np.concatenate([np.repeat(np.arange(N), 2), np.full(80, 7)])
    src = rng.integers(0, N, size=dst.size)
    ei = torch.from_numpy(np.stack([src, dst])).long()
    g = ops.pack_graph(ei.to(dev), None, None, N)
    P, Q, Te = torch.randn(N, F).to(dev), torch.randn(N, F).to(dev), torch.randn(60, F).to(dev)
    W, b = torch.randn(F, F).to(dev), torch.randn(F).to(dev)
    ops.pna_edge_fwd(P, Q, Te, g, 1, F, [W], [b], 2)
    with pytest.raises(GnxError):
        ops.check_range(dev)
    ops.check_range(dev)  # cleared after being reported


@pytest.mark.parametrize("name", ["pna_small", "pna_towers4", "pna_cfg2_full_1024", "pna_cfg5_shaped", "pna_lone_atoms",
                                  "pna_esper_molecules"])
def test_model_with_and_without_the_fused_edge_kernel(gpu_device, name):
    """Whole model, training-mode forward + loss + backward, fused edge kernel on vs off: predictions and loss identical
    (the fused kernel is bit-identical wherever the unfused product takes the split-operand kernel too, and both are
    fp32-faithful products otherwise); gradients equal up to the order of the weight-gradient atomics."""
    from gnnepcsaft_amd import functional as Fn
    from gnnepcsaft_amd.train.models import GNNePCSAFT
    from tests.model_cases import build_case
    cfg, batch, target = build_case(name)
    torch.manual_seed(0)
    model = GNNePCSAFT(cfg).to("cuda:0").train()
    state = copy.deepcopy(model.state_dict())
    b = batch.to("cuda:0")
    out = {}
    for fused in (True, False):
        Fn.set_fused_edge(fused)
        try:
            model.load_state_dict(state)
            model.zero_grad(set_to_none=True)
            model.max_degree_hint = len(cfg["deg"]) - 1
            if hasattr(b, "_gnx_pack"):
                del b._gnx_pack
            pred = model(b.x, b.edge_index, b.edge_attr, b.batch)
            loss, _ = Fn.HuberAPEFn.apply(pred, getattr(b, target), 0.01)
            loss.backward()
            torch.cuda.synchronize()
            out[fused] = (pred.detach().clone(), loss.detach().clone(),
                          {n: p.grad.detach().clone() for n, p in model.named_parameters()})
        finally:
            Fn.set_fused_edge(True)
    (p1, l1, g1), (p0, l0, g0) = out[True], out[False]
    big = batch.edge_index.size(1) >= 8192
    if big:
        assert torch.equal(p1, p0) and torch.equal(l1, l0)
    else:
        # below 8192 message rows the unfused pre-layer-1 product runs on the exact-fp32 MFMA kernel, the fused one always on
        # split operands: two fp32-faithful evaluations whose last-bit differences the model's discrete decisions (std
        # mask, min / max winners, ReLU after BatchNorm) amplify like any other rounding (tests/parity_util.py) --
        # a sanity bound only; the bit-exact statement is the >= 8192-row cases and the op-level tests above
        assert rel_err(p1, p0) <= 5e-3 and rel_err(l1, l0) <= 1e-3
    G = max(float(v.abs().max()) for v in g0.values())
    for n in g0:
        # (fp32 atomics of the weight gradients arrive in another order every run: REPEATS of one configuration differ by up
        # to 8.8e-5 in this metric over 30 runs -- tools/atomics_noise.py, worst on the lin biases, whose gradient in front of
        # BatchNorm is analytically zero, i.e. pure cancellation -- so 1e-4 failed once in a while; a missing launch shows at 1e-1)
        assert rel_err(g1[n], g0[n], floor=1e-2 * G) <= (1e-3 if big else 0.2), n


@pytest.mark.parametrize("name", ["pna_small", "pna_towers4", "pna_cfg2_shape_256", "pna_pre1_post1", "pna_hubs"])
def test_batched_weight_only_work_equals_the_per_layer_launches(gpu_device, name):
    """The weight-only work of all layers in a few batched launches (gnx_pna_weight_only_all forward, gnx_pna_stack_finish
    at the end of backward) vs the per-layer launch sequences: the forward products are the same k-ordered fmaf chains
    (bit-identical predictions and loss); the deferred gradients re-associate a few 60-row sums (1e-5 of the largest
    gradient entry).  Needs in-place gradient sinks (the training configuration: flat gradient buffer)."""
    from gnnepcsaft_amd import dp, ops
    from gnnepcsaft_amd import functional as Fn
    from gnnepcsaft_amd.train.models import GNNePCSAFT
    from tests.model_cases import build_case
    cfg, batch, target = build_case(name)
    torch.manual_seed(0)
    model = GNNePCSAFT(cfg).to("cuda:0").train()
    state = copy.deepcopy(model.state_dict())
    b = batch.to("cuda:0")
    flat = dp.FlatGradAllReduce(model)
    Fn.set_grad_in_place(True)
    ops.set_wgrad_side_stream(True)
    out = {}
    try:
        for batched in (True, False):
            Fn.set_batch_weight_only(batched)
            model.load_state_dict(state)
            flat.zero_grad()
            pred = model(b.x, b.edge_index, b.edge_attr, b.batch)
            loss, _ = Fn.HuberAPEFn.apply(pred, getattr(b, target), 0.01)
            loss.backward()
            torch.cuda.synchronize()
            out[batched] = (pred.detach().clone(), loss.detach().clone(), flat.flat.detach().clone())
    finally:
        Fn.set_batch_weight_only(True)
        Fn.set_grad_in_place(False)
        ops.set_wgrad_side_stream(False)
    (p1, l1, g1), (p0, l0, g0) = out[True], out[False]
    assert torch.equal(p1, p0) and torch.equal(l1, l0)
    assert float((g1 - g0).abs().max()) <= 1e-5 * float(g0.abs().max()), float((g1 - g0).abs().max() / g0.abs().max())


@pytest.mark.parametrize("N,maxdeg,T,F,isolated,tail", CASES)
def test_pna_edge_bwd_equals_three_launch_sequence(gpu_device, N, maxdeg, T, F, isolated, tail):
    """gnx_pna_edge_bwd (masked input gradient of pre-layer 1 + destination sums + bond-table sums in one pass over the
    message gradient) vs gnx_gemm(mask) + gnx_edge_combine_bwd + gnx_key_segment_sum: gh1 bit-exact where the three-launch
    product takes the split-operand kernel too (>= 8192 rows), dP bit-exact given gh1 (same summation order), dTe 1e-5
    (one-hot MFMA sums vs inverted-index sums: another order)."""
    from gnnepcsaft_amd import ops
    dev = gpu_device
    rng = np.random.default_rng(N * 17 + F)
    H = T * F
    ei = _mol_graph(rng, N, maxdeg, isolated, tail)
    E = ei.size(1)
    ea = torch.from_numpy(np.stack([rng.integers(0, d, size=E) for d in (5, 6, 2)], 1)).long()
    g = ops.pack_graph(ei.to(dev), ea.to(dev), None, N)
    gen = torch.Generator().manual_seed(N + 3 * F)
    ge = torch.randn(E, H, generator=gen).to(dev)
    h1 = torch.randn(E, H, generator=gen).relu_().to(dev)       # about half of the entries masked
    Ws = [(torch.randn(F, F, generator=gen) / F ** 0.5).to(dev) for _ in range(T)]
    gh1, dP, dTe = ops.pna_edge_bwd(ge, h1, g, T, F, 60, Ws, maxdeg)
    ref = torch.empty(E, H, device=dev)
    for t in range(T):
        sl = slice(t * F, (t + 1) * F)
        ops.gemm([(ge[:, sl], None, Ws[t])], ref[:, sl], b_trans=False, mask=h1[:, sl])
    ops.check_range(dev)
    torch.cuda.synchronize()
    if E >= 8192 and F >= 32:
        assert torch.equal(gh1, ref), float((gh1 - ref).abs().max())
    else:
        assert rel_err(gh1, ref) <= 2e-6
    dP_ref, _ = ops.edge_combine_bwd_pq(gh1, g)
    assert torch.equal(dP, dP_ref)
    dTe_ref = ops.bond_table_grad(gh1, g, 60, ops.bond_code_index(g, 60, H))
    assert rel_err(dTe, dTe_ref) <= 1e-5
    assert rel_err(dTe, torch.zeros(60, H, dtype=torch.float64).index_add_(0, g.code.long().cpu(), gh1.double().cpu())) <= 1e-5
