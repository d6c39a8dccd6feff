"""GPU parity tests, op by op, through the C ABI (ctypes) against the CPU oracle / numpy on the same seeded inputs.

Tolerances: integer / index work bit-exact; min/max/mean of the scatter-aggregate bit-exact (same summation order as
the CPU scatter); everything that re-associates fp32 sums within 1e-5 norm-wise relative (max|a-b|/max|ref|), the
north-star tolerance.
"""
import math

import numpy as np
import pytest
import torch

from oracle import pyg_restatement as O
from gnnepcsaft_amd import _lib
from tests.parity_util import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _graph(rng, N, E, empty_rows=True):
    src = rng.integers(0, N, size=E)
    dst = rng.integers(0, N if not empty_rows else max(1, (3 * N) // 4), size=E)
    return torch.from_numpy(np.stack([src, dst])).long()


def _pack(ei, ea, batch, N, B, dev):
    from gnnepcsaft_amd import ops
    return ops.pack_graph(ei.to(dev), None if ea is None else ea.to(dev), None if batch is None else batch.to(dev), N,
                          B)


# ---------------------------------------------------------------------------------------------------------------
# packer: bit-exact integer work
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,E", [(1, 0), (5, 0), (7, 13), (200, 1000), (5000, 20011), (70000, 140000)])
def test_pack_csr_bit_exact(gpu_device, N, E):
    rng = np.random.default_rng(N * 7 + E)
    ei = _graph(rng, N, E)
    ea = torch.from_numpy(np.stack([rng.integers(0, d, size=E) for d in (5, 6, 2)], 1)).long()
    sizes = rng.multinomial(N, np.ones(4) / 4) if N >= 4 else np.array([N])
    batch = torch.from_numpy(np.repeat(np.arange(len(sizes)), sizes)).long()
    g = _pack(ei, ea, batch, N, len(sizes), gpu_device)
    dst = ei[1].numpy()
    perm = np.argsort(dst, kind="stable")
    rowptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(np.bincount(dst, minlength=N), out=rowptr[1:])
    assert np.array_equal(g.rowptr.cpu().numpy(), rowptr)
    assert np.array_equal(g.perm.cpu().numpy(), perm)
    assert np.array_equal(g.src.cpu().numpy(), ei[0].numpy()[perm])
    assert np.array_equal(g.dst.cpu().numpy(), dst[perm])
    src_sorted = ei[0].numpy()[perm]
    cpos = np.argsort(src_sorted, kind="stable")
    colptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(np.bincount(src_sorted, minlength=N), out=colptr[1:])
    assert np.array_equal(g.colptr.cpu().numpy(), colptr)
    assert np.array_equal(g.cpos.cpu().numpy(), cpos)
    code = (ea[:, 0] * 6 + ea[:, 1]) * 2 + ea[:, 2]
    assert np.array_equal(g.code.cpu().numpy(), code.numpy()[perm])
    gptr = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(sizes, out=gptr[1:])
    assert np.array_equal(g.graph_ptr.cpu().numpy(), gptr)


def test_pack_range_errors(gpu_device):
    from gnnepcsaft_amd import ops
    from gnnepcsaft_amd._lib import GnxError
    ei = torch.tensor([[0, 1, 5], [1, 0, 2]])
    with pytest.raises(GnxError):
        ops.pack_graph(ei.to(gpu_device), None, None, 3)
    ei = torch.tensor([[0, 1], [1, 0]])
    ea = torch.tensor([[0, 0, 0], [5, 0, 0]])
    with pytest.raises(GnxError):
        ops.pack_graph(ei.to(gpu_device), ea.to(gpu_device), None, 2)
    with pytest.raises(GnxError):
        ops.pack_graph(ei.to(gpu_device), None, torch.tensor([1, 0]).to(gpu_device), 2, 2)
    # flag is cleared after being reported
    ops.pack_graph(ei.to(gpu_device), None, torch.tensor([0, 1]).to(gpu_device), 2, 2)


def test_degree_scalers(gpu_device):
    rng = np.random.default_rng(3)
    N, E = 300, 900
    ei = _graph(rng, N, E)
    g = _pack(ei, None, None, N, None, gpu_device)
    amp, att = g.degree_scalers(1.2345)
    d = O.degree(ei[1], N, dtype=torch.float32)
    avg = torch.tensor([1.2345])
    assert rel_err(amp, torch.log(d + 1) / avg) <= 1e-6
    assert rel_err(att, avg / torch.log(d.clamp(min=1) + 1)) <= 1e-6


# ---------------------------------------------------------------------------------------------------------------
# embeddings
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("H", [128, 20, 7])
def test_embed_sum(gpu_device, H):
    from gnnepcsaft_amd import nn as gnn
    torch.manual_seed(1)
    enc_o = O.AtomEncoder(H)
    enc_n = gnn.AtomEncoder(H)
    enc_n.load_state_dict(enc_o.state_dict())
    enc_n.to(gpu_device)
    rng = np.random.default_rng(5)
    N = 1000
    x = torch.from_numpy(np.stack([rng.integers(0, d, size=N) for d in O.ATOM_FEATURE_DIMS], 1)).long()
    yo = enc_o(x)
    yn = enc_n(x.to(gpu_device))
    assert torch.equal(yn.cpu(), yo), "embedding sum must be bit-exact (same left-to-right fp32 sum)"
    w = torch.randn(N, H)
    (yo * w).sum().backward()
    (yn * w.to(gpu_device)).sum().backward()
    for eo, en in zip(enc_o.atom_embedding_list, enc_n.atom_embedding_list):
        assert rel_err(en.weight.grad, eo.weight.grad) <= TOL


def test_embed_range_flag(gpu_device):
    from gnnepcsaft_amd import nn as gnn, ops
    from gnnepcsaft_amd._lib import GnxError
    enc = gnn.BondEncoder(16).to(gpu_device)
    bad = torch.tensor([[0, 6, 0]]).to(gpu_device)
    enc(bad)
    with pytest.raises(GnxError):
        ops.check_range(gpu_device)
    ops.check_range(gpu_device)


# ---------------------------------------------------------------------------------------------------------------
# dense
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(1, 3, 32), (130, 128, 128), (1000, 64, 100), (257, 200, 36), (60, 128, 128),
                                   (4096, 3, 32), (513, 130, 7)])
@pytest.mark.parametrize("relu", [False, True])
def test_gemm_nt(gpu_device, M, N, K, relu):
    from gnnepcsaft_amd import ops
    torch.manual_seed(M + N + K)
    a, w, b = torch.randn(M, K), torch.randn(N, K), torch.randn(N)
    ref = a.double() @ w.double().T + b.double()
    if relu:
        ref = ref.relu()
    out = torch.empty(M, N, device=gpu_device)
    ops.gemm([(a.to(gpu_device), None, w.to(gpu_device))], out, bias=b.to(gpu_device), relu=relu)
    assert rel_err(out, ref) <= TOL


def test_gemm_asymmetric_identity(gpu_device):
    """A = I with an asymmetric B catches a transposed C write (cdna_hip_programming.md §3)."""
    from gnnepcsaft_amd import ops
    K = N = 128
    a = torch.eye(K)
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) / 100.0
    out = torch.empty(K, N, device=gpu_device)
    ops.gemm([(a.to(gpu_device), None, w.to(gpu_device))], out)
    assert torch.equal(out.cpu(), w.T.contiguous())
    out2 = torch.empty(K, N, device=gpu_device)
    wn = w.T.contiguous()  # [k, n]
    ops.gemm([(a.to(gpu_device), None, wn.to(gpu_device))], out2, b_trans=False)
    assert torch.equal(out2.cpu(), wn)


@pytest.mark.parametrize("M,N,K", [(130, 128, 128), (1000, 100, 64), (77, 32, 3), (300, 36, 200)])
def test_gemm_nn_mask_accumulate(gpu_device, M, N, K):
    from gnnepcsaft_amd import ops
    torch.manual_seed(M * 3 + N)
    a, b = torch.randn(M, K), torch.randn(K, N)
    mask = torch.randn(M, N)
    c0 = torch.randn(M, N)
    ref = (a.double() @ b.double()) * (mask > 0)
    out = torch.empty(M, N, device=gpu_device)
    ops.gemm([(a.to(gpu_device), None, b.to(gpu_device))], out, b_trans=False, mask=mask.to(gpu_device))
    assert rel_err(out, ref) <= TOL
    out = c0.clone().to(gpu_device)
    ops.gemm([(a.to(gpu_device), None, b.to(gpu_device))], out, b_trans=False, accumulate=True)
    assert rel_err(out, c0.double() + a.double() @ b.double()) <= TOL


def test_gemm_segments_rowscale_strided(gpu_device):
    """The 13F-wide PNA post-layer operand [x | A | amp*A | att*A] as a 4-segment product on strided views."""
    from gnnepcsaft_amd import ops
    torch.manual_seed(11)
    M, F, T = 333, 32, 2
    H = T * F
    x, A = torch.randn(M, H), torch.randn(M, T * 4 * F)
    amp, att = torch.rand(M) + 0.5, torch.rand(M) + 0.5
    W = torch.randn(F, 13 * F)
    bias = torch.randn(F)
    t = 1
    xt, At = x[:, t * F:(t + 1) * F], A[:, t * 4 * F:(t + 1) * 4 * F]
    cat = torch.cat([xt, At, At * amp[:, None], At * att[:, None]], 1)
    ref = cat.double() @ W.double().T + bias.double()
    xd, Ad, Wd = x.to(gpu_device), A.to(gpu_device), W.to(gpu_device)
    out_full = torch.zeros(M, H, device=gpu_device)
    ops.gemm([(xd[:, t * F:(t + 1) * F], None, Wd[:, 0:F]),
              (Ad[:, t * 4 * F:(t + 1) * 4 * F], None, Wd[:, F:5 * F]),
              (Ad[:, t * 4 * F:(t + 1) * 4 * F], amp.to(gpu_device), Wd[:, 5 * F:9 * F]),
              (Ad[:, t * 4 * F:(t + 1) * 4 * F], att.to(gpu_device), Wd[:, 9 * F:13 * F])],
             out_full[:, t * F:(t + 1) * F], bias=bias.to(gpu_device))
    assert rel_err(out_full[:, t * F:(t + 1) * F], ref) <= TOL
    assert float(out_full[:, :F].abs().max()) == 0.0, "wrote outside the strided output view"
    # the matching input gradient: dA = g W1 + amp*(g W2) + att*(g W3)
    g = torch.randn(M, F)
    ref_dA = g.double() @ W[:, F:5 * F].double() + (g * amp[:, None]).double() @ W[:, 5 * F:9 * F].double() + \
        (g * att[:, None]).double() @ W[:, 9 * F:13 * F].double()
    gd = g.to(gpu_device)
    dA = torch.empty(M, 4 * F, device=gpu_device)
    ops.gemm([(gd, None, Wd[:, F:5 * F]), (gd, amp.to(gpu_device), Wd[:, 5 * F:9 * F]),
              (gd, att.to(gpu_device), Wd[:, 9 * F:13 * F])], dA, b_trans=False)
    assert rel_err(dA, ref_dA) <= TOL


def test_gemm_mid_size_kernel_against_the_tiled_kernel(gpu_device):
    """k_gemm_mid (M < 4096 and <= 48 tiles of 128 x 128: 16 x 16 patches on many workgroups) against the tiled fp32-MFMA
    kernel (GNX_OPT_GEMM_MID = 0) and fp64: a 32-graph batch's shapes (640 atoms, 1 280 bonds), multi-segment, row scale,
    mask / accumulate, both weight layouts, unaligned views, and the degree-class grouped post-layer-0 pair."""
    from gnnepcsaft_amd import _lib, ops
    dev = torch.device("cuda:0")

    def both(fn):
        outs = []
        for on in (1, 0):
            ops.set_option(dev, _lib.OPT_GEMM_MID, on)
            try:
                outs.append(fn())
            finally:
                ops.set_option(dev, _lib.OPT_GEMM_MID, 1)
        return outs

    torch.manual_seed(19)
    for M, N, K in [(640, 128, 128), (1280, 128, 128), (640, 128, 384), (3000, 256, 100), (515, 130, 36)]:
        a, w, wt = torch.randn(M, K, device=gpu_device), torch.randn(K, N, device=gpu_device), torch.randn(N, K, device=gpu_device)
        b, mask, c0 = torch.randn(N, device=gpu_device), torch.randn(M, N, device=gpu_device), torch.randn(M, N, device=gpu_device)

        def nt():
            out = torch.full((M, N), float("nan"), device=gpu_device)
            ops.gemm([(a, None, wt)], out, bias=b, relu=True)
            return out
        p, q = both(nt)
        assert rel_err(p, q) <= 2e-6 and rel_err(p, (a.double() @ wt.double().T + b.double()).relu()) <= TOL

        def nn_mask():
            out = torch.full((M, N), float("nan"), device=gpu_device)
            ops.gemm([(a, None, w)], out, b_trans=False, mask=mask)
            return out
        p, q = both(nn_mask)
        assert rel_err(p, q) <= 2e-6 and rel_err(p, (a.double() @ w.double()) * (mask > 0)) <= TOL

        def nn_acc():
            out = c0.clone()
            ops.gemm([(a, None, w)], out, b_trans=False, accumulate=True)
            return out
        p, q = both(nn_acc)
        assert rel_err(p, q) <= 2e-6 and rel_err(p, c0.double() + a.double() @ w.double()) <= TOL
    # three segments with a row scale on strided views (the dx product's shape at 640 rows)
    M, F = 640, 128
    g3 = torch.randn(M, 3 * F + 8, device=gpu_device)
    ws = [torch.randn(F, F, device=gpu_device) / 4 for _ in range(3)]
    rs = torch.rand(M, device=gpu_device) + 0.5

    def three():
        out = torch.zeros(M, F + 8, device=gpu_device)
        ops.gemm([(g3[:, 4:4 + F], None, ws[0]), (g3[:, 4 + F:4 + 2 * F], rs, ws[1]), (g3[:, 4 + 2 * F:4 + 3 * F], None, ws[2])],
                 out[:, 4:4 + F], b_trans=False)
        return out
    p, q = both(three)
    ref = g3[:, 4:4 + F].double() @ ws[0].double() + (rs[:, None] * g3[:, 4 + F:4 + 2 * F]).double() @ ws[1].double() + \
        g3[:, 4 + 2 * F:4 + 3 * F].double() @ ws[2].double()
    assert rel_err(p[:, 4:4 + F], q[:, 4:4 + F]) <= 2e-6 and rel_err(p[:, 4:4 + F], ref) <= TOL
    assert float(p[:, :4].abs().max()) == 0.0 and float(p[:, 4 + F:].abs().max()) == 0.0
    # grouped by in-degree class: x W0^T + A Weff(d)^T and the matching dA
    rng = np.random.default_rng(2)
    Nn, E = 640, 1300
    gk = _pack(_graph(rng, Nn, E), None, None, Nn, None, gpu_device)
    dc = gk.degree_classes()
    assert dc is not None
    x, A = torch.randn(Nn, F, device=gpu_device), torch.randn(Nn, 4 * F, device=gpu_device)
    W, bd = torch.randn(F, 13 * F, device=gpu_device) / 8, torch.randn(F, device=gpu_device)
    weff = ops.pna_weff(W, F, dc.D, 1.2)
    amp, att = gk.degree_scalers(1.2)

    def grouped():
        z = torch.full((Nn, F), float("nan"), device=gpu_device)
        ops.gemm_grouped([(x, None, W[:, 0:F], 0), (A, None, weff[0], 4 * F * F)], z, dc, bias=bd, relu=True)
        return z
    p, q = both(grouped)
    cat = torch.cat([x, A, A * amp[:, None], A * att[:, None]], 1).double()
    assert rel_err(p, q) <= 2e-6 and rel_err(p, (cat @ W.double().T + bd.double()).relu()) <= TOL

    def grouped_da():
        dA = torch.full((Nn, 4 * F), float("nan"), device=gpu_device)
        ops.gemm_grouped([(x, None, weff[0], 4 * F * F)], dA, dc, b_trans=False)
        return dA
    p, q = both(grouped_da)
    assert rel_err(p, q) <= 2e-6 and not torch.isnan(p).any()


@pytest.mark.parametrize("M,N,K", [(1000, 128, 128), (50000, 128, 512), (60, 32, 64), (4099, 3, 32), (777, 130, 36),
                                   (200000, 64, 64)])
def test_gemm_wgrad(gpu_device, M, N, K):
    from gnnepcsaft_amd import ops
    torch.manual_seed(M + 5 * N)
    dc, a, rs = torch.randn(M, N), torch.randn(M, K), torch.rand(M) + 0.5
    ref = dc.double().T @ (a * rs[:, None]).double()
    dw = torch.zeros(N, K, device=gpu_device)
    db = torch.zeros(N, device=gpu_device)
    ops.gemm_wgrad(dc.to(gpu_device), a.to(gpu_device), dw, rowscale=rs.to(gpu_device), dbias=db)
    assert rel_err(dw, ref) <= TOL
    assert rel_err(db, dc.double().sum(0)) <= TOL
    # accumulation semantics (+=) and strided dW view
    big = torch.ones(N, 3 * K, device=gpu_device)
    ops.gemm_wgrad(dc.to(gpu_device), a.to(gpu_device), big[:, K:2 * K])
    assert rel_err(big[:, K:2 * K], 1.0 + dc.double().T @ a.double()) <= TOL
    assert float((big[:, :K] - 1).abs().max()) == 0.0 and float((big[:, 2 * K:] - 1).abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------------------------
# the scatter-aggregate
# ---------------------------------------------------------------------------------------------------------------
def _oracle_pna_agg(m, index, N, T, F):
    """[N, T, 4F] = cat(mean, min, max, std) with the oracle's aggregators (m: [E, T, F])."""
    outs = [O.MeanAggregation()(m, index, dim_size=N, dim=0), O.MinAggregation()(m, index, dim_size=N, dim=0),
            O.MaxAggregation()(m, index, dim_size=N, dim=0), O.StdAggregation()(m, index, dim_size=N, dim=0)]
    return torch.cat(outs, dim=-1)


@pytest.mark.parametrize("N,E,T,F", [(50, 200, 1, 128), (300, 1000, 4, 32), (64, 150, 2, 6), (1000, 0, 1, 16),
                                     (20000, 60000, 1, 128)])
@pytest.mark.parametrize("ties", [False, True])
def test_pna_aggregate_fwd_bwd(gpu_device, N, E, T, F, ties):
    from gnnepcsaft_amd import ops
    rng = np.random.default_rng(N + E + T)
    ei = _graph(rng, N, E)
    g = _pack(ei, None, None, N, None, gpu_device)
    torch.manual_seed(E + F)
    m = torch.randn(E, T, F)
    if ties:
        m = (m * 2).round() / 2  # many exact ties and zero-variance segments
    m.requires_grad_(True)
    ref = _oracle_pna_agg(m, ei[1], N, T, F)
    perm = g.perm.cpu().long()
    m_csr = m.detach()[perm].reshape(E, T * F).contiguous().to(gpu_device)
    A = ops.pna_aggregate_fwd(m_csr, g, T, F)
    A_ref = ref.detach().reshape(N, T * 4 * F)
    Av, Rv = A.cpu().view(N, T, 4, F), A_ref.view(N, T, 4, F)
    assert torch.equal(Av[:, :, 0], Rv[:, :, 0]), "mean must be bit-exact (same summation order as scatter_add_)"
    assert torch.equal(Av[:, :, 1], Rv[:, :, 1]) and torch.equal(Av[:, :, 2], Rv[:, :, 2]), "min/max bit-exact"
    assert torch.equal(Av[:, :, 3] == 0, Rv[:, :, 3] == 0), "std mask (<= sqrt(1e-5) -> 0) must agree exactly"
    assert rel_err(Av[:, :, 3], Rv[:, :, 3]) <= 1e-6
    dA = torch.randn(N, T, 4 * F)
    ref.backward(dA)
    dm_ref = m.grad[perm].reshape(E, T * F)
    dAd = dA.reshape(N, -1).contiguous().to(gpu_device)
    # (1) the CPU path's own arithmetic: std gradient divided by the forward's std (mean(x^2) - mean(x)^2 in fp32)
    old = ops.set_option(gpu_device, _lib.OPT_STD_BWD_CENTERED, 0)
    try:
        dm_cpu_like = ops.pna_aggregate_bwd(dAd, m_csr, A, g, T, F)
    finally:
        ops.set_option(gpu_device, _lib.OPT_STD_BWD_CENTERED, old)
    assert rel_err(dm_cpu_like, dm_ref) <= TOL
    # (2) the default (centred two-pass std as the divisor) against the oracle evaluated in fp64 on the same inputs:
    # the fp32 cancellation in mean(x^2) - mean(x)^2 costs the CPU path up to ~1e-3 relative on gradient entries of
    # nodes whose messages are close together; the default must be within 1e-5 of the exact gradient regardless, except
    # where fp32 and fp64 disagree about the hard std mask itself (entries dropped, counted, must stay rare)
    m64 = m.detach().double().requires_grad_(True)
    ref64 = _oracle_pna_agg(m64, ei[1], N, T, F)
    ref64.backward(dA.double())
    mask_differs = ((ref64.detach().view(N, T, 4, F)[:, :, 3] == 0) != (Rv[:, :, 3] == 0))  # [N, T, F]
    ok_edge = ~mask_differs[ei[1]][perm].reshape(E, T * F) if E else torch.ones(0, T * F, dtype=torch.bool)
    assert int((~ok_edge).sum()) <= max(1, ok_edge.numel() // 2000)
    dm = ops.pna_aggregate_bwd(dAd, m_csr, A, g, T, F).cpu().double()
    dm64 = m64.grad[perm].reshape(E, T * F)
    assert rel_err(dm * ok_edge, dm64 * ok_edge) <= TOL


def test_pna_known_answers(gpu_device):
    """ties split evenly; empty segments 0 (also all-negative input); std edges."""
    from gnnepcsaft_amd import ops
    ei = torch.tensor([[0, 0, 0, 0, 0], [0, 0, 0, 2, 2]])
    g = _pack(ei, None, None, 4, None, gpu_device)
    m = torch.tensor([[1.0], [1.0], [0.5], [-3.0], [-1.0]]).repeat(1, 4)
    A = ops.pna_aggregate_fwd(m.to(gpu_device), g, 1, 4).cpu().view(4, 4, 4)
    assert A[0, 2, 0] == 1.0 and A[2, 2, 0] == -1.0 and A[2, 1, 0] == -3.0
    assert float(A[1].abs().max()) == 0.0 and float(A[3].abs().max()) == 0.0
    dA = torch.zeros(4, 16)
    dA[:, 8:12] = 1.0  # d/dmax
    dm = ops.pna_aggregate_bwd(dA.to(gpu_device), m.to(gpu_device), A.reshape(4, 16).to(gpu_device), g, 1, 4).cpu()
    assert torch.equal(dm[:, 0], torch.tensor([0.5, 0.5, 0.0, 0.0, 1.0]))
    # torch quirk reproduced: an extremum equal to 0 (the zero-filled scatter target) counts one extra tie
    m0 = torch.tensor([[0.0], [-1.0], [0.0], [2.0], [2.0]]).repeat(1, 4)
    A0 = ops.pna_aggregate_fwd(m0.to(gpu_device), g, 1, 4)
    dm0 = ops.pna_aggregate_bwd(dA.to(gpu_device), m0.to(gpu_device), A0, g, 1, 4).cpu()
    assert torch.allclose(dm0[:, 0], torch.tensor([1 / 3, 0.0, 1 / 3, 0.5, 0.5]), rtol=1e-6, atol=0)
    # std: var in {0, 1e-6, 1e-5} -> 0 ; var = 2e-5 -> 4.4721e-3
    for var, want in [(0.0, 0.0), (1e-6, 0.0), (2e-5, 4.4721e-3)]:
        s = math.sqrt(var)
        mm = torch.tensor([[s], [-s]]).repeat(1, 4)
        gg = _pack(torch.tensor([[0, 0], [0, 0]]), None, None, 1, None, gpu_device)
        sd = ops.pna_aggregate_fwd(mm.to(gpu_device), gg, 1, 4).cpu()[0, 12]
        assert abs(float(sd) - want) <= 1e-6 * max(1.0, want), (var, float(sd))


@pytest.mark.parametrize("H", [128, 36])
def test_edge_combine(gpu_device, H):
    from gnnepcsaft_amd import ops
    rng = np.random.default_rng(9)
    N, E, R = 500, 1700, 60
    ei = _graph(rng, N, E)
    ea = torch.from_numpy(np.stack([rng.integers(0, d, size=E) for d in (5, 6, 2)], 1)).long()
    g = _pack(ei, ea, None, N, None, gpu_device)
    torch.manual_seed(2)
    P, Q, Te = torch.randn(N, H), torch.randn(N, H), torch.randn(R, H)
    perm = g.perm.cpu().long()
    code = ((ea[:, 0] * 6 + ea[:, 1]) * 2 + ea[:, 2])[perm]
    ref = (P[ei[1][perm]] + Q[ei[0][perm]] + Te[code]).relu()
    h1 = ops.edge_combine_fwd(P.to(gpu_device), Q.to(gpu_device), Te.to(gpu_device), g, relu=True)
    assert rel_err(h1, ref) <= 1e-6
    gr = torch.randn(E, H)
    dP, dQ, dTe = ops.edge_combine_bwd(gr.to(gpu_device), g, R)
    rP = torch.zeros(N, H).index_add_(0, ei[1][perm], gr)
    rQ = torch.zeros(N, H).index_add_(0, ei[0][perm], gr)
    rT = torch.zeros(R, H).index_add_(0, code, gr)
    assert rel_err(dP, rP) <= TOL and rel_err(dQ, rQ) <= TOL and rel_err(dTe, rT) <= TOL


@pytest.mark.parametrize("H", [256, 20])
def test_gine_aggregate(gpu_device, H):
    from gnnepcsaft_amd import ops
    rng = np.random.default_rng(10)
    N, E, R = 400, 1300, 60
    ei = _graph(rng, N, E)
    ea = torch.from_numpy(np.stack([rng.integers(0, d, size=E) for d in (5, 6, 2)], 1)).long()
    g = _pack(ei, ea, None, N, None, gpu_device)
    torch.manual_seed(3)
    x = torch.randn(N, H, requires_grad=True)
    Le = torch.randn(R, H, requires_grad=True)
    code = (ea[:, 0] * 6 + ea[:, 1]) * 2 + ea[:, 2]
    msg = (x[ei[0]] + Le[code]).relu()
    ref = O.scatter(msg, ei[1], 0, N, "sum") + x
    out = ops.gine_aggregate_fwd(x.detach().to(gpu_device), Le.detach().to(gpu_device), g, 0.0)
    assert rel_err(out, ref) <= TOL
    d = torch.randn(N, H)
    ref.backward(d)
    dx, dLe = ops.gine_aggregate_bwd(d.to(gpu_device), x.detach().to(gpu_device), Le.detach().to(gpu_device), g, 0.0)
    assert rel_err(dx, x.grad) <= TOL and rel_err(dLe, Le.grad) <= TOL


@pytest.mark.parametrize("mode", ["add", "mean", "max"])
def test_segment_pool(gpu_device, mode):
    from gnnepcsaft_amd import functional as Fn
    torch.manual_seed(4)
    sizes = [3, 1, 0, 7, 20, 2]
    N, H, B = sum(sizes), 64, len(sizes)
    batch = torch.repeat_interleave(torch.arange(B), torch.tensor(sizes))
    x = (torch.randn(N, H) * 2).round() / 2
    x.requires_grad_(True)
    agg = {"add": O.SumAggregation, "mean": O.MeanAggregation, "max": O.MaxAggregation}[mode]()
    ref = agg(x, batch, dim_size=B)
    ptr = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=gpu_device)
    xd = x.detach().to(gpu_device).requires_grad_(True)
    out = Fn.SegmentPoolFn.apply(xd, ptr, B, mode)
    assert rel_err(out, ref) <= 1e-6
    d = torch.randn(B, H)
    ref.backward(d)
    out.backward(d.to(gpu_device))
    assert rel_err(xd.grad, x.grad) <= 1e-6


# ---------------------------------------------------------------------------------------------------------------
# BatchNorm (+ReLU) and loss
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,H", [(2, 128), (1000, 128), (70001, 64), (300, 36), (4096, 32)])
@pytest.mark.parametrize("relu", [False, True])
def test_batchnorm_train(gpu_device, M, H, relu):
    from gnnepcsaft_amd import nn as gnn
    torch.manual_seed(M + H)
    x = torch.randn(M, H) * 3 + 5  # |mean| > std: the case the shifted sums are for
    bo = torch.nn.BatchNorm1d(H)
    with torch.no_grad():
        bo.weight.uniform_(0.5, 1.5)
        bo.bias.uniform_(-1, 1)
    bn = gnn.BatchNorm1d(H)
    bn.load_state_dict(bo.state_dict())
    bn.to(gpu_device)
    xo = x.clone().requires_grad_(True)
    yo = bo(xo)
    yo = yo.relu() if relu else yo
    xn = x.to(gpu_device).requires_grad_(True)
    yn = bn(xn, relu=relu)
    assert rel_err(yn, yo) <= TOL
    assert rel_err(bn.running_mean, bo.running_mean) <= TOL and rel_err(bn.running_var, bo.running_var) <= TOL
    assert int(bn.num_batches_tracked) == 1
    d = torch.randn(M, H)
    yo.backward(d)
    yn.backward(d.to(gpu_device))
    if M > 2:  # with M=2 rows dx is analytically 0: comparing it is noise over noise (amplified by rstd)
        assert rel_err(xn.grad, xo.grad) <= 2e-5
    assert rel_err(bn.weight.grad, bo.weight.grad) <= TOL and rel_err(bn.bias.grad, bo.bias.grad) <= TOL
    # eval mode uses running statistics
    bo.eval()
    bn.eval()
    assert rel_err(bn(x.to(gpu_device)), bo(x)) <= TOL


def test_batchnorm_single_row_raises(gpu_device):
    from gnnepcsaft_amd import nn as gnn
    bn = gnn.BatchNorm1d(8).to(gpu_device)
    with pytest.raises(ValueError):
        bn(torch.randn(1, 8, device=gpu_device))


@pytest.mark.parametrize("count", [(7, 3), (4096, 3), (100000, 2)])
def test_huber_ape(gpu_device, count):
    from gnnepcsaft_amd import functional as Fn
    torch.manual_seed(7)
    t = torch.rand(*count) * 10 + 1
    p = (t * (1 + torch.randn(*count) * 0.02)).requires_grad_(True)  # errors on both sides of delta = 0.01
    ref = O.ape_huber_loss(p, t)
    ref.backward()
    pn = p.detach().to(gpu_device).requires_grad_(True)
    loss, both = Fn.HuberAPEFn.apply(pn, t.to(gpu_device), 0.01)
    loss.backward()
    assert rel_err(loss, ref) <= TOL and rel_err(both[1], O.mape(p.detach(), t)) <= TOL
    assert rel_err(pn.grad, p.grad) <= TOL


# ---------------------------------------------------------------------------------------------------------------
# in-degree classes: PNA post-layer 0 with one effective weight per degree
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,E", [(1, 0), (300, 0), (777, 2000), (70000, 140000)])
def test_degree_classes_bit_exact(gpu_device, N, E):
    from gnnepcsaft_amd.ops import DegreeClasses
    rng = np.random.default_rng(N + E)
    ei = _graph(rng, N, E)
    g = _pack(ei, None, None, N, None, gpu_device)
    dc = g.degree_classes()
    deg = np.bincount(ei[1].numpy(), minlength=N)
    assert dc is not None and dc.D == int(deg.max()) + 1
    assert np.array_equal(dc.dperm.cpu().numpy()[:N], np.argsort(deg, kind="stable"))
    cls_ptr = np.concatenate([[0], np.cumsum(np.bincount(deg, minlength=dc.D))])
    assert np.array_equal(dc.cls_ptr.cpu().numpy(), cls_ptr)
    for rows, info, cnt in ((DegreeClasses.GEMM_ROWS, dc.tiles, dc.ntiles), (DegreeClasses.WGRAD_ROWS, dc.chunks, dc.nchunks)):
        want = []
        for c in range(dc.D):
            for r in range(cls_ptr[c], cls_ptr[c + 1], rows):
                want.append((r, min(rows, cls_ptr[c + 1] - r), c))
        n = int(cnt)
        assert n == len(want) and n <= info.numel() // 3
        assert np.array_equal(info.cpu().numpy()[:3 * n].reshape(n, 3), np.array(want, dtype=np.int64).reshape(n, 3))


def test_degree_classes_fallback_above_64(gpu_device):
    ei = torch.stack([torch.arange(1, 101), torch.zeros(100, dtype=torch.long)])  # one hub of in-degree 100
    g = _pack(ei, None, None, 101, None, gpu_device)
    assert g.degree_classes() is None


def test_pna_post0_degree_classes_equal_scaled_segments(gpu_device):
    """x W0^T + A Weff(d)^T (grouped by in-degree) == the 13F-wide [x | A | amp*A | att*A] product, forward, input
    gradient and weight gradient."""
    from gnnepcsaft_amd import ops
    rng = np.random.default_rng(21)
    N, E, F = 3000, 7000, 32
    ei = _graph(rng, N, E)
    g = _pack(ei, None, None, N, None, gpu_device)
    dc = g.degree_classes()
    avg = 1.137
    amp, att = g.degree_scalers(avg)
    torch.manual_seed(5)
    x, A = torch.randn(N, F), torch.randn(N, 4 * F)
    W, b = torch.randn(F, 13 * F) / 8, torch.randn(F)
    xd, Ad, Wd, bd = x.to(gpu_device), A.to(gpu_device), W.to(gpu_device), b.to(gpu_device)
    cat = torch.cat([x, A, A * amp.cpu()[:, None], A * att.cpu()[:, None]], 1).double()
    ref = (cat @ W.double().T + b.double()).relu()
    weff = ops.pna_weff(Wd, F, dc.D, avg)
    d = torch.arange(dc.D, dtype=torch.float32)
    a_d, t_d = torch.log(d + 1) / avg, avg / torch.log(d.clamp(min=1) + 1)
    weff_ref = W[None, :, F:5 * F] + a_d[:, None, None] * W[None, :, 5 * F:9 * F] + t_d[:, None, None] * W[None, :, 9 * F:]
    assert rel_err(weff, weff_ref) <= 1e-6
    z = torch.full((N, F), float("nan"), device=gpu_device)
    ops.gemm_grouped([(xd, None, Wd[:, 0:F], 0), (Ad, None, weff[0], 4 * F * F)], z, dc, bias=bd, relu=True)
    assert rel_err(z, ref) <= TOL
    # input gradient dA = g Weff(d)
    gr = torch.randn(N, F)
    grd = gr.to(gpu_device)
    ref_dA = gr.double() @ W[:, F:5 * F].double() + (gr * amp.cpu()[:, None]).double() @ W[:, 5 * F:9 * F].double() + \
        (gr * att.cpu()[:, None]).double() @ W[:, 9 * F:].double()
    dA = torch.full((N, 4 * F), float("nan"), device=gpu_device)
    ops.gemm_grouped([(grd, None, weff[0], 4 * F * F)], dA, dc, b_trans=False)
    assert rel_err(dA, ref_dA) <= TOL
    # weight gradient of the three A blocks
    dW = torch.zeros(F, 13 * F, device=gpu_device)
    ops.pna_post0_wgrad_classes(grd, Ad, dc, F, avg, dW)
    ref_dW = gr.double().T @ cat
    assert rel_err(dW[:, F:], ref_dW[:, F:]) <= TOL
    assert float(dW[:, :F].abs().max()) == 0.0


def test_clip_rows_propagates_nan(gpu_device):
    """Tensor.clip keeps NaN (reference pred_with_bounds, models.py:246-253); fminf/fmaxf alone would return a bound."""
    from gnnepcsaft_amd import ops
    x = torch.tensor([[0.5, float("nan"), 700.0], [float("nan"), 3.0, float("inf")]], device=gpu_device)
    lo = torch.tensor([1.0, 1.9, 50.0], device=gpu_device)
    hi = torch.tensor([25.0, 4.5, 550.0], device=gpu_device)
    y = ops.clip_rows(x, lo, hi).cpu()
    ref = x.cpu().clip(lo.cpu(), hi.cpu())
    assert torch.equal(torch.isnan(y), torch.isnan(ref))
    assert torch.equal(torch.nan_to_num(y, nan=-1.0), torch.nan_to_num(ref, nan=-1.0))


def test_options_and_caller_owned_gemm_workspace(gpu_device):
    """gnx_set_option round trip; gnx_gemm with ws = NULL runs the fp32-MFMA kernel, with a workspace the split-operand
    kernel, with a too-small workspace it returns GNX_E_WORKSPACE (library contract: no allocation inside)."""
    import ctypes as C
    from gnnepcsaft_amd import _lib, ops
    old = ops.set_option(gpu_device, _lib.OPT_WGRAD_WGS, 77)
    assert ops.set_option(gpu_device, _lib.OPT_WGRAD_WGS, old) == 77
    torch.manual_seed(3)
    M, N, K = 6000, 128, 320
    a, w = torch.randn(M, K, device=gpu_device), torch.randn(N, K, device=gpu_device) / 16
    ref = a.double() @ w.double().T
    lib, h = _lib.load(), _lib.handle(gpu_device)
    seg = (_lib.GemmSeg * 1)()
    seg[0].a, seg[0].lda, seg[0].rowscale, seg[0].b, seg[0].ldb, seg[0].k = a.data_ptr(), K, None, w.data_ptr(), K, K
    need = lib.gnx_gemm_workspace_bytes(h, 1, seg, None, 1, M, N, None, _lib.GEMM_B_TRANS, 0)
    assert need == 3 * 128 * 320 * 2
    out = torch.empty(M, N, device=gpu_device)
    _lib.check(lib.gnx_gemm(h, 1, seg, M, N, None, None, 0, out.data_ptr(), N, _lib.GEMM_B_TRANS, None, 0))
    assert rel_err(out, ref) <= 1e-5
    ws = torch.empty(need, dtype=torch.uint8, device=gpu_device)
    out2 = torch.empty(M, N, device=gpu_device)
    _lib.check(lib.gnx_gemm(h, 1, seg, M, N, None, None, 0, out2.data_ptr(), N, _lib.GEMM_B_TRANS, ws.data_ptr(), need))
    assert rel_err(out2, ref) <= 1e-5
    st = lib.gnx_gemm(h, 1, seg, M, N, None, None, 0, out2.data_ptr(), N, _lib.GEMM_B_TRANS, ws.data_ptr(), need - 16)
    assert st == _lib.GNX_E_WORKSPACE
    # a product the split path does not take needs no workspace
    assert lib.gnx_gemm_workspace_bytes(h, 1, seg, None, 1, 1000, N, None, _lib.GEMM_B_TRANS, 0) == 0
    # the split on its own (GNX_GEMM_SPLIT_ONLY: e.g. ahead of time on another stream), then the product on the images as
    # they are (GNX_GEMM_PRESPLIT): the same bits as the plain call; SPLIT_ONLY touches nothing but the workspace
    ws2 = torch.zeros(need, dtype=torch.uint8, device=gpu_device)
    out3 = torch.full((M, N), float("nan"), device=gpu_device)
    _lib.check(lib.gnx_gemm(h, 1, seg, M, N, None, None, 0, out3.data_ptr(), N, _lib.GEMM_B_TRANS | _lib.GEMM_SPLIT_ONLY,
                            ws2.data_ptr(), need))
    assert torch.isnan(out3).all() and torch.equal(ws2, ws)
    _lib.check(lib.gnx_gemm(h, 1, seg, M, N, None, None, 0, out3.data_ptr(), N, _lib.GEMM_B_TRANS | _lib.GEMM_PRESPLIT,
                            ws2.data_ptr(), need))
    assert torch.equal(out3, out2)
    # SPLIT_ONLY on a call that needs no images (weights-stationary shape) does nothing at all
    a2, w2 = torch.randn(9000, 128, device=gpu_device), torch.randn(128, 128, device=gpu_device)
    seg2 = (_lib.GemmSeg * 1)()
    seg2[0].a, seg2[0].lda, seg2[0].rowscale, seg2[0].b, seg2[0].ldb, seg2[0].k = a2.data_ptr(), 128, None, w2.data_ptr(), 128, 128
    out4 = torch.full((9000, 128), float("nan"), device=gpu_device)
    _lib.check(lib.gnx_gemm(h, 1, seg2, 9000, 128, None, None, 0, out4.data_ptr(), 128, _lib.GEMM_B_TRANS | _lib.GEMM_SPLIT_ONLY,
                            None, 0))
    assert torch.isnan(out4).all()
