"""GPU parity of the tiled split-operand GEMM (k_split_weights + k_gemm3: multi-segment / grouped / wide products with
M >= 4096) against fp64 torch: both weight layouts, ragged M / K / N, several column tiles, persistent tile walk over
more tiles than workgroups, degree-class grouping, all epilogues, strided views; and against the exact-fp32 MFMA
kernel (GNX_GEMM_SPLIT=0)."""
import os

import numpy as np
import pytest
import torch

from gnnepcsaft_amd import _lib
from tests.parity_util import rel_err
from tests.test_ops_gpu import _graph, _pack

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(params=[1, 0], ids=["wave_specialised", "two_barrier"])
def wgrad_kernel(request, gpu_device):
    """Runs a weight-gradient test through k_gemm_wgrad3p (default) and through k_gemm_wgrad3 (GNX_OPT_WGRAD_PIPE = 0)."""
    from gnnepcsaft_amd import ops
    ops.set_option(torch.device("cuda:0"), _lib.OPT_WGRAD_PIPE, request.param)
    yield request.param
    ops.set_option(torch.device("cuda:0"), _lib.OPT_WGRAD_PIPE, 1)


@pytest.mark.parametrize("M", [4096, 20037, 70000])
def test_two_segments_nt_bias_relu(gpu_device, M):
    """post-layer-0 shape: [x | A] (K = 128 + 512) against W[:, :640], bias + ReLU."""
    from gnnepcsaft_amd import ops
    torch.manual_seed(M)
    F = 128
    x, A = torch.randn(M, F), torch.randn(M, 4 * F)
    W, b = torch.randn(F, 13 * F) / 8, torch.randn(F)
    ref = (torch.cat([x, A], 1).double() @ W[:, :5 * F].double().T + b.double()).relu()
    out = torch.full((M, F), float("nan"), device=gpu_device)
    Wd = W.to(gpu_device)
    ops.gemm([(x.to(gpu_device), None, Wd[:, :F]), (A.to(gpu_device), None, Wd[:, F:5 * F])], out,
             bias=b.to(gpu_device), relu=True)
    assert rel_err(out, ref) <= TOL


@pytest.mark.parametrize("M,N,K", [(8200, 512, 128), (4100, 132, 100), (33000, 256, 256)])
def test_nn_wide_mask_and_accumulate(gpu_device, M, N, K):
    """dA shape (several column tiles share the A rows), NN weights, ReLU-mask and accumulate epilogues."""
    from gnnepcsaft_amd import ops
    torch.manual_seed(M + N)
    a, w = torch.randn(M, K), torch.randn(K, N) / 4
    mask, c0 = torch.randn(M, N), torch.randn(M, N)
    ad, wd = a.to(gpu_device), w.to(gpu_device)
    prod = a.double() @ w.double()
    out = torch.full((M, N), float("nan"), device=gpu_device)
    ops.gemm([(ad, None, wd)], out, b_trans=False)
    assert rel_err(out, prod) <= TOL
    out = torch.full((M, N), float("nan"), device=gpu_device)
    ops.gemm([(ad, None, wd)], out, b_trans=False, mask=mask.to(gpu_device))
    assert rel_err(out, prod * (mask > 0)) <= TOL
    out = c0.clone().to(gpu_device)
    ops.gemm([(ad, None, wd)], out, b_trans=False, accumulate=True)
    assert rel_err(out, c0.double() + prod) <= TOL


def test_three_segments_nn_ragged_k_strided_views(gpu_device):
    """dx shape: three products accumulated in one launch; k not a multiple of 32, operands and output are column
    slices of wider tensors; nothing is written outside the output view."""
    from gnnepcsaft_amd import ops
    torch.manual_seed(4)
    M, H = 9001, 384
    G3 = torch.randn(M, H)
    W = torch.randn(3, 128, 132) / 4
    gd, wd = G3.to(gpu_device), W.to(gpu_device)
    ks = (100, 36, 128)
    segs, ref = [], 0
    for i, k in enumerate(ks):
        segs.append((gd[:, 128 * i:128 * i + k], None, wd[i, :k, :]))
        ref = ref + G3[:, 128 * i:128 * i + k].double() @ W[i, :k, :].double()
    out = torch.zeros(M, 512, device=gpu_device)
    ops.gemm(segs, out[:, 128:260], b_trans=False)
    assert rel_err(out[:, 128:260], ref) <= TOL
    assert float(out[:, :128].abs().max()) == 0.0 and float(out[:, 260:].abs().max()) == 0.0


def test_grouped_by_degree_class(gpu_device):
    """x W0^T + A Weff(d)^T with rows gathered per in-degree class (tiles never straddle classes; more tiles than
    workgroups so the persistent walk wraps) and the matching input gradient."""
    from gnnepcsaft_amd import ops
    rng = np.random.default_rng(33)
    N, E, F = 90000, 200000, 64
    ei = _graph(rng, N, E)
    g = _pack(ei, None, None, N, None, gpu_device)
    dc = g.degree_classes()
    assert dc is not None
    avg = 1.2
    amp, att = g.degree_scalers(avg)
    torch.manual_seed(6)
    x, A = torch.randn(N, F), torch.randn(N, 4 * F)
    W, b = torch.randn(F, 13 * F) / 8, torch.randn(F)
    xd, Ad, Wd, bd = x.to(gpu_device), A.to(gpu_device), W.to(gpu_device), b.to(gpu_device)
    cat = torch.cat([x, A, A * amp.cpu()[:, None], A * att.cpu()[:, None]], 1).double()
    ref = (cat @ W.double().T + b.double()).relu()
    weff = ops.pna_weff(Wd, F, dc.D, avg)
    z = torch.full((N, F), float("nan"), device=gpu_device)
    ops.gemm_grouped([(xd, None, Wd[:, 0:F], 0), (Ad, None, weff[0], 4 * F * F)], z, dc, bias=bd, relu=True)
    assert rel_err(z, ref) <= TOL
    gr = torch.randn(N, F)
    ref_dA = gr.double() @ W[:, F:5 * F].double() + (gr * amp.cpu()[:, None]).double() @ W[:, 5 * F:9 * F].double() + \
        (gr * att.cpu()[:, None]).double() @ W[:, 9 * F:].double()
    dA = torch.full((N, 4 * F), float("nan"), device=gpu_device)
    ops.gemm_grouped([(gr.to(gpu_device), None, weff[0], 4 * F * F)], dA, dc, b_trans=False)
    assert rel_err(dA, ref_dA) <= TOL


def _both_tiled_kernels(fn):
    """fn() through the software-pipelined kernel (default for >= 8 K-tiles) and through the two-barrier kernel."""
    from gnnepcsaft_amd import ops
    outs = []
    for pipe in (1, 0):
        ops.set_option(torch.device("cuda:0"), _lib.OPT_GEMM_PIPE, pipe)
        try:
            outs.append(fn())
        finally:
            ops.set_option(torch.device("cuda:0"), _lib.OPT_GEMM_PIPE, 1)
    return outs


def test_pipelined_and_two_barrier_tiled_kernels_are_bit_identical(gpu_device):
    """k_gemm3p and k_gemm3 issue the same MFMAs on the same operands in the same order per accumulator (only the
    staging differs: B fragments straight from a fragment-major image, one barrier per K-tile): bit-identical outputs
    for every epilogue, ragged shapes, several column tiles, strided views and degree-class grouping -- and both within
    1e-5 of fp64."""
    from gnnepcsaft_amd import ops
    torch.manual_seed(21)
    F = 128
    # (a) post-layer-0 shape, two segments, bias + ReLU; ragged M
    M = 20037
    x, A = torch.randn(M, F), torch.randn(M, 4 * F)
    W, b = torch.randn(F, 5 * F) / 8, torch.randn(F)
    xd, Ad, Wd, bd = x.to(gpu_device), A.to(gpu_device), W.to(gpu_device), b.to(gpu_device)

    def post0():
        out = torch.full((M, F), float("nan"), device=gpu_device)
        ops.gemm([(xd, None, Wd[:, :F]), (Ad, None, Wd[:, F:])], out, bias=bd, relu=True)
        return out
    p, q = _both_tiled_kernels(post0)
    assert torch.equal(p, q)
    assert rel_err(p, (torch.cat([x, A], 1).double() @ W.double().T + b.double()).relu()) <= TOL
    # (b) three NN segments with k not a multiple of 32 (4 + 4 + 5 K-tiles), strided operand / output views, accumulate
    M2 = 9001
    big = torch.randn(M2, 3 * 160)
    ks = (100, 128, 156)
    ws = [torch.randn(k, 96) / 4 for k in ks]
    c0 = torch.randn(M2, 200)
    bigd, wsd = big.to(gpu_device), [w.to(gpu_device) for w in ws]

    def three():
        outw = c0.clone().to(gpu_device)
        ops.gemm([(bigd[:, 160 * i:160 * i + k], None, wsd[i]) for i, k in enumerate(ks)], outw[:, 8:104], b_trans=False,
                 accumulate=True)
        return outw
    p, q = _both_tiled_kernels(three)
    assert torch.equal(p, q)
    ref = c0.double().clone()
    ref[:, 8:104] += sum(big[:, 160 * i:160 * i + k].double() @ ws[i].double() for i, k in enumerate(ks))
    assert rel_err(p, ref) <= TOL
    # (c) two column tiles, K = 400 (13 K-tiles), ReLU-mask epilogue
    M3, N3, K3 = 33000, 256, 400
    a, w, mask = torch.randn(M3, K3), torch.randn(K3, N3) / 4, torch.randn(M3, N3)
    ad, wd, md = a.to(gpu_device), w.to(gpu_device), mask.to(gpu_device)

    def wide():
        out = torch.full((M3, N3), float("nan"), device=gpu_device)
        ops.gemm([(ad, None, wd)], out, b_trans=False, mask=md)
        return out
    p, q = _both_tiled_kernels(wide)
    assert torch.equal(p, q)
    assert rel_err(p, (a.double() @ w.double()) * (mask > 0)) <= TOL
    # (d) rows gathered by in-degree class with per-class weights (K = 128 + 512)
    rng = np.random.default_rng(5)
    N, E = 30000, 70000
    g = _pack(_graph(rng, N, E), None, None, N, None, gpu_device)
    dc = g.degree_classes()
    assert dc is not None
    xg, Ag = torch.randn(N, F, device=gpu_device), torch.randn(N, 4 * F, device=gpu_device)
    Wp = torch.randn(F, 13 * F, device=gpu_device) / 8
    weff = ops.pna_weff(Wp, F, dc.D, 1.2)

    def grouped():
        z = torch.full((N, F), float("nan"), device=gpu_device)
        ops.gemm_grouped([(xg, None, Wp[:, 0:F], 0), (Ag, None, weff[0], 4 * F * F)], z, dc, bias=bd, relu=True)
        return z
    p, q = _both_tiled_kernels(grouped)
    assert torch.equal(p, q)
    assert not torch.isnan(p).any()


def test_activation_stationary_and_two_barrier_kernels_are_bit_identical(gpu_device):
    """k_gemm_as3 (one segment, 96 < K <= 128, N a multiple of 128 >= 256: the row tile is split once for all column
    tiles; a partial tile repeats its last row instead of predicating) against k_gemm3 (GNX_OPT_GEMM_AS = 0): same MFMAs
    per accumulator -> bit-identical, for every epilogue, a ragged last tile, K below 128, a strided output view that
    must stay untouched outside, and the degree-class grouped dA product."""
    from gnnepcsaft_amd import ops
    dev = torch.device("cuda:0")

    def both(fn):
        outs = []
        for on in (1, 0):
            ops.set_option(dev, _lib.OPT_GEMM_AS, on)
            try:
                outs.append(fn())
            finally:
                ops.set_option(dev, _lib.OPT_GEMM_AS, 1)
        return outs

    torch.manual_seed(77)
    for M, N, K in [(8200, 512, 128), (4099, 256, 100), (70001, 384, 128)]:
        a, w = torch.randn(M, K), torch.randn(K, N) / 4
        mask, c0, b = torch.randn(M, N), torch.randn(M, N), torch.randn(N)
        ad, wd, md, bd = a.to(gpu_device), w.to(gpu_device), mask.to(gpu_device), b.to(gpu_device)
        prod = a.double() @ w.double()

        def plain():
            out = torch.full((M, N + 24), float("nan"), device=gpu_device)
            ops.gemm([(ad, None, wd)], out[:, 8:8 + N], b_trans=False)
            return out
        p, q = both(plain)
        assert torch.equal(p[:, 8:8 + N], q[:, 8:8 + N]) and rel_err(p[:, 8:8 + N], prod) <= TOL
        assert torch.isnan(p[:, :8]).all() and torch.isnan(p[:, 8 + N:]).all()

        def masked():
            out = torch.full((M, N), float("nan"), device=gpu_device)
            ops.gemm([(ad, None, wd)], out, b_trans=False, mask=md)
            return out
        p, q = both(masked)
        assert torch.equal(p, q) and rel_err(p, prod * (mask > 0)) <= TOL

        def accum():
            out = c0.clone().to(gpu_device)
            ops.gemm([(ad, None, wd)], out, b_trans=False, accumulate=True)
            return out
        p, q = both(accum)
        assert torch.equal(p, q) and rel_err(p, c0.double() + prod) <= TOL

        def bias_relu():
            out = torch.full((M, N), float("nan"), device=gpu_device)
            ops.gemm([(ad, None, wd.T.contiguous())], out, bias=bd, relu=True)
            return out
        p, q = both(bias_relu)
        assert torch.equal(p, q) and rel_err(p, (prod + b.double()).relu()) <= TOL
    # grouped dA: rows gathered per in-degree class, per-class weights, K = 128 -> N = 512
    rng = np.random.default_rng(8)
    Nn, E, F = 50011, 120000, 128
    g = _pack(_graph(rng, Nn, E), None, None, Nn, None, gpu_device)
    dc = g.degree_classes()
    assert dc is not None
    Wp = torch.randn(F, 13 * F, device=gpu_device) / 8
    weff = ops.pna_weff(Wp, F, dc.D, 1.2)
    gr = torch.randn(Nn, F, device=gpu_device)

    def grouped():
        dA = torch.full((Nn, 4 * F), float("nan"), device=gpu_device)
        ops.gemm_grouped([(gr, None, weff[0], 4 * F * F)], dA, dc, b_trans=False)
        return dA
    p, q = both(grouped)
    assert torch.equal(p, q) and not torch.isnan(p).any()


def test_96_and_128_row_tiles_of_the_pipelined_kernel_are_bit_identical(gpu_device):
    """k_gemm3p<.., 3> (96-row tiles: taken when 128-row tiles leave the last round of persistent workgroups mostly
    idle) against k_gemm3p<.., 4>: the same arithmetic per row -> bit-identical, plain rows (GNX_OPT_GEMM_TILE_ROWS
    96 / 128) and the degree-class table built with 96 rows against the 128-row one; gnx_gemm_tile_rows' choice."""
    from gnnepcsaft_amd import ops
    dev = torch.device("cuda:0")
    lib, h = _lib.load(), ops.handle(dev)
    assert lib.gnx_gemm_tile_rows(h, 81920, 128) == 96      # 640 tiles on 512 slots -> 854 tiles of 96
    assert lib.gnx_gemm_tile_rows(h, 65536, 128) == 128     # exactly one round of 512
    assert lib.gnx_gemm_tile_rows(h, 327680, 128) == 128    # a tie keeps 128
    torch.manual_seed(9)
    F = 128
    for M in (81920, 20037):
        x, A = torch.randn(M, F, device=gpu_device), torch.randn(M, 4 * F, device=gpu_device)
        W, b = torch.randn(F, 5 * F, device=gpu_device) / 8, torch.randn(F, device=gpu_device)
        outs = []
        for rows in (96, 128):
            ops.set_option(dev, _lib.OPT_GEMM_TILE_ROWS, rows)
            try:
                out = torch.full((M, F), float("nan"), device=gpu_device)
                ops.gemm([(x, None, W[:, :F]), (A, None, W[:, F:])], out, bias=b, relu=True)
                outs.append(out)
            finally:
                ops.set_option(dev, _lib.OPT_GEMM_TILE_ROWS, 0)
        assert torch.equal(outs[0], outs[1]) and not torch.isnan(outs[0]).any()
        ref = (torch.cat([x, A], 1).double() @ W.double().T + b.double()).relu()
        assert rel_err(outs[0], ref) <= TOL
    # grouped: 81 920 rows -> the forward table has 96-row tiles
    rng = np.random.default_rng(15)
    N, E = 81920, 163840
    g = _pack(_graph(rng, N, E), None, None, N, None, gpu_device)
    dc = g.degree_classes()
    assert dc is not None and dc.tile_rows_p == 96 and dc.tiles_p is not dc.tiles
    xg, Ag = torch.randn(N, F, device=gpu_device), torch.randn(N, 4 * F, device=gpu_device)
    Wp = torch.randn(F, 13 * F, device=gpu_device) / 8
    bd = torch.randn(F, device=gpu_device)
    weff = ops.pna_weff(Wp, F, dc.D, 1.2)
    zs = []
    for fwd in (True, False):
        z = torch.full((N, F), float("nan"), device=gpu_device)
        ops.gemm_grouped([(xg, None, Wp[:, 0:F], 0), (Ag, None, weff[0], 4 * F * F)], z, dc, bias=bd, relu=True, forward_tiles=fwd)
        zs.append(z)
    assert torch.equal(zs[0], zs[1]) and not torch.isnan(zs[0]).any()


def test_split_and_exact_kernels_agree(gpu_device):
    """Same call through the split-operand kernel and the exact-fp32 MFMA kernel: both within 1e-5 of fp64 and within
    2e-6 (norm-wise) of each other."""
    from gnnepcsaft_amd import ops
    torch.manual_seed(9)
    M, F = 12345, 128
    x, A = torch.randn(M, F), torch.randn(M, 4 * F)
    W = torch.randn(F, 5 * F) / 8
    xd, Ad, Wd = x.to(gpu_device), A.to(gpu_device), W.to(gpu_device)
    ref = torch.cat([x, A], 1).double() @ W.double().T
    outs = {}
    for mode in ("1", "0"):
        ops.set_option(torch.device("cuda:0"), _lib.OPT_GEMM_SPLIT, int(mode))
        try:
            out = torch.empty(M, F, device=gpu_device)
            ops.gemm([(xd, None, Wd[:, :F]), (Ad, None, Wd[:, F:])], out)
            outs[mode] = out.double().cpu()
        finally:
            ops.set_option(torch.device("cuda:0"), _lib.OPT_GEMM_SPLIT, 1)
    assert rel_err(outs["1"], ref) <= TOL and rel_err(outs["0"], ref) <= TOL
    assert rel_err(outs["1"], outs["0"]) <= 2e-6
    assert rel_err(outs["1"], ref) <= rel_err(outs["0"], ref) * 1.5 + 1e-8


@pytest.mark.parametrize("M,N,K", [(20001, 128, 128), (8192, 36, 100), (5000, 130, 260)])
def test_split_weight_gradient_single(gpu_device, wgrad_kernel, M, N, K):
    """dW += dC^T A and db += column sums on the split-operand kernel (no row scale, M >= 4096): ragged shapes, strided
    views, accumulation into existing values."""
    from gnnepcsaft_amd import ops
    torch.manual_seed(M + N)
    wide_dc, wide_a = torch.randn(M, N + 8), torch.randn(M, K + 12)
    dc, a = wide_dc[:, 4:4 + N], wide_a[:, 8:8 + K]
    dcd, ad = wide_dc.to(gpu_device)[:, 4:4 + N], wide_a.to(gpu_device)[:, 8:8 + K]
    dw = torch.full((N, K), 2.0, device=gpu_device)
    db = torch.full((N,), -1.0, device=gpu_device)
    ops.gemm_wgrad(dcd, ad, dw, dbias=db)
    assert rel_err(dw, 2.0 + dc.double().T @ a.double()) <= TOL
    assert rel_err(db, -1.0 + dc.double().sum(0)) <= TOL


def test_split_weight_gradient_batched(gpu_device, wgrad_kernel):
    """A layer's worth of independent weight gradients in one launch (different M, shared operands)."""
    from gnnepcsaft_amd import ops
    torch.manual_seed(12)
    F = 128
    Mn, Me = 9000, 18000
    g = [torch.randn(Mn, F) for _ in range(4)] + [torch.randn(Me, F)]
    x = [torch.randn(Mn, F) for _ in range(4)] + [torch.randn(Me, F)]
    dws = [torch.zeros(F, F, device=gpu_device) for _ in range(5)]
    dbs = [torch.zeros(F, device=gpu_device) for _ in range(5)]
    gd = [t.to(gpu_device) for t in g]
    xd = [t.to(gpu_device) for t in x]
    for i in range(5):
        ops.queue_wgrad(gd[i], xd[i], dws[i], dbias=dbs[i] if i % 2 == 0 else None)
    ops.queue_wgrad(gd[0], xd[1], dws[1])  # a second contribution to the same dW
    ops.flush_wgrads()
    ops.join_side_stream(gpu_device)
    for i in range(5):
        ref = g[i].double().T @ x[i].double()
        if i == 1:
            ref = ref + g[0].double().T @ x[1].double()
        assert rel_err(dws[i], ref) <= TOL, i
        if i % 2 == 0:
            assert rel_err(dbs[i], g[i].double().sum(0)) <= TOL, i
        else:
            assert float(dbs[i].abs().max()) == 0.0


def test_split_weight_gradient_by_degree_class(gpu_device, wgrad_kernel):
    """Post-layer-0 weight gradient through per-degree-class partial sums (gathered rows, K = 4F wide)."""
    from gnnepcsaft_amd import ops
    rng = np.random.default_rng(41)
    N, E, F = 60000, 150000, 64
    ei = _graph(rng, N, E)
    g = _pack(ei, None, None, N, None, gpu_device)
    dc = g.degree_classes()
    assert dc is not None
    avg = 1.15
    amp, att = g.degree_scalers(avg)
    torch.manual_seed(8)
    gr, A = torch.randn(N, F), torch.randn(N, 4 * F)
    cat = torch.cat([torch.zeros(N, F), A, A * amp.cpu()[:, None], A * att.cpu()[:, None]], 1).double()
    ref = gr.double().T @ cat
    dW = torch.zeros(F, 13 * F, device=gpu_device)
    ops.pna_post0_wgrad_classes(gr.to(gpu_device), A.to(gpu_device), dc, F, avg, dW)
    ops.join_side_stream(gpu_device)
    assert rel_err(dW[:, F:], ref[:, F:]) <= TOL
    assert float(dW[:, :F].abs().max()) == 0.0


@pytest.mark.parametrize("K,nseg", [(640, 2), (1664, 4), (128, 1)])
@pytest.mark.parametrize("kind", ["cancelling", "denormal_range", "mixed_magnitude"])
def test_split_product_is_fp32_faithful_at_layer_widths(gpu_device, K, nseg, kind):
    """the three-bf16-piece product beyond K = 128 -- post-layer 0's K = 640 (5F) and the reference
    formulation's K = 1664 (13F), as multi-segment products -- on the inputs that stress it:
      * cancelling: rows whose terms alternate in sign and nearly cancel (|sum| << sum |a||b|), where dropping
        correction terms or absorbing them in the leading accumulator would show;
      * denormal_range: operands around 1e-30 .. 1e-38 (third bf16 pieces and some products underflow);
      * mixed_magnitude: full 24-bit significands spread over 12 orders of magnitude.
    Error against fp64, normalised by sum_k |a||b| (what fp32 rounding scales with), must stay at the level of the
    exact-fp32 MFMA kernel; for the denormal case the normaliser is floored at the smallest normal fp32."""
    from gnnepcsaft_amd import ops
    torch.manual_seed(100 + K)
    M, N = 8192, 128
    a = torch.randn(M, K)
    w = torch.randn(N, K) / 8
    if kind == "cancelling":
        base = torch.randn(M, K // 2)
        a = torch.stack([base, -base * (1 + 1e-4 * torch.randn(M, K // 2))], dim=2).reshape(M, K)
        w = torch.repeat_interleave(torch.randn(N, K // 2), 2, dim=1) * (1 + 1e-6 * torch.randn(N, K))
    elif kind == "denormal_range":
        a = a * 1e-19 * torch.exp(2 * torch.randn(M, K))
        w = w * 1e-15
    else:
        a = a * torch.exp(4 * torch.randn(M, K))
    ad, wd = a.to(gpu_device), w.to(gpu_device)
    ref = a.double() @ w.double().T
    norm = (a.double().abs() @ w.double().abs().T).clamp_min(1.1754943508222875e-38)
    step = K // nseg
    segs = [(ad[:, i * step:(i + 1) * step], None, wd[:, i * step:(i + 1) * step]) for i in range(nseg)]
    errs = {}
    for mode in (1, 0):
        ops.set_option(gpu_device, _lib.OPT_GEMM_SPLIT, mode)
        try:
            out = torch.empty(M, N, device=gpu_device)
            ops.gemm(segs, out)
            errs[mode] = float(((out.double().cpu() - ref).abs() / norm).max())
        finally:
            ops.set_option(gpu_device, _lib.OPT_GEMM_SPLIT, 1)
    eps32 = 2.0 ** -24
    print(K, kind, {k: v / eps32 for k, v in errs.items()})
    # the exact-fp32 MFMA chain: error grows ~ sqrt(K) roundings; the split product sums 16 exact products per rounding
    assert errs[0] <= (8 + 2 * K ** 0.5) * eps32
    assert errs[1] <= max(8 * eps32, 1.25 * errs[0])
