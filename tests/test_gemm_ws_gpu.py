"""GPU parity of the weights-stationary persistent GEMM (k_gemm_ws: one segment, K,N <= 128, M >= 8192) against fp64
torch, all epilogue variants, both weight layouts, ragged M / K / N; and against the tiled kernel (GNX_GEMM_WS=0)."""
import os

import pytest
import torch

from gnnepcsaft_amd import _lib
from tests.parity_util import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize("M,N,K", [(8192, 128, 128), (20001, 128, 128), (81920, 128, 128), (9000, 64, 96), (8200, 36, 32),
                                   (163840, 128, 128), (10000, 128, 100)])
@pytest.mark.parametrize("b_trans", [True, False])
def test_gemm_ws_variants(gpu_device, M, N, K, b_trans):
    from gnnepcsaft_amd import ops
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K)
    w = torch.randn(N, K) if b_trans else torch.randn(K, N)
    b = torch.randn(N)
    mask = torch.randn(M, N)
    c0 = torch.randn(M, N)
    ad, wd, bd = a.to(gpu_device), w.to(gpu_device), b.to(gpu_device)
    prod = a.double() @ (w.double().T if b_trans else w.double())
    out = torch.full((M, N), float("nan"), device=gpu_device)
    ops.gemm([(ad, None, wd)], out, bias=bd, relu=True, b_trans=b_trans)
    assert rel_err(out, (prod + b.double()).relu()) <= TOL
    out = torch.full((M, N), float("nan"), device=gpu_device)
    ops.gemm([(ad, None, wd)], out, b_trans=b_trans, mask=mask.to(gpu_device))
    assert rel_err(out, prod * (mask > 0)) <= TOL
    out = c0.clone().to(gpu_device)
    ops.gemm([(ad, None, wd)], out, b_trans=b_trans, accumulate=True, bias=bd)
    assert rel_err(out, c0.double() + prod + b.double()) <= TOL


@pytest.mark.parametrize("M,K", [(8192, 128), (20001, 128), (81920 + 37, 128), (10000, 100)])
@pytest.mark.parametrize("b_trans", [True, False])
def test_predicate_free_and_predicated_ws3_are_bit_identical(gpu_device, M, K, b_trans):
    """k_gemm_ws3<.., FAST> (N = 128: quad-transposed 16-byte stores, a partial last tile repeats row M - 1, exact
    waits) against the predicated form (GNX_OPT_GEMM_WS_FAST = 0): same MFMAs -> bit-identical outputs for the plain /
    bias + ReLU / mask epilogues, a ragged last tile, K below 128, and a strided output view that stays untouched outside."""
    from gnnepcsaft_amd import ops
    dev = torch.device("cuda:0")
    N = 128
    torch.manual_seed(M + K)
    a = torch.randn(M, K, device=gpu_device)
    w = torch.randn(N, K, device=gpu_device) if b_trans else torch.randn(K, N, device=gpu_device)
    b, mask = torch.randn(N, device=gpu_device), torch.randn(M, N, device=gpu_device)

    def both(fn):
        outs = []
        for on in (1, 0, 2):  # 2: the 4-wave form (32-row tiles, two workgroups per CU)
            ops.set_option(dev, _lib.OPT_GEMM_WS_FAST, on)
            try:
                outs.append(fn())
            finally:
                ops.set_option(dev, _lib.OPT_GEMM_WS_FAST, 2)
        assert torch.equal(torch.nan_to_num(outs[2], nan=-7.0), torch.nan_to_num(outs[0], nan=-7.0))  # (NaN guard frame)
        return outs[:2]

    def bias_relu():
        out = torch.full((M + 3, N + 8), float("nan"), device=gpu_device)
        ops.gemm([(a, None, w)], out[:M, 4:4 + N], bias=b, relu=True, b_trans=b_trans)
        return out
    p, q = both(bias_relu)
    assert torch.equal(p[:M, 4:4 + N], q[:M, 4:4 + N]) and not torch.isnan(p[:M, 4:4 + N]).any()
    assert torch.isnan(p[M:]).all() and torch.isnan(p[:, :4]).all() and torch.isnan(p[:, 4 + N:]).all()

    def plain():
        out = torch.full((M, N), float("nan"), device=gpu_device)
        ops.gemm([(a, None, w)], out, b_trans=b_trans)
        return out
    p, q = both(plain)
    assert torch.equal(p, q)

    def masked():
        out = torch.full((M, N), float("nan"), device=gpu_device)
        ops.gemm([(a, None, w)], out, b_trans=b_trans, mask=mask)
        return out
    p, q = both(masked)
    assert torch.equal(p, q)
    prod = a.double() @ (w.double().T if b_trans else w.double())
    assert rel_err(p, prod * (mask > 0)) <= TOL


def test_gemm_ws_strided_views_and_guard_rows(gpu_device):
    """Tower-style column slices (lda = ldc = H > K) and no write outside the output view."""
    from gnnepcsaft_amd import ops
    torch.manual_seed(3)
    M, H, F = 12345, 256, 64
    x = torch.randn(M, H)
    w = torch.randn(F, 3 * F)
    xd, wd = x.to(gpu_device), w.to(gpu_device)
    out = torch.zeros(M, H, device=gpu_device)
    ops.gemm([(xd[:, F:2 * F], None, wd[:, F:2 * F])], out[:, 2 * F:3 * F])
    ref = x[:, F:2 * F].double() @ w[:, F:2 * F].double().T
    assert rel_err(out[:, 2 * F:3 * F], ref) <= TOL
    assert float(out[:, :2 * F].abs().max()) == 0.0 and float(out[:, 3 * F:].abs().max()) == 0.0


@pytest.mark.parametrize("kernel", [1, 2], ids=["bf16_scattered_one_hot", "fp32_computed_one_hot"])
@pytest.mark.parametrize("N,H", [(20000, 128), (9000, 36), (81920, 128), (640, 128), (300, 64), (4097, 256)])
def test_embed_backward_on_mfma(gpu_device, N, H, kernel):
    """Large-batch AtomEncoder backward = one-hot^T x gradient on the matrix cores (exact 0/1 products): the bf16 kernel
    (one-hot scattered into LDS, gradient as three bf16 pieces) and the fp32-MFMA kernel, ragged row counts, H below /
    above one channel tile."""
    import numpy as np
    from gnnepcsaft_amd import nn as gnn, ops
    ops.set_option(torch.device("cuda:0"), _lib.OPT_EMBED_BWD_MFMA, kernel)
    try:
        _embed_backward_case(gpu_device, N, H)
    finally:
        ops.set_option(torch.device("cuda:0"), _lib.OPT_EMBED_BWD_MFMA, 1)


def _embed_backward_case(gpu_device, N, H):
    import numpy as np
    from gnnepcsaft_amd import nn as gnn
    from oracle import pyg_restatement as O
    torch.manual_seed(1)
    enc_o = O.AtomEncoder(H)
    enc_n = gnn.AtomEncoder(H)
    enc_n.load_state_dict(enc_o.state_dict())
    enc_n.to(gpu_device)
    rng = np.random.default_rng(N)
    x = torch.from_numpy(np.stack([rng.integers(0, d, size=N) for d in O.ATOM_FEATURE_DIMS], 1)).long()
    w = torch.randn(N, H)
    enc_o.double()  # fp64 truth: sums of up to N/2 terms per table row differ by ~1e-5 between two fp32 orders
    (enc_o(x) * w.double()).sum().backward()
    (enc_n(x.to(gpu_device)) * w.to(gpu_device)).sum().backward()
    for eo, en in zip(enc_o.atom_embedding_list, enc_n.atom_embedding_list):
        assert rel_err(en.weight.grad, eo.weight.grad) <= TOL


@pytest.mark.parametrize("b_trans", [True, False])
@pytest.mark.parametrize("scale", [1.0, 1e-3, 37.0])
def test_split_operand_product_is_fp32_faithful(gpu_device, b_trans, scale):
    """The default kernel writes each fp32 operand as three bf16 pieces and issues six bf16 MFMAs (fp32 accumulate).
    Its error against fp64 must be at the level of the exact-fp32 MFMA kernel (GNX_GEMM_SPLIT=0) -- measured as the
    max error over all outputs, normalised by sum_k |a||b| (the quantity fp32 rounding scales with)."""
    from gnnepcsaft_amd import ops
    torch.manual_seed(11)
    M, N, K = 16384, 128, 128
    a = torch.randn(M, K) * scale
    w = torch.randn(N, K) if b_trans else torch.randn(K, N)
    # operands with full 24-bit significands and mixed magnitudes
    a = a * torch.exp(torch.randn(M, K))
    ad, wd = a.to(gpu_device), w.to(gpu_device)
    wm = w.double().T if b_trans else w.double()
    ref = a.double() @ wm
    norm = a.double().abs() @ wm.abs()
    errs = {}
    for mode in ("1", "0"):
        ops.set_option(torch.device("cuda:0"), _lib.OPT_GEMM_SPLIT, int(mode))
        try:
            out = torch.empty(M, N, device=gpu_device)
            ops.gemm([(ad, None, wd)], out, b_trans=b_trans)
            errs[mode] = float(((out.double().cpu() - ref).abs() / norm).max())
        finally:
            ops.set_option(torch.device("cuda:0"), _lib.OPT_GEMM_SPLIT, 1)
    eps32 = 2.0 ** -24
    # measured: exact-fp32 MFMA (sequential k chain) ~10 ulp max / 0.7 rms; split product ~4 ulp max / 0.27 rms
    assert errs["0"] <= 24 * eps32
    assert errs["1"] <= 8 * eps32
    assert errs["1"] <= errs["0"] + eps32
