"""Randomised shape sweep of gnx_gemm / gnx_gemm_wgrad through every dispatch path against fp64 (not part of the
suite; a one-off robustness check).  Prints the worst error per path and exits non-zero above 1e-5."""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnepcsaft_amd import _lib, ops
from tests.parity_util import rel_err
dev = torch.device("cuda:0")
random.seed(int(os.environ.get("SEED", "7")))
torch.manual_seed(1)
worst, bad = {}, 0
for case in range(int(os.environ.get("CASES", "120"))):
    M = random.choice([1, 37, 255, 256, 257, 1000, 4095, 4096, 4097, 8191, 8192, 8193, 12345, 20001, 33333, 70001])
    odd = os.environ.get("ODD") == "1"
    N = random.randint(1, 300) if odd else 4 * random.randint(1, 80)
    nseg = random.choice([1, 1, 2, 3])
    kmax = 170 if random.random() < 0.3 else 70   # up to 3 x 680: well past the 12 K-tiles of the pipelined tiled kernel
    ks = [random.randint(1, 4 * kmax) if odd else 4 * random.randint(1, kmax) for _ in range(nseg)]
    bt = random.random() < 0.5
    epi = random.choice(["plain", "relu", "mask", "accum"])
    lda_pad = random.choice([0, 1, 3, 4, 64]) if odd else random.choice([0, 4, 64])
    As = [torch.randn(M, k + lda_pad) for k in ks]
    Ws = [torch.randn(N, k) if bt else torch.randn(k, N) for k in ks]
    bias = torch.randn(N) if random.random() < 0.7 else None
    ref = sum(a[:, :k].double() @ (w.double().T if bt else w.double()) for a, w, k in zip(As, Ws, ks))
    if bias is not None:
        ref = ref + bias.double()
    out = torch.randn(M, N + 8)
    c0 = out[:, 4:4 + N].clone()
    mask = torch.randn(M, N)
    outd = out.to(dev)
    segs = [(a.to(dev)[:, :k], None, w.to(dev)) for a, w, k in zip(As, Ws, ks)]
    kw = dict(bias=None if bias is None else bias.to(dev), b_trans=bt)
    if epi == "relu":
        ref = ref.relu(); kw["relu"] = True
    elif epi == "mask":
        ref = ref * (mask > 0); kw["mask"] = mask.to(dev)
    elif epi == "accum":
        ref = ref + c0.double(); kw["accumulate"] = True
    ops.gemm(segs, outd[:, 4:4 + N], **kw)
    e = rel_err(outd[:, 4:4 + N], ref)
    if not odd:  # the two tiled split kernels must agree bit for bit
        out2 = out.to(dev)
        ops.set_option(dev, _lib.OPT_GEMM_PIPE, 0)
        ops.gemm(segs, out2[:, 4:4 + N], **kw)
        ops.set_option(dev, _lib.OPT_GEMM_PIPE, 1)
        if not torch.equal(out2, outd):
            bad += 1
            print("FAIL pipelined != two-barrier", dict(M=M, N=N, ks=ks, epi=epi, bt=bt), flush=True)
    untouched = torch.equal(outd[:, :4].cpu(), out[:, :4]) and torch.equal(outd[:, 4 + N:].cpu(), out[:, 4 + N:])
    key = f"gemm nseg={nseg} {'NT' if bt else 'NN'} {epi}"
    worst[key] = max(worst.get(key, 0.0), e)
    if e > 1e-5 or not untouched:
        bad += 1
        print("FAIL", key, dict(M=M, N=N, ks=ks, lda_pad=lda_pad), e, untouched, flush=True)
    # weight gradient of the first segment
    g = torch.randn(M, N)
    dw = torch.zeros(N, ks[0], device=dev); db = torch.zeros(N, device=dev)
    few = random.random() < 0.5   # few workgroups -> long row ranges -> the wave-specialised kernel (>= 512 rows each)
    if few: ops.set_option(dev, _lib.OPT_WGRAD_WGS, random.choice([4, 16, 32]))
    ops.gemm_wgrad(g.to(dev), As[0].to(dev)[:, :ks[0]], dw, dbias=db)
    if few: ops.set_option(dev, _lib.OPT_WGRAD_WGS, 0)
    e = max(rel_err(dw, g.double().T @ As[0][:, :ks[0]].double()), rel_err(db, g.double().sum(0)))
    worst["wgrad"] = max(worst.get("wgrad", 0.0), e)
    if e > 1e-5:
        bad += 1
        print("FAIL wgrad", dict(M=M, N=N, K=ks[0]), e, flush=True)
ops.join_side_stream(dev)
for k in sorted(worst): print(f"{k:28s} worst rel err {worst[k]:.2e}")
print("failures:", bad)
sys.exit(1 if bad else 0)
