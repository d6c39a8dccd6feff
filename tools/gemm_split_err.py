"""Error of the split-bf16 vs exact-fp32 weights-stationary GEMM against fp64 (normalised by sum |a||b|)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnepcsaft_amd import _lib, ops
dev = torch.device("cuda:0")
torch.manual_seed(11)
M, N, K = 16384, 128, 128
for bt in (True, False):
    a = torch.randn(M, K) * torch.exp(torch.randn(M, K)); w = torch.randn(N, K) if bt else torch.randn(K, N)
    wm = w.double().T if bt else w.double()
    ref = a.double() @ wm; norm = a.double().abs() @ wm.abs()
    for mode in ("1", "0"):
        ops.set_option(torch.device("cuda:0"), _lib.OPT_GEMM_SPLIT, int(mode))
        out = torch.empty(M, N, device=dev)
        ops.gemm([(a.to(dev), None, w.to(dev))], out, b_trans=bt)
        e = (out.double().cpu() - ref).abs() / norm
        print(f"bt={bt} split={mode}: max {float(e.max())/2**-24:.3f} ulp  rms {float((e**2).mean().sqrt())/2**-24:.4f} ulp", flush=True)
