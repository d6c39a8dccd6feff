"""Timing of the tiled multi-segment GEMM on the three PNA shapes that use it (cfg-2 sizes; DIAG_M = rows).
DIAG_MODES: 2 = split operands, software-pipelined kernel where eligible (default build), 1 = split operands, two-barrier
kernel (GNX_OPT_GEMM_PIPE = 0), 0 = fp32 MFMA (GNX_OPT_GEMM_SPLIT = 0)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnepcsaft_amd import _lib, ops
dev = torch.device("cuda:0")
M, F = int(os.environ.get("DIAG_M", "81920")), 128
x = torch.randn(M, F, device=dev); A = torch.randn(M, 4 * F, device=dev)
g = torch.randn(M, F, device=dev); dP = torch.randn(M, F, device=dev); dQ = torch.randn(M, F, device=dev)
Wp = torch.randn(F, 13 * F, device=dev); W0 = torch.randn(F, 3 * F, device=dev)
Weff = torch.randn(F, 4 * F, device=dev)
z = torch.empty(M, F, device=dev); dA = torch.empty(M, 4 * F, device=dev)
cases = {
    "post0 NT K=640 N=128": (lambda: ops.gemm([(x, None, Wp[:, :F]), (A, None, Wp[:, F:5 * F])], z, relu=True), (M * (5 * F + F)) * 4, 2 * M * 5 * F * F),
    "dA    NN K=128 N=512": (lambda: ops.gemm([(g, None, Weff)], dA, b_trans=False), (M * 5 * F) * 4, 2 * M * 4 * F * F),
    "dx    NN K=384 N=128": (lambda: ops.gemm([(g, None, Wp[:, :F]), (dP, None, W0[:, :F]), (dQ, None, W0[:, F:2 * F])], z, b_trans=False), (M * 4 * F) * 4, 2 * M * 3 * F * F),
}
for name, (fn, byts, flops) in cases.items():
    for mode in (os.environ.get("DIAG_MODES", "2,1,0").split(",")):
        ops.set_option(torch.device("cuda:0"), _lib.OPT_GEMM_SPLIT, 1 if int(mode) else 0)
        ops.set_option(torch.device("cuda:0"), _lib.OPT_GEMM_PIPE, 1 if int(mode) == 2 else 0)
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        print(f"{name} split={mode}: {us:7.1f} us  {byts/us/1e6:5.2f} TB/s  {flops/us/1e6:6.1f} TF/s", flush=True)
