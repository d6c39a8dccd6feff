"""Launch-to-launch time of the weights-stationary split GEMM vs row count (fixed overhead vs streaming slope)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnepcsaft_amd import _lib, ops
dev = torch.device("cuda:0")
if len(sys.argv) > 1:  # GNX_OPT_GEMM_WS_FAST: 0 = predicated epilogue, 1 = predicate-free 8-wave form, 2 (default) = 4-wave form
    ops.set_option(dev, _lib.OPT_GEMM_WS_FAST, int(sys.argv[1]))
    print("GNX_OPT_GEMM_WS_FAST =", sys.argv[1])
w = torch.randn(128, 128, device=dev)
for M in (8192, 16384, 32768, 65536, 81920, 163840, 327680):
    a = torch.randn(M, 128, device=dev); out = torch.empty(M, 128, device=dev)
    for bt in (True, False):
        for _ in range(5): ops.gemm([(a, None, w)], out, b_trans=bt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): ops.gemm([(a, None, w)], out, b_trans=bt)
        e1.record(); torch.cuda.synchronize()
        print(f"M={M:7d} bt={bt}: {e0.elapsed_time(e1)*1e3/50:6.1f} us", flush=True)
