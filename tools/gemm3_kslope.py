"""Per-K-tile cost of the tiled product: time vs K at fixed M, N=128; the slope is what one more 128 x 128 x 32 K-tile costs
a CU (fixed launch / prologue / epilogue cost drops out).  DIAG_MODES: 0 = fp32 MFMA kernel, 1 = split operands through the
two-barrier kernel, 2 = split operands through the software-pipelined kernel (>= 12 K-tiles).
M=65536: K=256 / 512 keep A inside the Infinity Cache, K=1024 streams it from HBM."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnepcsaft_amd import _lib, ops
dev = torch.device("cuda:0")
M, F = int(os.environ.get("DIAG_M", "65536")), 128
for mode in os.environ.get("DIAG_MODES", "2,1,0").split(","):
    mode = int(mode)
    ops.set_option(dev, _lib.OPT_GEMM_SPLIT, 1 if mode else 0)
    ops.set_option(dev, _lib.OPT_GEMM_PIPE, 1 if mode == 2 else 0)
    res = []
    for K in (384, 768, 1536):
        A = torch.randn(M, K, device=dev); W = torch.randn(F, K, device=dev); z = torch.empty(M, F, device=dev)
        fn = lambda: ops.gemm([(A, None, W)], z, relu=True)
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / 20)
    tiles_per_cu = M / 128 / 256
    print(f"mode {mode}: K=384 {res[0]:.1f} us  K=768 {res[1]:.1f} us  K=1536 {res[2]:.1f} us   "
          f"per K-tile per CU: {(res[2] - res[1]) / (24 * tiles_per_cu):.3f} us (768->1536), {(res[1] - res[0]) / (12 * tiles_per_cu):.3f} us (384->768)", flush=True)
