"""A/B timing of the weight-gradient kernels: wave-specialised split kernel (default, mode 2), two-barrier split kernel
(mode 1: GNX_OPT_WGRAD_PIPE = 0), fp32 MFMA (mode 0: GNX_OPT_GEMM_SPLIT = 0)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnepcsaft_amd import _lib, ops
dev = torch.device("cuda:0")
ops.set_wgrad_side_stream(False)
for M, N, K, nprob in ((81920, 128, 128, 1), (81920, 128, 128, 8), (163840, 128, 128, 1), (81920, 128, 512, 1)):
    g = [torch.randn(M, N, device=dev) for _ in range(nprob)]
    x = [torch.randn(M, K, device=dev) for _ in range(nprob)]
    dw = [torch.zeros(N, K, device=dev) for _ in range(nprob)]
    def run():
        for i in range(nprob):
            ops.queue_wgrad(g[i], x[i], dw[i])
        ops.flush_wgrads()
    for mode in os.environ.get("DIAG_MODES", "2,1,0").split(","):
        ops.set_option(torch.device("cuda:0"), _lib.OPT_GEMM_SPLIT, 1 if int(mode) else 0)
        ops.set_option(torch.device("cuda:0"), _lib.OPT_WGRAD_PIPE, 1 if int(mode) == 2 else 0)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        byts = nprob * M * (N + K) * 4
        print(f"M={M} N={N} K={K} x{nprob} split={mode}: {us:7.1f} us  {byts/us/1e6:5.2f} TB/s  {2*nprob*M*N*K/us/1e6:6.1f} TF/s", flush=True)
