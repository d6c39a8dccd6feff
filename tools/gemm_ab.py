"""A/B timing of the weights-stationary GEMM: split-bf16 (default) vs exact fp32 MFMA (GNX_GEMM_SPLIT=0)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnepcsaft_amd import _lib, ops
dev = torch.device("cuda:0")
for M in (81920, 163840):
    a = torch.randn(M, 128, device=dev); w = torch.randn(128, 128, device=dev); out = torch.empty(M, 128, device=dev)
    mask = torch.randn(M, 128, device=dev)
    for bt, mk in ((True, None), (False, None), (True, mask), (False, mask)):
        for mode in ("1", "0"):
            ops.set_option(torch.device("cuda:0"), _lib.OPT_GEMM_SPLIT, int(mode))
            for _ in range(5): ops.gemm([(a, None, w)], out, b_trans=bt, mask=mk)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): ops.gemm([(a, None, w)], out, b_trans=bt, mask=mk)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 50
            byts = M * 128 * 4 * (3 if mk is not None else 2)
            print(f"M={M} bt={bt} mask={mk is not None} split={mode}: {us:7.1f} us  {byts/us/1e6:6.2f} TB/s  {2*M*128*128/us/1e6:6.1f} TF/s", flush=True)
