"""Host-side cost of one launch through the Python wrappers (the step has ~330 launches; phases made of tiny kernels are
host-bound).  Prints microseconds of host time per call for a few wrappers, GPU running ahead-of-time asynchronously."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnepcsaft_amd import _lib, ops
dev = torch.device("cuda:0")
a = torch.randn(60, 128, device=dev); w = torch.randn(128, 128, device=dev); out = torch.empty(60, 128, device=dev)
bias = torch.randn(128, device=dev)
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = time.perf_counter() - t0; torch.cuda.synchronize()
    return dt / n * 1e6
print("ops.gemm small       ", round(t(lambda: ops.gemm([(a, None, w)], out, bias=bias)), 2), "us")
print("ops.zeros(60,128)    ", round(t(lambda: ops.zeros(60, 128, device=dev)), 2), "us")
print("torch.empty(60,128)  ", round(t(lambda: torch.empty(60, 128, device=dev)), 2), "us")
lib, h = _lib.load(), _lib.handle(dev)
print("_lib.handle(dev)     ", round(t(lambda: _lib.handle(dev)), 2), "us")
print("raw gnx_fill         ", round(t(lambda: lib.gnx_fill(h, out.data_ptr(), out.numel(), 0.0)), 2), "us")
print("data_ptr()           ", round(t(lambda: out.data_ptr()), 2), "us")
print("ops._mat             ", round(t(lambda: ops._mat(out, 'x')), 2), "us")
big = torch.randn(81920, 128, device=dev); bo = torch.empty(81920, 128, device=dev)
print("ops.gemm ws3 (async) ", round(t(lambda: ops.gemm([(big, None, w)], bo, bias=bias), 300), 2), "us (GPU-bound if > 26)")
