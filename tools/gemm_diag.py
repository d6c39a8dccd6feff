import os, sys, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gnnepcsaft_amd import ops
dev = torch.device("cuda:0")
def bench(M, N, K, b_trans=True, iters=50):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) if b_trans else torch.randn(K, N, device=dev)
    out = torch.empty(M, N, device=dev)
    for _ in range(5): ops.gemm([(a, None, w)], out, b_trans=b_trans)
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): ops.gemm([(a, None, w)], out, b_trans=b_trans)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for shape in [(81920,128,128), (163840,128,128), (81920,128,640), (81920,128,1664), (327680,128,128)]:
    us = bench(*shape)
    M,N,K = shape
    print(f"dbg={os.environ.get('GNX_GEMM_DBG','0')} M={M} N={N} K={K}: {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TF  ideal_mfma {2*M*N*K/157.3e6:6.1f} us  bytes {(M*K+M*N)*4/us/1e6:6.2f} TB/s", flush=True)
