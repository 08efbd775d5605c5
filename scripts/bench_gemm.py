"""Microbenchmark of asr_gemm_f32 on the hot path's shapes (random data)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2e_asr_amd import ops
dev = torch.device("cuda:0")
shapes = [("4096^3 NN", 4096, 4096, 4096, 0, 0), ("L2 proj NN", 12800, 1024, 1024, 0, 0), ("L1 proj NN", 25600, 1024, 80, 0, 0),
          ("dX NT", 12800, 1024, 1024, 0, 1), ("dKx TN", 1024, 1024, 12800, 1, 0), ("dKh TN", 256, 1024, 25600, 1, 0),
          ("dP NT", 3840, 256, 1000, 0, 1), ("wgrad out TN", 256, 1000, 3840, 1, 0)]
for name, M, N, K, ta, tb in shapes:
    a = torch.randn((K, M) if ta else (M, K), device=dev)
    b = torch.randn((N, K) if tb else (K, N), device=dev)
    c = torch.zeros(M, N, device=dev)
    for _ in range(3):
        ops.gemm(a, b, None, bool(ta), bool(tb), out=c, accumulate=bool(ta))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.gemm(a, b, None, bool(ta), bool(tb), out=c, accumulate=bool(ta))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    # rocBLAS (torch.matmul) on the same operands, for orientation only
    at, bt = (a.t() if ta else a), (b.t() if tb else b)
    for _ in range(3):
        torch.matmul(at, bt, out=c)
    e0.record()
    for _ in range(n):
        torch.matmul(at, bt, out=c)
    e1.record(); torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1) / n
    print("%-14s M=%5d N=%5d K=%5d  %8.1f us  %6.1f TF/s   | rocBLAS %8.1f us %6.1f TF/s" % (
        name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9, ms2 * 1e3, 2.0 * M * N * K / ms2 / 1e9))
