"""Microbenchmark of asr_gemm_f32 on the hot path's shapes (random data): the split3 path (fp32 on the bf16 matrix pipe by
exact 3-way operand splitting) next to the exact fp32 MFMA kernel, rocBLAS (torch.matmul) beside them for orientation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2e_asr_amd import ops
dev = torch.device("cuda:0")
shapes = [("4096^3 NN", 4096, 4096, 4096, 0, 0), ("L2 proj NN", 12800, 1024, 1024, 0, 0), ("L1 proj NN", 25600, 1024, 80, 0, 0),
          ("dX NT", 12800, 1024, 1024, 0, 1), ("dX L1->L2 NT", 25600, 1024, 2048, 0, 1), ("dKx TN", 1024, 1024, 12800, 1, 0),
          ("dKh TN", 256, 1024, 25600, 1, 0), ("dP NT", 3840, 256, 1000, 0, 1), ("wgrad out TN", 256, 1000, 3840, 1, 0)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, M, N, K, ta, tb in shapes:
    a = torch.randn((K, M) if ta else (M, K), device=dev)
    b = torch.randn((N, K) if tb else (K, N), device=dev)
    c = torch.zeros(M, N, device=dev)
    res = []
    for mname, split in (("split3", True), ("exact", False)):
        ops.set_gemm_split(split)
        ms = timed(lambda: ops.gemm(a, b, None, bool(ta), bool(tb), out=c, accumulate=bool(ta)))
        res.append("%s %7.1f us %6.1f TF/s" % (mname, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
    ops.set_gemm_split(True)
    at, bt = (a.t() if ta else a), (b.t() if tb else b)
    ms2 = timed(lambda: torch.matmul(at, bt, out=c))
    print("%-14s M=%5d N=%5d K=%5d | %s | %s | rocBLAS %7.1f us %6.1f TF/s" % (
        name, M, N, K, res[0], res[1], ms2 * 1e3, 2.0 * M * N * K / ms2 / 1e9))
