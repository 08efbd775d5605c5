"""The decoder's small products (M = T*B = 3840): split3 on 64x64 tiles (gemm_split3s_kernel, default) against the 64x64
instantiation of the fp32-input MFMA kernel (ASR_GEMM_S3S=0); rocBLAS beside them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2e_asr_amd import ops
dev = torch.device("cuda:0")
shapes = [("hf NN", 3200, 128, 512, 0, 0), ("wk NN", 768, 1024, 256, 0, 0), ("x NN", 3840, 256, 256, 0, 0), ("x ctx NN", 3808, 256, 512, 0, 0),
          ("dP NT", 3840, 256, 1000, 0, 1), ("dQC NT", 3840, 768, 256, 0, 1), ("dXH NT", 3840, 256, 1024, 0, 1), ("logits NN", 3840, 1000, 256, 0, 0)]
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, M, N, K, ta, tb in shapes:
    a = torch.randn((K, M) if ta else (M, K), device=dev); b = torch.randn((N, K) if tb else (K, N), device=dev)
    c = torch.zeros(M, N, device=dev)
    ms = timed(lambda: ops.gemm(a, b, None, bool(ta), bool(tb), out=c))
    at, bt = (a.t() if ta else a), (b.t() if tb else b)
    ms2 = timed(lambda: torch.matmul(at, bt, out=c))
    print("S3S=%s %-10s M=%5d N=%5d K=%5d  %6.1f us %6.1f TF/s | rocBLAS %6.1f us" % (os.environ.get("ASR_GEMM_S3S", "1"), name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9, ms2 * 1e3))
