"""Knock-out timings of the P3 KK kernel (ASR_P3_DBG: 0 real, 1 no LDS-DMA in the loop, 2 no MFMAs, 3 no fragment reads)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2e_asr_amd import ops
dev = torch.device("cuda:0")


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


for name, M, N, K in [("512 tiles K=4096", 8192, 2048, 4096), ("256 tiles K=4096", 4096, 2048, 4096), ("L2 proj fused", 12800, 2048, 1024)]:
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev)
    c = torch.empty(M, N, device=dev)
    ap, bp = ops.p3_split(a, 3), ops.p3_split(b, 3)
    t = timed(lambda: ops.gemm_p3_kk(ap, bp, None, out=c))
    print("dbg %s %-18s %8.1f us %6.1f TF/s" % (os.environ.get("ASR_P3_DBG", "0"), name, t * 1e3, 2.0 * M * N * K / t / 1e9), flush=True)
