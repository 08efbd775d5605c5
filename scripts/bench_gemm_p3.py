"""P3 GEMMs (operands pre-split into bf16 planes, csrc/gemm_p3.hip) against round 3's split3 kernel (fp32 operands split inside
the k-loop) on the hot path's shapes: accuracy of both against float64, time of each alone (HIP events, random data), and the
time of the split passes the P3 path needs when a producer does not emit planes itself."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2e_asr_amd import ops
dev = torch.device("cuda:0")
NP = int(os.environ.get("NP", "3"))


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


def err64(c, a, b):
    """max |c - a b^T| / (|a| |b|^T) over a sample of rows (float64 on the device)."""
    idx = torch.randperm(a.shape[0], device=dev)[:256]
    ref = a[idx].double() @ b.double().t()
    den = a[idx].double().abs() @ b.double().abs().t()
    return ((c[idx].double() - ref).abs() / den).max().item()


# KK form: C = A . B^T  (forward projection with the weights stored transposed; dX = dG . K_x^T with the weights as they are)
shapes = [("L2 proj", 12800, 1024, 1024), ("L2 proj fused dirs", 12800, 2048, 1024), ("L3 proj fused", 6400, 2048, 1024),
          ("L4 proj fused", 3200, 2048, 1024), ("dX L2 (K=8H)", 12800, 1024, 2048), ("4096^3", 4096, 4096, 4096)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev)
    b = torch.randn(N, K, device=dev)          # = B^T
    bias = torch.randn(N, device=dev)
    c = torch.empty(M, N, device=dev)
    ap, bp = ops.p3_split(a, NP), ops.p3_split(b, NP)
    ops.gemm_p3_kk(ap, bp, bias, out=c)
    e_p3 = err64(c - bias, a, b)
    t_p3 = timed(lambda: ops.gemm_p3_kk(ap, bp, bias, out=c))
    t_sa = timed(lambda: ops.p3_split(a, NP, out=ap))
    t_sb = timed(lambda: ops.p3_split(b, NP, out=bp))
    c2 = torch.empty(M, N, device=dev)
    ops.gemm(a, b, bias, trans_b=True, out=c2)
    e_s3 = err64(c2 - bias, a, b)
    t_s3 = timed(lambda: ops.gemm(a, b, bias, trans_b=True, out=c2))
    fl = 2.0 * M * N * K
    print("KK %-20s M=%5d N=%5d K=%5d | p3 %7.1f us %6.1f TF/s err %.2e | split3 %7.1f us %6.1f TF/s err %.2e | split A %5.1f us B %5.1f us | max|p3-split3| %.2e" % (
        name, M, N, K, t_p3 * 1e3, fl / t_p3 / 1e9, e_p3, t_s3 * 1e3, fl / t_s3 / 1e9, e_s3, t_sa * 1e3, t_sb * 1e3,
        (c - c2).abs().max().item()), flush=True)

# RR form: C = A^T . B (weight gradients: the contraction runs over the B*T rows of X and dG)
shapes = [("dKx L2 (both dirs)", 1024, 2048, 12800), ("dKx L2 one dir", 1024, 1024, 12800), ("dKx L3", 1024, 2048, 6400),
          ("dKx L4", 1024, 2048, 3200), ("dKh L1 one dir", 256, 1024, 25600), ("dKh L2 one dir", 256, 1024, 12800)]
for name, M, N, K in shapes:
    a = torch.randn(K, M, device=dev)
    b = torch.randn(K, N, device=dev)
    c = torch.zeros(M, N, device=dev)
    ap, bp = ops.p3_split(a, NP), ops.p3_split(b, NP)
    ops.gemm_p3_rr(ap, bp, out=c, accumulate=False)
    e_p3 = err64(c, a.t().contiguous(), b.t().contiguous())
    t_p3 = timed(lambda: ops.gemm_p3_rr(ap, bp, out=c, accumulate=True))
    t_sa = timed(lambda: ops.p3_split(a, NP, out=ap))
    t_sb = timed(lambda: ops.p3_split(b, NP, out=bp))
    c2 = torch.zeros(M, N, device=dev)
    ops.gemm(a, b, None, trans_a=True, out=c2, accumulate=True)
    e_s3 = err64(c2, a.t().contiguous(), b.t().contiguous())
    t_s3 = timed(lambda: ops.gemm(a, b, None, trans_a=True, out=c2, accumulate=True))
    fl = 2.0 * M * N * K
    print("RR %-20s M=%5d N=%5d K=%5d | p3 %7.1f us %6.1f TF/s err %.2e | split3 %7.1f us %6.1f TF/s err %.2e | split A %5.1f us B %5.1f us" % (
        name, M, N, K, t_p3 * 1e3, fl / t_p3 / 1e9, e_p3, t_s3 * 1e3, fl / t_s3 / 1e9, e_s3, t_sa * 1e3, t_sb * 1e3), flush=True)
