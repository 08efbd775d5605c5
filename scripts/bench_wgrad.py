"""Weight-gradient GEMM shapes of the encoder (X^T . dG, K = B*T) alone: TFLOP/s with the K slices pinned to XCDs
(ASR_GEMM_XCD_SPLIT=1, default) or spread over them (=0).  Diagnostic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2e_asr_amd import ops
dev = torch.device("cuda:0")
for K, M, N in ((25600, 336, 1024), (12800, 1280, 1024), (6400, 1280, 1024), (3200, 1280, 1024)):
    a = torch.randn(K, M, device=dev); b = torch.randn(K, N, device=dev); c = torch.empty(M, N, device=dev)
    for _ in range(3):
        ops.gemm(a, b, None, True, False, out=c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm(a, b, None, True, False, out=c)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    ref = a.t() @ b
    err = (c - ref).abs().max().item() / ref.abs().max().item()
    print("K=%5d M=%4d N=%4d: %.3f ms = %.1f TF/s  (rel err vs torch %.1e)" % (K, M, N, ms, 2.0 * M * N * K / ms / 1e9, err))
