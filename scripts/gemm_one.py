"""One GEMM shape, a few launches: target for rocprofv3 --pmc runs (scripts/ are diagnostics, not timing claims)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2e_asr_amd import ops
M, N, K, ta, tb = [int(x) for x in sys.argv[1:6]] if len(sys.argv) > 5 else (4096, 4096, 4096, 0, 0)
dev = torch.device("cuda:0")
a = torch.randn((K, M) if ta else (M, K), device=dev)
b = torch.randn((N, K) if tb else (K, N), device=dev)
c = torch.zeros(M, N, device=dev)
for _ in range(5):
    ops.gemm(a, b, None, bool(ta), bool(tb), out=c, accumulate=False)
torch.cuda.synchronize()
