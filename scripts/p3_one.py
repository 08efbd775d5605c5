"""One P3 GEMM shape, a few launches: target for rocprofv3 --pmc runs.  usage: p3_one.py kk|rr M N K [np]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2e_asr_amd import ops
form = sys.argv[1]
M, N, K = [int(x) for x in sys.argv[2:5]]
NP = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = torch.device("cuda:0")
if form == "kk":
    ap, bp = ops.p3_split(torch.randn(M, K, device=dev), NP), ops.p3_split(torch.randn(N, K, device=dev), NP)
    c = torch.empty(M, N, device=dev)
    for _ in range(5):
        ops.gemm_p3_kk(ap, bp, None, out=c)
else:
    ap, bp = ops.p3_split(torch.randn(K, M, device=dev), NP), ops.p3_split(torch.randn(K, N, device=dev), NP)
    c = torch.zeros(M, N, device=dev)
    for _ in range(5):
        ops.gemm_p3_rr(ap, bp, out=c, accumulate=True)
torch.cuda.synchronize()
