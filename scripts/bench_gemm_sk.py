"""Stream-K against whole tiles on the encoder's products (config 2): forward projections NN, data gradients NT."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2e_asr_amd import ops
dev = torch.device("cuda:0")
shapes = [("NN", 12800, 2048, 1024), ("NN", 6400, 2048, 1024), ("NN", 3200, 2048, 1024), ("NN", 12800, 1024, 1024),
          ("NT", 12800, 1024, 2048), ("NT", 6400, 1024, 2048), ("NT", 3200, 1024, 2048), ("NN", 4096, 4096, 4096)]
for form, M, N, K in shapes:
    tb = form[1] == "T"
    a = torch.randn(M, K, device=dev); b = torch.randn((N, K) if tb else (K, N), device=dev); c = torch.empty(M, N, device=dev)
    res = {}
    for sk in ("0", "1"):
        os.environ["ASR_GEMM_SK"] = sk
        for _ in range(3):
            ops.gemm(a, b, None, False, tb, out=c)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            ops.gemm(a, b, None, False, tb, out=c)
        e1.record(); torch.cuda.synchronize()
        res[sk] = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * M * N * K
    print("%s %6d x %5d x %5d  tiles %5d   whole tiles %7.1f us (%5.1f TF/s)   stream-K %7.1f us (%5.1f TF/s)" % (
        form, M, N, K, (M // 128) * (N // 128), res["0"], fl / res["0"] / 1e6, res["1"], fl / res["1"] / 1e6))
