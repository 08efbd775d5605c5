"""What slows the persistent BPTT when the weight-gradient GEMMs run next to it?  The recurrent kernels of one BiLSTM layer
(B=32, H=256, T=400) alone and with a second stream kept busy by (a) a device-to-device copy (HBM/L2 traffic, no MFMA),
(b) the library's fp32 GEMM on an L2-resident problem (MFMA + LDS + L2, little HBM), (c) the same GEMM on a streaming
problem (the weight-gradient shape).  Diagnostic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2e_asr_amd import ops
dev = torch.device("cuda:0")
B, H, T, IN = 32, 256, 400, 1024
x = torch.randn(B, T, IN, device=dev) * 0.3
ln = torch.full((B,), T, dtype=torch.int32, device=dev)
k = [torch.randn(IN + H, 4 * H, device=dev) * 0.05 for _ in range(2)]
bz = [torch.zeros(4 * H, device=dev) for _ in range(2)]
dk = [torch.zeros_like(k[0]) for _ in range(2)]
db = [torch.zeros_like(bz[0]) for _ in range(2)]
side = torch.cuda.Stream(priority=0)
big_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev); big_b = torch.empty_like(big_a)
ga, gb, gc = torch.randn(1024, 1024, device=dev), torch.randn(1024, 1024, device=dev), torch.empty(1024, 1024, device=dev)
wa, wb, wc = torch.randn(12800, 1280, device=dev), torch.randn(12800, 1024, device=dev), torch.empty(1280, 1024, device=dev)


def load(kind, n):
    with torch.cuda.stream(side):
        for _ in range(n):
            if kind == "copy":
                big_b.copy_(big_a, non_blocking=True)
            elif kind == "gemm_l2":
                for _ in range(20):
                    ops.gemm(ga, gb, None, False, False, out=gc)
            elif kind == "gemm_wgrad":
                ops.gemm(wa, wb, None, True, False, out=wc)


for kind, n in (("none", 0), ("copy", 40), ("gemm_l2", 40), ("gemm_wgrad", 40)):
    res = []
    for it in range(4):
        out, gates, act, hp = ops.lstm_layer_fwd(x, ln, k[0], bz[0], k[1], bz[1], save=True)
        dout = torch.ones_like(out)
        torch.cuda.synchronize()
        ops.prof_enable(False); ops.prof_enable(True)
        load(kind, n)
        ops.lstm_layer_bwd(x, ln, k[0], k[1], dout, gates, act, hp, dk[0], db[0], dk[1], db[1], need_dx=False, join=True)
        torch.cuda.synchronize()
        b_ms, b_n = ops.prof_read("lstm_rec_bwd")
        res.append(b_ms / max(b_n, 1) / T * 1e3)
        side.synchronize()
    print("%-11s BPTT %.3f us/step  (runs: %s)" % (kind, min(res[1:]), " ".join("%.3f" % r for r in res)))
