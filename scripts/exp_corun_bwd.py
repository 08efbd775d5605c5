"""How much of a weight-gradient GEMM gets done NEXT TO the lowest layer's BPTT?  (the exposed tail of the train step is the
side stream's backlog after the last BPTT, profiles/r03_timeline_train_step.txt)
Main stream: the BPTT of the config-2 first layer (B = 32, T = 800, 80 inputs, H = 256; weight gradients of its own switched
to the caller's stream and excluded by timing the recurrent kernel alone with the library's event pairs).  Side stream: NG
weight-gradient products of the layer above (X^T.dG: 1024 x 2048 over 12 800 rows, split3), back to back.
Reported: BPTT alone, GEMMs alone, both together, and the share of the GEMM work that ran beside the recurrence."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2e_asr_amd import ops
dev = torch.device("cuda:0")
B, T, IN, H = int(os.environ.get("B", 32)), int(os.environ.get("T", 800)), 80, 256
NG = int(os.environ.get("NG", 2))
x = torch.randn(B, T, IN, device=dev) * 0.3
ln = torch.full((B,), T, dtype=torch.int32, device=dev)
k = [torch.randn(IN + H, 4 * H, device=dev) * 0.05 for _ in range(2)]
bz = [torch.zeros(4 * H, device=dev) for _ in range(2)]
dk = [torch.zeros_like(k[0]) for _ in range(2)]
db = [torch.zeros_like(bz[0]) for _ in range(2)]
wa, wb, wc = torch.randn(12800, 1024, device=dev), torch.randn(12800, 2048, device=dev), torch.empty(1024, 2048, device=dev)
side = torch.cuda.Stream()


def run(with_bptt, with_gemm):
    res = []
    for it in range(5):
        out, gates, act, hp = ops.lstm_layer_fwd(x, ln, k[0], bz[0], k[1], bz[1], save=True)
        dout = torch.ones_like(out)
        torch.cuda.synchronize()
        ops.prof_enable(False); ops.prof_enable(True)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        side.wait_stream(torch.cuda.current_stream())
        e[0].record()
        if with_gemm:
            with torch.cuda.stream(side):
                e[2].record()
                for _ in range(NG):
                    ops.gemm(wa, wb, None, True, False, out=wc)
                e[3].record()
        if with_bptt:
            ops.lstm_layer_bwd(x, ln, k[0], k[1], dout, gates, act, hp, dk[0], db[0], dk[1], db[1], need_dx=False, join=True)
        e[1].record()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        b_ms, b_n = ops.prof_read("lstm_rec_bwd")
        res.append((b_ms if with_bptt else 0.0, e[2].elapsed_time(e[3]) if with_gemm else 0.0))
    res = res[2:]
    return sum(r[0] for r in res) / len(res), sum(r[1] for r in res) / len(res)


b0, _ = run(True, False)
_, g0 = run(False, True)
b1, g1 = run(True, True)
print("T=%d: BPTT alone %.3f ms (%.2f us/step); %d GEMMs alone %.3f ms; together: BPTT %.3f ms (%.2f us/step), GEMMs done after %.3f ms" % (
    T, b0, b0 / T * 1e3, NG, g0, b1, b1 / T * 1e3, g1))
print("serial %.3f ms, together max %.3f ms: gained %.3f ms = %.0f %% of the GEMM time" % (
    b0 + g0, max(b1, g1), b0 + g0 - max(b1, g1), 100 * (b0 + g0 - max(b1, g1)) / g0))
