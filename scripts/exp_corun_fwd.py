"""What does a projection GEMM cost the forward recurrence when the two run at the same time?  (the measurement behind not
building the progress word for the forward pass)
A BiLSTM layer of the config-2 encoder (B = 32, H = 256, T = 400, in = 1024) runs on the main stream; a side stream runs the
NEXT layer's input projection (6 400 x 1024 x 2048, split3) back to back, as many launches as fit under the recurrence.
Reported: the layer alone, the GEMM alone, both together."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2e_asr_amd import ops
dev = torch.device("cuda:0")
B, T, IN, H = 32, int(os.environ.get("T", 400)), 1024, 256
x = torch.randn(B, T, IN, device=dev) * 0.3
ln = torch.full((B,), T, dtype=torch.int32, device=dev)
k = [torch.randn(IN + H, 4 * H, device=dev) * 0.05 for _ in range(2)]
bz = [torch.zeros(4 * H, device=dev) for _ in range(2)]
a = torch.randn(B * T // 2, 1024, device=dev) * 0.3
w = torch.randn(1024, 2048, device=dev) * 0.05
o = torch.empty(B * T // 2, 2048, device=dev)
side = torch.cuda.Stream()
NG = int(os.environ.get("NG", 3))


def timed(fn_main, fn_side, reps=6):
    tm, ts = [], []
    for it in range(reps + 2):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        torch.cuda.synchronize()
        if fn_side:
            side.wait_stream(torch.cuda.current_stream())
        e[0].record()
        if fn_main:
            fn_main()
        e[1].record()
        if fn_side:
            with torch.cuda.stream(side):
                e[2].record()
                fn_side()
                e[3].record()
        torch.cuda.synchronize()
        if it >= 2:
            tm.append(e[0].elapsed_time(e[1]) if fn_main else 0.0)
            ts.append(e[2].elapsed_time(e[3]) if fn_side else 0.0)
    return sum(tm) / len(tm), sum(ts) / len(ts)


layer = lambda: ops.lstm_layer_fwd(x, ln, k[0], bz[0], k[1], bz[1], save=True)
gemms = lambda: [ops.gemm(a, w, out=o) for _ in range(NG)]
m0, _ = timed(layer, None)
_, s0 = timed(None, gemms)
m1, s1 = timed(layer, gemms)
print("T=%d: layer alone %.3f ms; %d GEMMs alone %.3f ms (%.3f each); together: layer %.3f ms (+%.3f), GEMMs %.3f ms" % (
    T, m0, NG, s0, s0 / NG, m1, m1 - m0, s1))
print("hidden %.3f ms of GEMM for %.3f ms of recurrence: net %+.3f ms" % (s0, m1 - m0, s0 - (m1 - m0)))
