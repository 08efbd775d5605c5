"""Fixed cost per launch of the recurrent kernels: one BiLSTM layer (B=32, H=256, 1024 inputs) at several T, least-squares line
time = a + b T for forward and BPTT (diagnostic; kernel times from the library's profiling events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from e2e_asr_amd import ops
dev = torch.device("cuda:0")
B, H, IN = 32, 256, 1024
res = []
for T in [int(t) for t in os.environ.get("TS", "25,50,100,200,400").split(",")]:
    x = torch.randn(B, T, IN, device=dev) * 0.3
    ln = torch.full((B,), T, dtype=torch.int32, device=dev)
    k = [torch.randn(IN + H, 4 * H, device=dev) * 0.05 for _ in range(2)]
    bz = [torch.zeros(4 * H, device=dev) for _ in range(2)]
    dk = [torch.zeros_like(k[0]) for _ in range(2)]
    db = [torch.zeros_like(bz[0]) for _ in range(2)]
    n = 8
    for it in range(n + 2):
        if it == 2:
            torch.cuda.synchronize(); ops.prof_enable(False); ops.prof_enable(True)
        out, gates, act, hp = ops.lstm_layer_fwd(x, ln, k[0], bz[0], k[1], bz[1], save=True)
        dout = torch.ones_like(out)
        torch.cuda.synchronize()
        ops.lstm_layer_bwd(x, ln, k[0], k[1], dout, gates, act, hp, dk[0], db[0], dk[1], db[1], need_dx=True, join=True)
        torch.cuda.synchronize()
    f_ms, f_n = ops.prof_read("lstm_rec_fwd")
    b_ms, b_n = ops.prof_read("lstm_rec_bwd")
    res.append((T, f_ms / f_n * 1e3, b_ms / b_n * 1e3))
    print("T=%4d  fwd %.1f us  bwd %.1f us" % res[-1])
Ts = np.array([r[0] for r in res], float)
for name, col in (("fwd", 1), ("bwd", 2)):
    y = np.array([r[col] for r in res])
    b, a = np.polyfit(Ts, y, 1)
    print("%s: %.2f us per launch + %.4f us per step" % (name, a, b))
