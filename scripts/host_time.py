"""How long does the HOST take to enqueue one train step (no synchronisation inside)?  If this approaches the GPU's step time
the GPU starves.  Diagnostic."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from e2e_asr_amd.weights import synthetic_batch
dev = torch.device("cuda:0")
m = bench.build_model(dev)
b = synthetic_batch(B=32, T=800, F=80, t_dec=121, vocab=1000, seed=1234)
b = {k: (torch.as_tensor(v).to(dev) if k == "logmel" else v) for k, v in b.items()}
for _ in range(5):
    m.step(b)
torch.cuda.synchronize()
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.forward(b); t1 = time.perf_counter()
    m.backward(); t2 = time.perf_counter()
    m.apply_gradients(); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1, t3 - t2, t4 - t0))
import numpy as np
a = np.array(ts) * 1e3
print("host enqueue ms: forward %.2f  backward %.2f  optimizer %.2f | step incl. GPU %.2f" % tuple(np.median(a, 0)))
