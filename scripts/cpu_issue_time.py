"""How long does the host take to ISSUE one train step (no synchronisation), next to the GPU's time per step?
If the two are close the step is launch-bound somewhere.  Diagnostic."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from e2e_asr_amd.weights import synthetic_batch
dev = torch.device("cuda:0")
model = bench.build_model(dev, training=True)
batch = synthetic_batch(B=32, T=800, F=80, t_dec=121, vocab=1000, variable_len=False, seed=1234)
batch = {k: (torch.as_tensor(v).to(dev) if k == "logmel" else v) for k, v in batch.items()}
for _ in range(5):
    model.step(batch)
torch.cuda.synchronize()
n = 20
marks = {}
import e2e_asr_amd.seq2seq_model as M
t0 = time.perf_counter()
issue = 0.0
fw = bw = ap = 0.0
for _ in range(n):
    a = time.perf_counter(); model.forward(batch); b = time.perf_counter(); model.backward(); c = time.perf_counter(); model.apply_gradients(); d = time.perf_counter()
    fw += b - a; bw += c - b; ap += d - c
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("per step: wall %.2f ms; host issue: forward %.2f ms, backward %.2f ms, apply %.2f ms (sum %.2f ms)" % (
    tot / n * 1e3, fw / n * 1e3, bw / n * 1e3, ap / n * 1e3, (fw + bw + ap) / n * 1e3))
