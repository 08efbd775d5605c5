"""Train-step time when every step sees a NEW batch (token ids / lengths differ, so the content-keyed device cache of the small
integer arrays misses), logmel already resident: the cost of the per-step small uploads that bench.py's repeated batch hides."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from e2e_asr_amd.weights import synthetic_batch
dev = torch.device("cuda:0")
model = bench.build_model(dev, training=True)
var = len(sys.argv) > 1 and sys.argv[1] == "var"
batches = []
for i in range(40):
    b = synthetic_batch(B=32, T=800, F=80, t_dec=121, vocab=1000, variable_len=var, seed=1000 + i)
    b["logmel"] = torch.as_tensor(b["logmel"]).to(dev)
    batches.append(b)
for i in range(5):
    model.step(batches[i])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(5, 35):
    model.step(batches[i])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 30
for i in range(3):
    model.step(batches[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(30):
    model.step(batches[0])
torch.cuda.synchronize()
dt0 = (time.perf_counter() - t0) / 30
print("fresh batch every step: %.3f ms/step; same batch: %.3f ms/step (variable_len=%s)" % (dt * 1e3, dt0 * 1e3, var))
