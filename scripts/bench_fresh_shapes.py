import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "scripts")
import numpy as np, torch, bench
from e2e_asr_amd.weights import synthetic_batch
dev = torch.device("cuda:0")
model = bench.build_model(dev, training=True)
rng = np.random.default_rng(0)
batches = []
for i in range(45):
    T = int(rng.integers(700, 801)); td = int(rng.integers(60, 121))
    b = synthetic_batch(B=32, T=T, F=80, t_dec=td, vocab=1000, variable_len=True, seed=1000 + i)
    b["logmel"] = torch.as_tensor(b["logmel"]).to(dev)
    batches.append(b)
for i in range(8): model.step(batches[i])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(8, 43): model.step(batches[i])
torch.cuda.synchronize()
print("varying T per batch (700-800), resident logmel: %.3f ms/step" % ((time.perf_counter() - t0) / 35 * 1e3))
print(torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_stats()["num_device_alloc"], "device allocs")
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(8, 43): model.step(batches[i])
torch.cuda.synchronize()
print("second pass over the same shapes: %.3f ms/step" % ((time.perf_counter() - t0) / 35 * 1e3), torch.cuda.memory_stats()["num_device_alloc"], "device allocs")
