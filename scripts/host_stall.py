"""Where does a one-off host stall inside an un-synchronised run of train steps come from?  (BENCH_r04: 10.34 ms per step on the
driver's clock over 20 steps = 51 ms more than 20 x 7.78 ms; bench.py's per-step events show ONE step of ~42 ms whose host
enqueue took ~40 ms.)  Per step: host time of forward / backward / optimizer, every garbage-collector pass (generation,
duration) through gc.callbacks, the caching allocator's counters (a cudaMalloc inside the loop would show), and the
GPU's per-step event times.  Diagnostic."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from e2e_asr_amd.weights import synthetic_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
m = bench.build_model(dev)
b = synthetic_batch(B=32, T=800, F=80, t_dec=121, vocab=1000, seed=1234)
b = {k: (torch.as_tensor(v).to(dev) if k == "logmel" else v) for k, v in b.items()}
for _ in range(5):
    m.step(b)
torch.cuda.synchronize()
gcs = []
_t = [0.0]
def cb(phase, info):
    if phase == "start":
        _t[0] = time.perf_counter()
    else:
        gcs.append((info["generation"], (time.perf_counter() - _t[0]) * 1e3, info.get("collected", 0), time.perf_counter()))
gc.callbacks.append(cb)
print("gc counts before the loop", gc.get_count(), "thresholds", gc.get_threshold(), "objects", len(gc.get_objects()))
st0 = torch.cuda.memory_stats()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
rows = []
t00 = time.perf_counter()
ev[0].record()
for i in range(n):
    t0 = time.perf_counter(); m.forward(b)
    t1 = time.perf_counter(); m.backward()
    t2 = time.perf_counter(); m.apply_gradients()
    t3 = time.perf_counter(); ev[i + 1].record()
    rows.append((t1 - t0, t2 - t1, t3 - t2, t3))
torch.cuda.synchronize()
wall = time.perf_counter() - t00
st1 = torch.cuda.memory_stats()
a = np.array([r[:3] for r in rows]) * 1e3
g = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print("wall %.2f ms per step; GPU step median %.3f max %.3f; host median fwd %.3f bwd %.3f opt %.3f" % (
    wall / n * 1e3, np.median(g), max(g), *np.median(a, 0)))
for i in range(n):
    if a[i].sum() > 3.0 or g[i] > 9.0:
        print("  step %d: host fwd %.2f bwd %.2f opt %.2f ms | GPU %.2f ms" % (i, a[i, 0], a[i, 1], a[i, 2], g[i]))
for gen, ms, coll, when in gcs:
    step = next((i for i, r in enumerate(rows) if r[3] >= when), n)
    print("  gc generation %d: %.2f ms, collected %d, during step %d" % (gen, ms, coll, step))
for k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "segment.all.allocated", "reserved_bytes.all.current"):
    print("  allocator %s: %s -> %s" % (k, st0.get(k), st1.get(k)))
