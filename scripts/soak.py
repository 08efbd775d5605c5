"""Soak: many train steps over randomly shaped ragged batches (batch 1..40, frames 40..300, targets 3..40, scheduled
sampling + dropout on) through the config-2-width model -- every step must finish (no exchange time-out), with a finite
loss and finite parameters.  Exercises the persistent kernels across group counts, segment patterns and lengths.
soak.py seed steps [f32|bf16]   (bf16: BASELINE config 3's precision mode, the plane kernels in the encoder)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from e2e_asr_amd import ops
from e2e_asr_amd.weights import synthetic_batch

dev = torch.device("cuda:0")
if len(sys.argv) > 3:
    ops.set_gemm_precision(sys.argv[3])
model = bench.build_model(dev, training=True)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
t0 = time.time()
for it in range(n):
    B = int(rng.integers(1, 41)); T = int(rng.integers(40, 301)); td = int(rng.integers(4, 41))
    b = synthetic_batch(B=B, T=T, F=80, t_dec=td, vocab=1000, variable_len=True, seed=int(rng.integers(1 << 30)))
    if it % 7 == 0:
        b["logmel_len"] = np.maximum(1, (np.asarray(b["logmel_len"]) // int(rng.integers(2, 9)))); b["logmel_len"][0] = T
    losses = model.step(b)
    loss = float(losses["char"].item())
    try:
        ops.check_device_flag(dev)
    except RuntimeError:
        print("FAILED at step %d: B=%d T=%d t_dec=%d lens=%s tlens=%s" % (it, B, T, td, list(b["logmel_len"]), list(b["char_len"])), flush=True)
        raise
    assert np.isfinite(loss), (it, B, T, td, loss)
    if it % 10 == 0:
        assert torch.isfinite(model.variables.flat).all(), it
        print("step %3d  B=%2d T=%3d t_dec=%2d  loss %.4f  (%.1f s)" % (it, B, T, td, loss, time.time() - t0), flush=True)
print("soak ok: %d steps in %.1f s" % (n, time.time() - t0))
