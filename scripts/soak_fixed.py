"""Repeat train steps on ONE batch shape (hunting an intermittent exchange time-out): soak_fixed.py B T t_dec steps"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from e2e_asr_amd import ops
from e2e_asr_amd.weights import synthetic_batch
dev = torch.device("cuda:0")
B, T, td, n = [int(x) for x in sys.argv[1:5]]
model = bench.build_model(dev, training=True)
t0 = time.time(); fails = 0
for it in range(n):
    b = synthetic_batch(B=B, T=T, F=80, t_dec=td, vocab=1000, variable_len=True, seed=it)
    model.step(b)
    try:
        ops.check_device_flag(dev)
    except RuntimeError as e:
        fails += 1
        print("step %d: %s" % (it, str(e)[-70:]), flush=True)
        if fails >= 5:
            break
print("B=%d T=%d t_dec=%d: %d failures in %d steps (%.1f s) env %s" % (B, T, td, fails, it + 1, time.time() - t0,
      {k: v for k, v in os.environ.items() if k.startswith("ASR_")}))
