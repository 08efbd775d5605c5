"""Char-LM train step (lm_model.py:39-115: the LM that shares the decoder's inner LSTM, embedding and OutputProjection;
batch 128, lm_model.py:31) on one MI355X.  Diagnostic."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from e2e_asr_amd import ops
from e2e_asr_amd.lm_encoder import LMEncoder
from e2e_asr_amd.lm_model import LMModel

dev = torch.device("cuda:0")
model = bench.build_model(dev, training=True)
ep = LMEncoder.class_params()
lm = LMModel(LMEncoder(isTraining=True, params=ep, variables=model.variables))
B, T = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(0)
lens = rng.integers(T // 2, T + 1, B); lens[0] = T
ids = np.zeros((B, T + 1), np.int64)
for b in range(B):
    ids[b, :lens[b] + 1] = rng.integers(3, 1000, lens[b] + 1)
batch = {"char": ids, "char_len": lens}
for _ in range(3):
    lm.step(batch)
torch.cuda.synchronize(); ops.check_device_flag(dev)
n = 10
t0 = time.perf_counter()
for _ in range(n):
    loss = lm.step(batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("LM train step B=%d T=%d: %.2f ms = %.2f M tokens/s (loss %.3f)" % (B, T, dt * 1e3, lens.sum() / dt / 1e6, float(loss)))
