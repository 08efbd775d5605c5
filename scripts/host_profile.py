"""cProfile of the host side of train steps (what the Python layer spends while it enqueues a step).  Diagnostic."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from e2e_asr_amd.weights import synthetic_batch
dev = torch.device("cuda:0")
m = bench.build_model(dev)
b = synthetic_batch(B=32, T=800, F=80, t_dec=121, vocab=1000, seed=1234, variable_len=True)
b = {k: (torch.as_tensor(v).to(dev) if k == "logmel" else v) for k, v in b.items()}
for _ in range(5):
    m.step(b)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    m.step(b)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
