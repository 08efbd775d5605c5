"""Target of the SQ / TCC PMC passes for the persistent recurrent pair (diagnostic, not a timing claim): one encoder layer of
config 2 (layer 2: B=32, T=400, in=1024, H=256, both directions), forward with saved activations, then backward, a few
repeats.  usage: pmc_lstm.py [T] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from e2e_asr_amd import ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 400
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
B, IN, H = 32, 1024, 256
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, T, IN)).astype(np.float32)).to(dev)
kf = torch.from_numpy(rng.uniform(-0.075, 0.075, (IN + H, 4 * H)).astype(np.float32)).to(dev)
kb = torch.from_numpy(rng.uniform(-0.075, 0.075, (IN + H, 4 * H)).astype(np.float32)).to(dev)
bz = torch.zeros(4 * H, device=dev)
ln = torch.full((B,), T, dtype=torch.int32, device=dev)
dk = [torch.zeros_like(kf), torch.zeros_like(kb)]
db = [torch.zeros_like(bz), torch.zeros_like(bz)]
for _ in range(reps):
    out, gates, act, hprev = ops.lstm_layer_fwd(x, ln, kf, bz, kb, bz, save=True)
    dout = torch.ones_like(out) * 1e-3
    ops.lstm_layer_bwd(x, ln, kf, kb, dout, gates, act, hprev, dk[0], db[0], dk[1], db[1], need_dx=True)
torch.cuda.synchronize()
ops.check_device_flag(dev)
