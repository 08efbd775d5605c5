"""Diagnostic: per-phase cycle shares of one recurrent LSTM step (stamped build; shares only)."""
import os, sys
os.environ["ASR_LSTM_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from e2e_asr_amd import ops, _lib
dev = torch.device("cuda:0")
dbg = torch.zeros(64, dtype=torch.int64, device=dev)
_lib.lib().asr_debug_set_buffer(dbg.data_ptr())
rng = np.random.default_rng(0)
B, T, IN, H = 32, 800, 80, 256
x = torch.from_numpy(rng.standard_normal((B, T, IN)).astype(np.float32)).to(dev)
k = torch.from_numpy(rng.uniform(-0.075, 0.075, (IN + H, 4 * H)).astype(np.float32)).to(dev)
bz = torch.zeros(4 * H, device=dev)
ln = torch.full((B,), T, dtype=torch.int32, device=dev)
for _ in range(3):
    ops.lstm_layer_fwd(x, ln, k, bz, k, bz)
torch.cuda.synchronize()
v1 = os.environ.get("ASR_LSTM_V2") == "0"
names = ["prefetch", "poll+lds-write", "barrier1", "matvec+dpp+sums", "barrier2", "cell+store+publish"]
roles = (("cell wave (tid 0)", 0, names), ("polling wave (last tid)", 8, names))
if not v1:       # version 2 (slice per wave): stamps of tid 0 (cell wave) and tid 64 (first polling wave)
    roles = (("cell wave (tid 0)", 0, ["own slice -> barrier arrive", "barrier wait", "sum partials + cell + publish", "bookkeeping + prefetch",
                                       "own slice matvec", "(one stamp, back to back)"]),
             ("polling wave 1 (tid 64)", 8, ["barrier exit -> poll start", "poll until hit + lds write", "lds hop + matvec + swap + write",
                                             "barrier wait", "-", "-"]))
for who, off, nm in roles:
    d = dbg.cpu().numpy()[off:off + 8]
    S = int(d[6]); tot = d[:6].sum()
    print(who, "steps", S, "cycles/step", tot / S)
    for n, v in zip(nm, d[:6]):
        print("   %-34s %8.1f cyc/step  %5.1f%%" % (n, v / S, 100.0 * v / tot))

if not v1:
    d = dbg.cpu().numpy()
    S = int(d[6])
    print("per polling wave (cycles/step): barrier exit -> hit | product | barrier wait")
    for w in range(1, 8):
        print("   wave %d: %7.1f | %7.1f | %7.1f" % (w, d[16 + w] / S, d[32 + w] / S, d[48 + w] / S))
