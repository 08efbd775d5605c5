"""Per-phase s_memtime shares of the all-gather BPTT kernel (diagnostic build, ASR_LSTM_STAMP=1): thread 0 (cell wave) and
thread 511 (polling wave) of workgroup 0; (a) one layer alone, (b) inside the full train step, where the side-stream
weight-gradient GEMMs co-run."""
import os, sys
os.environ["ASR_LSTM_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from e2e_asr_amd import _lib, ops
from e2e_asr_amd.weights import synthetic_batch
dev = torch.device("cuda:0")
dbg = torch.zeros(64, dtype=torch.int64, device=dev)
_lib.lib().asr_debug_set_buffer(dbg.data_ptr())
names = ["poll (gather dG)", "barrier 1", "matvec + DPP", "barrier 2", "cell + publish"]
V1 = os.environ.get("ASR_LSTM_V2") == "0"
names_cell = names if V1 else ["own slice -> barrier arrive", "barrier wait", "sum partials + cell + publish", "bookkeeping + own staging", "own slice contraction"]
names_poll = names if V1 else ["barrier exit -> poll start", "poll until hit + staging", "contraction + reduce-scatter", "barrier wait", "-"]


def report(tag):
    d = dbg.cpu().numpy()
    steps = float(d[32 + 7])
    for who, off, nm in (("cell wave (thread 0)", 32, names_cell), ("polling wave (thread 511; version 2: thread 64)", 40, names_poll)):
        v = d[off:off + 5].astype(float)
        print("%s, %s: %.0f cycles per step" % (tag, who, v.sum() / steps))
        print("   " + "  ".join("%s %.0f" % (n, x / steps) for n, x in zip(nm, v)))
    if not V1:
        print("   busy cycles per step between barriers, waves 0..8 (0 cell, 1-7 contraction, 8 loader): " +
              " ".join("%.0f" % (x / steps) for x in d[48:57].astype(float)))


# (a) alone: one BiLSTM layer, B=32, T=400
B, T, IN, H = 32, 400, 1024, 256
x = torch.randn(B, T, IN, device=dev) * 0.3
ln = torch.full((B,), T, dtype=torch.int32, device=dev)
k = [torch.randn(IN + H, 4 * H, device=dev) * 0.05 for _ in range(2)]
bz = [torch.zeros(4 * H, device=dev) for _ in range(2)]
dk = [torch.zeros_like(k[0]) for _ in range(2)]; db = [torch.zeros_like(bz[0]) for _ in range(2)]
for it in range(3):
    out, gates, act, hp = ops.lstm_layer_fwd(x, ln, k[0], bz[0], k[1], bz[1], save=True)
    torch.cuda.synchronize()
    if it == 2:
        dbg.zero_()
    ops.lstm_layer_bwd(x, ln, k[0], k[1], torch.ones_like(out), gates, act, hp, dk[0], db[0], dk[1], db[1], need_dx=True, join=True)
    torch.cuda.synchronize()
report("alone")
# (b) in the train step
model = bench.build_model(dev, training=True)
batch = synthetic_batch(B=32, T=800, F=80, t_dec=121, vocab=1000, variable_len=False, seed=1)
for _ in range(2):
    model.step(batch)
torch.cuda.synchronize()
dbg.zero_()
model.step(batch)
torch.cuda.synchronize()
report("in the train step")
