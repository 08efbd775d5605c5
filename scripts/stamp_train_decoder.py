"""Per-phase s_memtime shares of one step of the TRAINING-graph decoder kernel (diagnostic build: ASR_CHAIN_STAMP=1; wave 0 of
workgroup 0).  Phase = code up to each barrier of a step, in program order (same phase list as scripts/stamp_greedy.py)."""
import os, sys
os.environ["ASR_CHAIN_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from e2e_asr_amd import _lib
from e2e_asr_amd.weights import synthetic_batch
dev = torch.device("cuda:0")
dbg = torch.zeros(96, dtype=torch.int64, device=dev)
_lib.lib().asr_debug_set_buffer(dbg.data_ptr())
model = bench.build_model(dev, training=True)
batch = synthetic_batch(B=32, T=800, F=80, t_dec=121, vocab=1000, variable_len=False, seed=1)
for _ in range(2):
    model.forward(batch)
torch.cuda.synchronize()
d = dbg.cpu().numpy()[48:72]
names = ["(1) LM cell + publish; (2) gather lm_out", "outer cell matvec", "cell, publish (q,h); (3) gather (q,h)", "y matvec",
         "y publish; (4) gather y", "scores (tanh)", "e publish; (5) gather scores", "softmax", "context partials",
         "ctx publish; (6) gather ctx", "AttnProjection matvec", "p publish; (7) gather p", "logit slice + next LM h-part",
         "slice argmax, publish; (8) gather partial maxima", "token reduce"]
steps = 120
tot = float(d.sum())
print("training decoder: %.0f ticks per step" % (tot / steps))
for n, v in zip(names + ["?"] * 9, d):
    if v:
        print("  %-52s %6.1f %%  %6.0f ticks/step" % (n, 100.0 * v / tot, v / steps))
