"""Per-phase s_memtime shares of one step of the persistent decoder backward chain (diagnostic build: ASR_CHAIN_STAMP=1;
wave 0 of workgroup 0).  Phases = code between consecutive barriers of a step, in program order."""
import os, sys
os.environ["ASR_CHAIN_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from e2e_asr_amd import _lib, ops
from e2e_asr_amd.weights import synthetic_batch
dev = torch.device("cuda:0")
dbg = torch.zeros(32, dtype=torch.int64, device=dev)
_lib.lib().asr_debug_set_buffer(dbg.data_ptr())
model = bench.build_model(dev, training=True)
batch = synthetic_batch(B=32, T=800, F=80, t_dec=121, vocab=1000, variable_len=False, seed=1)
for _ in range(2):
    model.step(batch)
torch.cuda.synchronize()
dbg.zero_()
model.step(batch)
torch.cuda.synchronize()
d = dbg.cpu().numpy()
names = ["gather dG (all-gather) + prefetch hand-over", "[dh|dctx] = dG.[K_h;WK_c]^T for my outputs", "dh, dctx_tot: publish + gather (all-gather)",
         "dalpha for my positions, S", "(d) tanh backward", "dy row reduce", "X2 publish + gather (reduce-scatter)",
         "dY save, dy publish + gather (all-gather)", "dq for my units", "cell pointwise + dG publish"]
tot = float(d[:10].sum())
steps = 120
print("cycles per step: %.0f  (%.2f us at 100 MHz s_memtime clock)" % (tot / steps, tot / steps / 100.0))
for n, v in zip(names, d[:10]):
    print("  %-50s %6.1f %%  %7.0f ticks/step" % (n, 100.0 * v / tot, v / steps))

f = d[16:]
fn = ["(1) gather state [h|ctx]", "(2) cell matvec + DPP", "cell, publish q and h; (3) gather q", "y matvec", "y publish; (4) gather y", "scores (tanh)",
      "e publish; (5) gather scores", "softmax (replicated)", "context partials", "context publish, alpha store"]
totf = float(f[:10].sum())
print("forward chain: %.0f cycles per step over 120 steps; prologues of the call's launches: %.0f cycles in total" % (totf / steps, float(f[15])))
for n, v in zip(fn, f[:10]):
    print("  %-50s %6.1f %%  %7.0f ticks/step" % (n, 100.0 * v / totf, v / steps))
