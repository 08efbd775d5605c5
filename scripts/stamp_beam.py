"""Per-phase s_memtime shares of the persistent beam-search kernel (diagnostic: ASR_BEAM_STAMP=1; workgroup 0, thread 0):
phase code up to its grid barrier, then the barrier itself, for the eight phases of a token (config-5 shapes)."""
import os, sys, time
os.environ["ASR_BEAM_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from e2e_asr_amd import _lib
from e2e_asr_amd.beam_search import BeamSearch
from e2e_asr_amd.weights import init_weights
dev = torch.device("cuda:0")
dbg = torch.zeros(64, dtype=torch.int64, device=dev)
_lib.lib().asr_debug_set_buffer(dbg.data_ptr())
rng = np.random.default_rng(0)
wd = {k: v for k, v in init_weights(seed=3).items() if "rnn_decoder_char" in k}
wl = {k: v for k, v in init_weights(seed=4).items() if "rnn_decoder_char" in k}
sp = BeamSearch.class_params()
sp.beam_size = 16; sp.lm_weight = 0.1; sp.lm_path = wl
bs = BeamSearch(wd, sp)
enc = torch.as_tensor((rng.standard_normal((100, 512)) * 0.3).astype(np.float32)).to(dev)
bs(enc); bs(enc)
torch.cuda.synchronize()
dbg.zero_()
t0 = time.perf_counter()
out = bs(enc)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
d = dbg.cpu().numpy()
tok = max(int(d[16]), 1)
names = ["1 LM cells", "2 InputProjection | LM logits", "3 outer cell", "4 attention", "5 AttnProjection", "6 OutputProjection",
         "7 scoring", "8 merge"]
print("%d tokens, %.1f us per token by the host clock; %.0f ticks per token in the kernel" % (tok, dt / tok * 1e6, d[:16].sum() / tok))
for i, n in enumerate(names):
    print("  %-32s work %7.0f   barrier %7.0f ticks per token" % (n, d[2 * i] / tok, d[2 * i + 1] / tok))
