"""BASELINE config 5: batch-1 beam search (width 16) with LM shallow fusion on one MI355X -- utterances/s and per-step
time of the device path.  Token-id parity with the float64 oracle is asserted in tests/test_gpu_beam.py, not here.
Diagnostic (the bench line is bench.py's config 2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from e2e_asr_amd.beam_search import BeamSearch
from e2e_asr_amd.weights import init_weights

rng = np.random.default_rng(0)
wd = {k: v for k, v in init_weights(seed=3).items() if "rnn_decoder_char" in k}
wl = {k: v for k, v in init_weights(seed=4).items() if "rnn_decoder_char" in k}
sp = BeamSearch.class_params()
sp.beam_size = 16; sp.lm_weight = 0.1; sp.lm_path = wl
bs = BeamSearch(wd, sp)
encs = [(rng.standard_normal((100, 512)) * 0.3).astype(np.float32) for _ in range(8)]
outs = [bs(e) for e in encs[:2]]                 # warm-up
torch.cuda.synchronize()
t0 = time.time()
outs = [bs(e) for e in encs]
torch.cuda.synchronize()
dt = time.time() - t0
steps = sum(len(o) for o in outs)
print("device beam search: %.2f utterances/s, %.1f ms/utterance, %.0f us per emitted token (beam 16, lm_weight 0.1, T_enc 100)" % (
    len(encs) / dt, dt / len(encs) * 1e3, dt / max(steps, 1) * 1e6))
