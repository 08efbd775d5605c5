"""Which products carry config 3's logit error (diagnostic): encoder / decoder GEMMs in different operand precisions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from e2e_asr_amd import ops
from e2e_asr_amd.encoder import Encoder
from e2e_asr_amd.weights import synthetic_batch
from oracle import asr_oracle as O
from tests.test_gpu_model import _model, _f64
b = synthetic_batch(B=32, T=800, F=80, t_dec=121, vocab=1000, variable_len=True, seed=4321)
m = _model(feat=80, vocab={"char": 1000}, params_update=dict(max_output={"char": 120}), seed=17)
w = _f64(m.variables.to_arrays())
b64 = dict(b); b64["logmel"] = b["logmel"].astype(np.float64)
r = O.seq2seq_forward(b64, w, is_training=True)
orig = Encoder.__call__
for enc_p, dec_p in (("bf16", "bf16"), ("bf16", "f32"), ("f32", "bf16"), ("bf16", "bf16x2"), ("bf16x2", "bf16")):
    def call(self, *a, **k):
        ops.set_gemm_precision(enc_p)
        try:
            return orig(self, *a, **k)
        finally:
            ops.set_gemm_precision(dec_p)
    Encoder.__call__ = call
    ops.set_gemm_precision(dec_p)
    m.forward(b)
    err = np.abs(m.outputs["char"].cpu().numpy() - r["outputs"]["char"]).max()
    enc_err = np.abs(m.encoder_hidden_states[4].cpu().numpy() - r["enc"][4]).max()
    print("encoder %-6s decoder %-6s: max|logit diff| %.3g   max|encoder state diff| %.3g" % (enc_p, dec_p, err, enc_err))
