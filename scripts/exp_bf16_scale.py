"""Config-3 (bf16 operands) logit error against the float64 oracle as the LSTM kernels are scaled up ("trained-like" weights),
next to the fp32 path's own error on the same weights: how much of the growth is the bf16 rounding and how much the network's
sensitivity (a chaotic recurrence amplifies ANY perturbation, fp32 rounding included)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_model import _model, _f64
from e2e_asr_amd import ops
from e2e_asr_amd.weights import synthetic_batch
from oracle import asr_oracle as O
T = int(os.environ.get("T", "800"))
for wscale in (1.0, 1.5, 2.0, 3.0):
    res = {}
    for prec in ("f32", "bf16"):
        ops.set_gemm_precision(prec)
        m = _model(feat=80, vocab={"char": 1000}, params_update=dict(max_output={"char": 120}), seed=17)
        with torch.no_grad():
            for n in m.variables.names():
                if n.endswith("/kernel") and "basic_lstm_cell" in n:
                    m.variables[n].mul_(wscale)
        b = synthetic_batch(B=32, T=T, F=80, t_dec=121, vocab=1000, variable_len=True, seed=4321)
        m.forward(b)
        out = m.outputs["char"].cpu().numpy()
        if prec == "f32":
            w = _f64(m.variables.to_arrays())
            b64 = dict(b); b64["logmel"] = b["logmel"].astype(np.float64)
            ref = O.seq2seq_forward(b64, w, is_training=True)["outputs"]["char"]
        res[prec] = np.abs(out - ref).max()
    ops.set_gemm_precision("f32")
    print("LSTM kernels x %.1f: max |logit - float64 oracle|  fp32 path %.3g   bf16 mode %.3g   ratio %.0f   (max |logit| %.2f)" % (
        wscale, res["f32"], res["bf16"], res["bf16"] / max(res["f32"], 1e-30), np.abs(ref).max()), flush=True)
