"""Run-to-run determinism of one config-2-width train step (diagnostic): logits, loss and every gradient of two runs from
identical weights.  Forward values must be bit-identical; gradients may differ by the float atomics of the split-K GEMMs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from e2e_asr_amd import ops
from e2e_asr_amd.attn_decoder import AttnDecoder
from e2e_asr_amd.seq2seq_model import Seq2SeqModel
from e2e_asr_amd.weights import synthetic_batch
def params():
    p = Seq2SeqModel.class_params()
    p.num_layers = {"char": 4}; p.max_output = {"char": 30}
    p.encoder_params.use_lstm = True; p.encoder_params.out_prob = 1.0
    dp = AttnDecoder.class_params(); dp.out_prob_dec = 1.0; dp.samp_prob = 0.0; dp.vocab_size = 1000
    p.decoder_params = {"char": dp}
    return p
T = int(os.environ.get("T", 160))
b = synthetic_batch(B=32, T=T, F=80, t_dec=21, vocab=1000, variable_len=bool(int(os.environ.get("VARLEN", "0"))), seed=100)
res = []
from e2e_asr_amd.encoder import Encoder
_orig = Encoder.backward
cap = {}
def _bw(self, d_states, **kw):
    cap["denc"] = {d: v.clone() for d, v in d_states.items()}
    return _orig(self, d_states, **kw)
Encoder.backward = _bw
_olb = ops.lstm_layer_bwd
def _lb(x, seq_len, kf, kb, dout, gates, act, hprev, *a, **kw):
    cap.setdefault("layers", []).append(dict(dout=dout.clone(), act=act.clone(), hprev=hprev.clone(), x=x.clone()))
    dx = _olb(x, seq_len, kf, kb, dout, gates, act, hprev, *a, **kw)
    cap["layers"][-1].update(dG=gates.clone(), dx=None if dx is None else dx.clone())
    return dx
ops.lstm_layer_bwd = _lb
layers = []
dencs = []
for run in range(3):
    m = Seq2SeqModel(None, True, params(), device="cuda:0", feat_length=80, seed=6)
    m.forward(b); m.backward(); torch.cuda.synchronize()
    ops.check_device_flag(torch.device("cuda:0"))
    enc = {d: v.cpu().numpy().copy() for d, v in m.encoder_hidden_states.items()}
    dencs.append({d: v.cpu().numpy() for d, v in cap["denc"].items()})
    layers.append(cap.pop("layers"))
    res.append((m.outputs["char"].cpu().numpy().copy(), {n: m.variables.grad_of(n).cpu().numpy().copy() for n in m.variables.names()}, enc))
for run in (1, 2):
    for li, (l0, l1) in enumerate(zip(layers[0], layers[run])):
        print("  bwd call %d:" % li, {k: ("%.3g" % float((l1[k] - l0[k]).abs().max() / max(1e-30, float(l0[k].abs().max())))) for k in l0 if l0[k] is not None})
    print("denc run %d vs 0:" % run, {d: float(np.abs(dencs[run][d] - dencs[0][d]).max() / np.abs(dencs[0][d]).max()) for d in dencs[0]})
    print("run %d vs 0: logits max|diff| %.3g" % (run, np.abs(res[run][0] - res[0][0]).max()),
          " encoder states:", {d: float(np.abs(res[run][2][d] - res[0][2][d]).max()) for d in res[0][2]})
    worst = sorted(((float(np.abs(res[run][1][n] - res[0][1][n]).max() / max(1e-30, np.abs(res[0][1][n]).max())), n) for n in res[0][1]), reverse=True)
    for w, n in worst[:int(os.environ.get("TOP", 6))]:
        print("   %.3g  %s" % (w, n))
