"""Soak of the multitask model (char depth 4 + phone decoder on depth `nlp`) over random ragged batch shapes: every step
must finish without an exchange time-out.  soak_multi.py seed steps nlp [Tmax]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from e2e_asr_amd import ops
from e2e_asr_amd.seq2seq_model import Seq2SeqModel
from e2e_asr_amd.attn_decoder import AttnDecoder
from e2e_asr_amd.weights import synthetic_batch

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
nlp = int(sys.argv[3]) if len(sys.argv) > 3 else 3
tmax = int(sys.argv[4]) if len(sys.argv) > 4 else 600      # depth 2 and Tmax = 900: the forward chain's two-pass groups (Te 257..432) and beyond
dev = torch.device("cuda:0")
p = Seq2SeqModel.class_params()
p.encoder_params.use_lstm = True
p.tasks = ["char", "phone"]
p.num_layers = {"char": 4, "phone": nlp}
dp = AttnDecoder.class_params(); dp.vocab_size = 50
p.decoder_params = {"char": AttnDecoder.class_params(), "phone": dp}
model = Seq2SeqModel(None, isTraining=True, params=p, device=dev, feat_length=80, seed=10)
rng = np.random.default_rng(seed)
t0 = time.time()
for it in range(n):
    B = int(rng.integers(1, 41)); T = int(rng.integers(40, tmax + 1)); td = int(rng.integers(4, 41)); tp = int(rng.integers(4, 90))
    b = synthetic_batch(B=B, T=T, F=80, t_dec=td, vocab=1000, variable_len=True, seed=int(rng.integers(1 << 30)), tasks=("char",))
    bp = synthetic_batch(B=B, T=T, F=80, t_dec=tp, vocab=50, variable_len=True, seed=int(rng.integers(1 << 30)), tasks=("phone",))
    b["phone"], b["phone_len"] = bp["phone"], bp["phone_len"]
    losses = model.step(b)
    try:
        ops.check_device_flag(dev)
    except RuntimeError:
        print("FAILED at step %d: B=%d T=%d t_dec=%d t_phone=%d" % (it, B, T, td, tp), flush=True)
        raise
    assert np.isfinite(float(losses["char"].item())) and np.isfinite(float(losses["phone"].item())), (it, B, T)
    if it % 50 == 0:
        print("step %4d B=%2d T=%3d  char %.3f phone %.3f (%.1f s)" % (it, B, T, losses["char"].item(), losses["phone"].item(), time.time() - t0), flush=True)
print("soak_multi ok: %d steps, nlp=%d, in %.1f s" % (n, nlp, time.time() - t0))
