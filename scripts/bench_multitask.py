"""BASELINE config 4 (per-GPU workload): config 2 + an auxiliary phone decoder (V = 50, up to 250 output steps) on the
encoder states of depth `-nlp` (default 3, seq2seq_model.py:206).  Train-step time on one MI355X.  Diagnostic."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from e2e_asr_amd import ops
from e2e_asr_amd.seq2seq_model import Seq2SeqModel
from e2e_asr_amd.attn_decoder import AttnDecoder
from e2e_asr_amd.weights import synthetic_batch

nlp = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
p = Seq2SeqModel.class_params()
p.encoder_params.use_lstm = True
p.tasks = ["char", "phone"]
p.num_layers = {"char": 4, "phone": nlp}
dp = AttnDecoder.class_params(); dp.vocab_size = 50
p.decoder_params = {"char": p.decoder_params["char"] if isinstance(p.decoder_params, dict) and "char" in p.decoder_params else AttnDecoder.class_params(),
                    "phone": dp}
model = Seq2SeqModel(None, isTraining=True, params=p, device=dev, feat_length=80, seed=10)
b = synthetic_batch(B=32, T=800, F=80, t_dec=121, vocab=1000, variable_len=False, seed=1234, tasks=("char",))
bp = synthetic_batch(B=32, T=800, F=80, t_dec=251, vocab=50, variable_len=False, seed=99, tasks=("phone",))
b["phone"], b["phone_len"] = bp["phone"], bp["phone_len"]
b = {k: (torch.as_tensor(v).to(dev) if k == "logmel" else v) for k, v in b.items()}
for _ in range(4):
    model.step(b)
torch.cuda.synchronize(); ops.check_device_flag(dev)
n = 10
t0 = time.perf_counter()
for _ in range(n):
    losses = model.step(b)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("config 4 (char depth 4 + phone depth %d): %.2f ms/step = %.2f M frames/s; losses char %.3f phone %.3f" % (
    nlp, dt * 1e3, 32 * 800 / dt / 1e6, losses["char"].item(), losses["phone"].item()))
