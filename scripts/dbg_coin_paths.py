import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_model import _model, _batch
rng = np.random.default_rng(51)
kw = dict(feat=80, vocab={"char": 1000}, num_layers={"char": 3}, seed=13, params_update=dict(max_output={"char": 14}),
          enc_update=dict(hidden_size=256, out_prob=float(os.environ.get("ENC_KEEP", "0.9"))),
          dec_update=dict(hidden_size_dec=256, lm_hidden_size=256, emb_size=256, attention_vec_size=128, samp_prob=0.3,
                          out_prob_dec=float(os.environ.get("DEC_KEEP", "0.9"))))
b = _batch(rng, 32, 48, 80, 15, 1000)
for sd in (8, 10):
    for mode in ("launch", "traink", "segchain"):
        os.environ["ASR_DEC_CHAIN"] = "0" if mode == "launch" else "1"
        os.environ["ASR_LM_CHAIN"] = os.environ["ASR_DEC_CHAIN"]
        os.environ["ASR_DEC_TRAINK"] = "0" if mode == "segchain" else "1"
        toks, encs = [], []
        for rep in range(5):
            m = _model(**kw)
            m.decoder["char"].coin_seed = sd
            m.global_step = 1
            m.forward(b)
            torch.cuda.synchronize()
            toks.append(m.decoder["char"].saved["ws"]["tok"].cpu().numpy().copy())
            encs.append(m.encoder_hidden_states[3].cpu().numpy().copy())
        diffs = [np.argwhere(t != toks[0]).tolist() for t in toks[1:]]
        print("seed", sd, mode, "token diffs vs rep 0:", diffs, "| encoder bit-equal:", [bool((e == encs[0]).all()) for e in encs[1:]], flush=True)
