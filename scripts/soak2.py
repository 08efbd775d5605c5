"""Soak of the paths added late in round 1, random shapes, every result cross-checked against the slower path of the same
library: (a) inference graph: one-launch greedy decoder vs per-step launches (token ids equal, logits within 2e-5);
(b) training graph at the depth-2 tap (257..512 encoder positions: one-utterance groups) -- chain vs launch path, loss
within 1e-5 relative and finite gradients; (c) beam search: device-resident selection vs host scoring (ids equal).
A mismatch or an exchange time-out aborts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from e2e_asr_amd import ops
from e2e_asr_amd.seq2seq_model import Seq2SeqModel
from e2e_asr_amd.beam_search import BeamSearch
from e2e_asr_amd.weights import synthetic_batch, init_weights

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
t0 = time.time()


def model(training, nl):
    p = Seq2SeqModel.class_params()
    p.encoder_params.use_lstm = True
    p.num_layers = {"char": nl}
    p.max_output = {"char": int(rng.integers(3, 40))}
    p.encoder_params.out_prob = 1.0
    p.decoder_params["char"].out_prob_dec = 1.0 if isinstance(p.decoder_params, dict) and "char" in p.decoder_params else 1.0
    return Seq2SeqModel(None, isTraining=training, params=p, device=dev, feat_length=80, seed=int(rng.integers(1 << 20)))


ev = model(False, 4)
tr = model(True, 2)
wd = {k: v for k, v in init_weights(seed=3).items() if "rnn_decoder_char" in k}
wl = {k: v for k, v in init_weights(seed=4).items() if "rnn_decoder_char" in k}
for it in range(n):
    # (a) greedy decode
    B = int(rng.integers(1, 41)); T = int(rng.integers(8, 400))
    b = synthetic_batch(B=B, T=T, F=80, t_dec=5, vocab=1000, variable_len=True, seed=int(rng.integers(1 << 30)))
    res = []
    for g in ("1", "0"):
        os.environ["ASR_DEC_GREEDY"] = g
        ev.forward(b)
        ops.check_device_flag(dev)
        res.append((ev.greedy_ids().cpu().numpy(), ev.outputs["char"].cpu().numpy()))
    ids_g, ids_l = res[0][0], res[1][0]                       # [B, T_out]
    lg = res[0][1].reshape(-1, B, 1000); ll = res[1][1].reshape(-1, B, 1000)
    for bb in range(B):
        diff = np.nonzero(ids_g[bb] != ids_l[bb])[0]
        t_ok = len(ids_g[bb]) if len(diff) == 0 else int(diff[0])
        # identical history up to t_ok: logits must agree there (and including the step where the argmax flipped)
        upto = min(t_ok + 1, lg.shape[0])
        assert np.abs(lg[:upto, bb] - ll[:upto, bb]).max() < 2e-5, ("greedy logits", it, B, T, bb)
        if len(diff):      # a flip is legitimate only between two candidates the per-step path itself cannot separate
            ta, tb = int(ids_g[bb, t_ok]), int(ids_l[bb, t_ok])
            gap = abs(float(ll[t_ok, bb, ta]) - float(ll[t_ok, bb, tb]))
            assert gap < 2e-5, ("greedy ids", it, B, T, bb, t_ok, ta, tb, gap)
            print("   near-tie flip: iter %d row %d step %d tokens %d/%d logit gap %.2e" % (it, bb, t_ok, ta, tb, gap), flush=True)
    # (b) long-encoder chains, training graph
    B = int(rng.integers(1, 20)); T = int(rng.integers(514, 1024)); td = int(rng.integers(4, 20))
    b = synthetic_batch(B=B, T=T, F=80, t_dec=td, vocab=1000, variable_len=True, seed=int(rng.integers(1 << 30)))
    tr.decoder["char"].params.samp_prob = 0.2
    lo = []
    for c in ("1", "0"):
        os.environ["ASR_DEC_CHAIN"] = c
        tr.decoder["char"].coin_seed = it
        tr.forward(b); tr.backward()
        ops.check_device_flag(dev)
        lo.append((float(tr.total_loss.item()), tr.variables.grad.clone()))
    assert abs(lo[0][0] - lo[1][0]) <= 1e-5 * abs(lo[1][0]), ("chain loss", it, B, T, lo[0][0], lo[1][0])
    gd = (lo[0][1] - lo[1][1]).abs().max().item() / max(1e-6, lo[1][1].abs().max().item())
    assert np.isfinite(gd) and gd < 1e-3, ("chain grads", it, B, T, gd)
    # (c) beam search
    if it % 3 == 0:
        sp = BeamSearch.class_params()
        sp.beam_size = int(rng.integers(1, 17)); sp.lm_weight = float(rng.choice([0.0, 0.1, 0.3])); sp.lm_path = wl
        sp.word_ins_penalty = float(rng.choice([0.0, 0.2]))
        bs = BeamSearch(wd, sp)
        enc = (rng.standard_normal((int(rng.integers(1, 200)), 512)) * 0.3).astype(np.float32)
        os.environ["ASR_BEAM_HOST"] = "0"; d_ids = bs(enc)
        os.environ["ASR_BEAM_HOST"] = "1"; h_ids = bs(enc)
        assert np.array_equal(d_ids, h_ids), ("beam", it, sp.beam_size, enc.shape)
    if it % 5 == 0:
        print("iter %3d ok  (%.1f s)" % (it, time.time() - t0), flush=True)
print("soak2 ok: %d iterations in %.1f s" % (n, time.time() - t0))
