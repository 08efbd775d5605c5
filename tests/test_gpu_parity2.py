"""GPU parity, second set: the shapes BASELINE's configs really run and the branches round 1 left unpinned.

* config 4 at its stated shape: phone decoder on layer-2 states (Te = 260 / 400 positions -> the one-utterance-per-group
  instantiation of the persistent decoder chains at H=256, D=512, A=128), logits + loss vs the float64 oracle and EVERY
  gradient vs the float64 autograd twin;
* config-2 decoder widths (two-utterance groups, H=256) gradients vs the autograd twin (not just vs the launch path);
* asr_beam_step vs the reference's own get_top_k step fixtures (scores, three new states, context; tests/golden);
* device beam selection with a word-insertion penalty vs the float64 oracle beam search;
* the Gumbel-max sampler (tf.multinomial stand-in, decoder.py:156-180) vs softmax probabilities (chi-square);
* encoder options that were coded but never run: initial_res_fac, stack_cons, the "state" tap;
* Decoder.prepare_decoder_input at the reference's signature; Eval.beam_search_decode end to end.
"""
import os

import numpy as np
import pytest
import torch

from oracle import asr_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _f64(w):
    return {k: np.asarray(v, np.float64) for k, v in w.items()}


def _model(params_update=None, enc_update=None, dec_update=None, tasks=("char",), num_layers=None,
           feat=20, training=True, vocab=None, seed=3, max_output=None):
    from e2e_asr_amd.seq2seq_model import Seq2SeqModel
    from e2e_asr_amd.attn_decoder import AttnDecoder
    p = Seq2SeqModel.class_params()
    p.tasks = list(tasks)
    p.num_layers = num_layers or {"char": 4}
    p.max_output = max_output or {"char": 12, "phone": 14}
    p.encoder_params.use_lstm = True
    p.encoder_params.out_prob = 1.0
    for k, v in (enc_update or {}).items():
        p.encoder_params[k] = v
    p.decoder_params = {}
    for t in tasks:
        dp = AttnDecoder.class_params()
        dp.out_prob_dec = 1.0
        dp.samp_prob = 0.0
        dp.vocab_size = (vocab or {"char": 50, "phone": 20})[t]
        for k, v in (dec_update or {}).items():
            dp[k] = v
        p.decoder_params[t] = dp
    for k, v in (params_update or {}).items():
        p[k] = v
    return Seq2SeqModel(None, isTraining=training, params=p, device=DEV, feat_length=feat, seed=seed)


def _batch(seed, B, T, F, tdec, vocab, lens=None, tasks=("char",)):
    from e2e_asr_amd.weights import synthetic_batch
    b = synthetic_batch(B=B, T=T, F=F, t_dec=tdec, vocab=vocab, variable_len=True, seed=seed, tasks=tasks)
    if lens is not None:
        b["logmel_len"] = np.asarray(lens, np.int64)
    return b


def _grad_check(m, b, tol=2e-3, outs=None, **kw):
    from oracle import torch_ref as R
    w = _f64(m.variables.to_arrays())
    W = R.weights_to_torch(w)
    b64 = dict(b); b64["logmel"] = b["logmel"].astype(np.float64)
    total, losses, logits = R.seq2seq_loss(b64, W, **kw)
    if outs is not None:           # the twin's forward (pinned to the NumPy oracle to 1e-12 by tests/test_oracle_torch_ref.py)
        outs.update(losses={t: float(v.item()) for t, v in losses.items()}, outputs={t: v.detach().numpy() for t, v in logits.items()})
    total.backward()
    worst = 0.0
    for name in m.variables.names():
        got = m.variables.grad_of(name).cpu().numpy()
        ref = W[name].grad.numpy()
        err = float(np.abs(got - ref).max()) / max(1e-3, float(np.abs(ref).max()))
        worst = max(worst, err)
        assert err < tol, (name, err)
    return float(total.item()), worst


# ------------------------------------------------------------------ config 4 at its stated shape
@pytest.mark.parametrize("Te", [260, 400])
@pytest.mark.parametrize("fwd_path,bwd_wide", [("one_launch", "1"), ("segment_chains", "1"), ("one_launch", "0")])
def test_config4_phone_decoder_on_layer2_states_real_widths(monkeypatch, Te, fwd_path, bwd_wide):
    """BASELINE config 4: char decoder on depth 4 + phone decoder (V=50) on LAYER-2 states, real widths (BiLSTM(256),
    decoder 256, A=128).  Te = T/2 > 256 encoder positions: the phone decoder's forward is the one-launch training kernel
    with 16 positions per workgroup (decoder_greedy_kernel<..., TRAIN, 16>, round 4), or -- ASR_DEC_GREEDY_TEMAX=256, the
    path of rounds 1-3 -- decoder_chain_fwd_kernel<256,512,128,R=1> per scheduled-sampling segment; the backward is
    decoder_chain_bwd_kernel<256,512,128,R=2,PASSES=2> (round 5: two utterances per group, their 2 x 25 positions per
    workgroup in two passes, hf / dhf in registers) or -- ASR_CHAIN_BWD_WIDE=0, rounds 2-4 -- <R=1>, one utterance per
    group.  Te = 400 is the 800-frame batch of the config.  Logits and
    losses vs the float64 oracle (seq2seq_model.py:88-144, losses averaged), every gradient vs float64 autograd."""
    from e2e_asr_amd import _lib, ops
    L = _lib.lib()
    T = 2 * Te
    if fwd_path == "segment_chains":
        monkeypatch.setenv("ASR_DEC_GREEDY_TEMAX", "256")
    monkeypatch.setenv("ASR_CHAIN_BWD_WIDE", bwd_wide)
    assert L.asr_decoder_chain_rows(Te) == 1 and L.asr_decoder_chain_supported(3, Te, 512, 128, 256) == 1
    assert L.asr_decoder_chain_bwd_rows(Te, 512, 128, 256) == (2 if bwd_wide == "1" else 1)
    assert L.asr_decoder_chain_bwd_rows(401, 512, 128, 256) == 1 and L.asr_decoder_chain_bwd_rows(256, 512, 128, 256) == 2
    assert L.asr_decoder_greedy_supported(3, Te, 512, 128, 256, 256, 256, 50) == (1 if fwd_path == "one_launch" else 0)
    tasks = ("char", "phone")
    nl = {"char": 4, "phone": 2}
    m = _model(tasks=tasks, num_layers=nl, feat=80, vocab={"char": 1000, "phone": 50}, seed=23)
    b = _batch(77 + Te, 3, T, 80, 9, 50, lens=[T, T - 37, T // 2 + 1], tasks=tasks)
    b["char"] = np.where(b["char"] > 2, b["char"] * 17 % 997 + 3, b["char"])           # spread char ids over V=1000
    m.forward(b)
    for t in tasks:
        assert m.decoder[t].saved["ws"].get("chain_ws") is not None                    # the persistent chains really ran
    assert (m.decoder["phone"].saved["ws"].get("greedy_ws") is not None) == (fwd_path == "one_launch")
    ops.check_device_flag(torch.device(DEV))
    assert m.encoder_hidden_states[2].shape[1] == Te
    got = {t: m.outputs[t].cpu().numpy() for t in tasks}
    got_loss = {t: m.losses[t].item() for t in tasks}
    m.backward()
    ops.check_device_flag(torch.device(DEV))
    if Te == 260:      # the NumPy oracle itself; at Te = 400 its float64 torch twin (one pass over the 800 frames instead of two)
        w = _f64(m.variables.to_arrays())
        b64 = dict(b); b64["logmel"] = b["logmel"].astype(np.float64)
        r = O.seq2seq_forward(b64, w, tasks=tasks, num_layers=nl, is_training=True)
        loss, worst = _grad_check(m, b, tasks=tasks, num_layers=nl)
    else:
        r = {}
        loss, worst = _grad_check(m, b, outs=r, tasks=tasks, num_layers=nl)
    for t in tasks:
        err = np.abs(got[t] - r["outputs"][t]).max()
        assert err < 1e-3, (t, err)                                                    # north-star tolerance
        np.testing.assert_allclose(got_loss[t], r["losses"][t], rtol=2e-5)
    np.testing.assert_allclose(m.total_loss.item(), loss, rtol=2e-5)
    print("config 4, Te=%d: worst relative gradient error %.2e" % (Te, worst))


@pytest.mark.parametrize("nb,T", [(6, 64), (5, 333)])
def test_config2_widths_decoder_gradients_vs_autograd(nb, T):
    """Config-2 architecture (H=256, D=512, A=128, E=256, V=1000, lm 256) on the two-utterance-per-group chain
    (decoder_chain_*_kernel<256,512,128,2>): every gradient against float64 autograd -- the H=256 backward chain meets
    the fp64 twin directly (round 1 compared it only with the launch path).  T=333: odd lengths at every pyramid level
    and 42 encoder positions (3 per workgroup)."""
    from e2e_asr_amd import ops
    m = _model(feat=80, vocab={"char": 1000}, seed=29, max_output={"char": 20})
    b = _batch(91 + nb, nb, T, 80, 11, 1000)
    m.forward(b)
    ws = m.decoder["char"].saved["ws"]
    assert ws.get("chain_ws") is not None and ws.get("lm_act") is not None
    m.backward()
    ops.check_device_flag(torch.device(DEV))
    loss, worst = _grad_check(m, b)
    np.testing.assert_allclose(m.total_loss.item(), loss, rtol=2e-5)
    print("config-2 widths B=%d T=%d: worst relative gradient error %.2e" % (nb, T, worst))


# ------------------------------------------------------------------ beam search step vs the reference's get_top_k
def _weights(g, prefix):
    return {k[len(prefix):]: g[k] for k in g.files if k.startswith(prefix)}


@pytest.mark.parametrize("variant", ["plain", "simple"])
@pytest.mark.parametrize("lm_weight", [0.0, 0.1])
@pytest.mark.parametrize("k", [1, 4, 16])
def test_beam_step_vs_reference_get_top_k_fixtures(golden_dir, variant, lm_weight, k):
    """asr_beam_gather + asr_beam_step on the inputs of the committed `step_lm*_k*` fixtures, which the REFERENCE's own
    get_top_k (beam_search.py:178-219) produced: sorted top-k indices exact, model scores / scores, the three new LSTM
    states and the new context within float32 tolerance.  The fixture feeds arbitrary embedding vectors; they are put
    into row TOK of the two embedding tables."""
    from e2e_asr_amd.beam_search import BeamSearch
    g = np.load(os.path.join(golden_dir, "decoder_step_%s.npz" % variant))
    wd, wl = _weights(g, "w_dec/"), _weights(g, "w_lm/")
    tag = "step_lm%g_k%d_" % (lm_weight, k)
    TOK, emb = 5, "model/rnn_decoder_char/decoder/embedding"
    wd[emb] = wd[emb].copy(); wd[emb][TOK] = g[tag + "x"]
    wl[emb] = wl[emb].copy(); wl[emb][TOK] = g[tag + "x_lm"]
    sp = BeamSearch.class_params()
    sp.beam_size = k; sp.lm_weight = lm_weight; sp.lm_path = wl
    bs = BeamSearch(wd, sp)
    step = bs.top_k_setup_with_lm(g["enc_T100"])
    sets = step.keep[0]
    fields = (("dc", "dec_c"), ("dh", "dec_h"), ("dlc", "declm_c"), ("dlh", "declm_h"), ("lc", "lm_c"), ("lh", "lm_h"))
    for f, name in fields:                                   # row 0 of the "previous step's output" = the parent
        sets[1][f][0] = torch.from_numpy(g[tag + name + "_in"].astype(np.float32)).to(DEV)
    sets[1]["ctx"][0] = torch.from_numpy(g[tag + "ctx_in"].astype(np.float32)).to(DEV)
    (idx, ms, sc), = step([TOK], [0], beam_size=k)
    order = np.argsort(idx)
    np.testing.assert_array_equal(idx[order], g[tag + "idx"])
    np.testing.assert_allclose(ms[order], g[tag + "model_score"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(sc[order], g[tag + "score"], rtol=0, atol=2e-5)
    for f, name in fields:
        np.testing.assert_allclose(sets[1][f][0].cpu().numpy(), g[tag + name + "_out"], rtol=0, atol=5e-6, err_msg=name)
    np.testing.assert_allclose(sets[1]["ctx"][0].cpu().numpy(), g[tag + "ctx_out"], rtol=0, atol=5e-6)


@pytest.mark.parametrize("k,wip,lm_weight", [(4, 0.3, 0.1), (16, 0.3, 0.1), (8, -0.2, 0.0), (16, 1.5, 0.1)])
def test_device_beam_selection_with_word_penalty_vs_oracle(golden_dir, monkeypatch, k, wip, lm_weight):
    """asr_beam_select (float64 scoring, top-k, parents, EOS bookkeeping and beam shrinking, word-insertion penalty on the
    carried score -- beam_search.py:290-327) against the float64 ORACLE beam search, V = 37 so that hypotheses do finish.
    wip = 1.5 makes long hypotheses win (the loop runs to 120 steps); -0.2 makes EOS attractive early."""
    from e2e_asr_amd.beam_search import BeamSearch
    monkeypatch.setenv("ASR_BEAM_HOST", "0")
    g = np.load(os.path.join(golden_dir, "decoder_step_plain.npz"))
    wd, wl = _weights(g, "w_dec/"), _weights(g, "w_lm/")
    sp = BeamSearch.class_params()
    sp.beam_size = k; sp.lm_weight = lm_weight; sp.lm_path = wl; sp.word_ins_penalty = wip
    bs = BeamSearch(wd, sp)
    rng = np.random.default_rng(13)
    for enc in (g["enc_T100"], g["enc_T100"][:37] * 1.5, (rng.standard_normal((64, 48)) * 0.5).astype(np.float32)):
        got = bs(enc)
        ref = O.beam_search(enc, wd, wl, beam_size=k, lm_weight=lm_weight, word_ins_penalty=wip)
        np.testing.assert_array_equal(got, ref)


# ------------------------------------------------------------------ sampler statistics
def test_gumbel_sampler_matches_softmax_distribution():
    """decoder.py:156-180: tf.multinomial(prev, 1) draws symbol v with probability softmax(prev)[v].  The device sampler is
    Gumbel-max over a counter-based generator; 64 rows x 400 calls = 25 600 draws from one logit vector must pass a
    chi-square goodness-of-fit test against softmax (V = 24, dof 23: 99.9 % quantile 49.7), rows and calls must be
    independent draws, and (seed, step) must reproduce."""
    from e2e_asr_amd import ops
    rng = np.random.default_rng(5)
    V, B, N = 24, 64, 400
    logit = rng.standard_normal(V) * 1.5
    logit[3] = -30.0                                          # an (almost) impossible symbol must never be drawn
    p = np.exp(logit - logit.max()); p /= p.sum()
    x = torch.from_numpy(np.tile(logit.astype(np.float32), (B, 1))).to(DEV)
    draws = np.stack([ops.next_token(x, sample=True, seed=1234, step=s).cpu().numpy() for s in range(N)])   # [N,B]
    counts = np.bincount(draws.reshape(-1), minlength=V).astype(np.float64)
    assert counts[3] == 0
    keep = p * B * N > 5
    chi2 = float((((counts - p * B * N) ** 2) / (p * B * N))[keep].sum())
    assert chi2 < 49.7 + 10, (chi2, counts, p * B * N)
    assert len(set(map(tuple, draws.T))) == B                 # rows are distinct sequences
    assert len(set(map(tuple, draws))) == N                   # calls are distinct
    again = ops.next_token(x, sample=True, seed=1234, step=7).cpu().numpy()
    np.testing.assert_array_equal(again, draws[7])
    other = ops.next_token(x, sample=True, seed=99, step=7).cpu().numpy()
    assert (other != draws[7]).any()
    # pairwise independence of consecutive calls on one row: chi-square on the 2x2 table "most likely symbol or not"
    top = int(np.argmax(p))
    a, bq = (draws[:-1] == top).reshape(-1), (draws[1:] == top).reshape(-1)
    n11 = float((a & bq).sum()); n = float(a.size); e11 = a.mean() * bq.mean() * n
    assert abs(n11 - e11) < 5 * np.sqrt(e11), (n11, e11)


def test_scheduled_sampling_feeds_sampled_tokens_with_softmax_frequencies():
    """The whole training-graph decoder with samp_prob = 1 (every step feeds a draw from the previous step's posterior,
    attn_decoder.py:131-139): at step 1 every row has the same state history when the batch repeats ONE utterance, so the
    tokens fed at step 1 are i.i.d. draws from softmax(logits[0]) -- checked by chi-square over 256 rows x 6 seeds."""
    from e2e_asr_amd import ops
    B = 256
    m = _model(enc_update=dict(hidden_size=64), num_layers={"char": 2}, vocab={"char": 12}, seed=31,
               dec_update=dict(hidden_size_dec=64, lm_hidden_size=64, emb_size=24, attention_vec_size=16, samp_prob=1.0))
    b1 = _batch(3, 1, 16, 20, 6, 12)
    b = {k: (np.repeat(np.asarray(v), B, axis=0) if hasattr(v, "__len__") else v) for k, v in b1.items()}
    counts = np.zeros(12)
    p = None
    for step in range(6):
        m.global_step = step
        m.forward(b)
        ops.check_device_flag(torch.device(DEV))
        ws = m.decoder["char"].saved["ws"]
        logits0 = m.outputs["char"][:B].double()
        if p is None:
            p = torch.softmax(logits0[0], 0).cpu().numpy()
        assert torch.allclose(logits0, logits0[0:1].expand_as(logits0), atol=1e-5)
        counts += np.bincount(ws["tok"][1].cpu().numpy(), minlength=12)
    n = counts.sum()
    keep = p * n > 5
    chi2 = float((((counts - p * n) ** 2) / (p * n))[keep].sum())
    assert chi2 < 31.3 + 8, (chi2, counts, p * n)              # dof <= 11: 99.9 % quantile 31.3


# ------------------------------------------------------------------ encoder options
def test_encoder_initial_res_fac_stack_cons_and_state_tap_vs_oracle():
    """encoder.py:149-153 (input stride `initial_res_fac`, len <- ceil(len/fac), counted against max_scaling_down),
    seq2seq_model.py:164-183 (`stack_cons` consecutive frames stacked on the feature axis, zero-padded at the end) and the
    time-major tap for a task literally named "state" (encoder.py:143-144,160-161)."""
    rng = np.random.default_rng(17)
    B, T, F = 4, 45, 12
    x = rng.standard_normal((B, T, F)).astype(np.float32)
    lens = np.array([45, 44, 9, 23])
    for bq in range(B):
        x[bq, lens[bq]:] = 0
    # (1) initial_res_fac = 2, max_scaling_down = 4: only ONE pyramid reduction happens (2 -> 4), layer 3 keeps the length
    m = _model(enc_update=dict(hidden_size=64, initial_res_fac=2, max_scaling_down=4), feat=F, num_layers={"char": 3},
               dec_update=dict(hidden_size_dec=64, lm_hidden_size=64, emb_size=24, attention_vec_size=16))
    att, tm, sl = m.encoder(torch.from_numpy(x).to(DEV), lens, {"char": 3, "state": 2})
    w = _f64(m.variables.to_arrays())
    ratt, rtm, rsl = O.encoder(x.astype(np.float64), lens, w, {"char": 3, "state": 2}, initial_res_fac=2, max_scaling_down=4)
    np.testing.assert_array_equal(sl[3], rsl[3]); np.testing.assert_array_equal(sl[2], rsl[2])
    assert att[3].shape == ratt[3].shape and sl[3].max() == rsl[2].max()                # no reduction after layer 2
    np.testing.assert_allclose(att[3].cpu().numpy(), ratt[3], rtol=0, atol=5e-5)
    assert tuple(tm[2].shape) == rtm[2].shape                                           # time-major [T_2, B, 2H]
    np.testing.assert_allclose(tm[2].cpu().numpy(), rtm[2], rtol=0, atol=5e-5)
    # (2) stack_cons = 3 through Seq2SeqModel.get_batch: the encoder sees [x_t | x_{t+1} | x_{t+2}] (zeros past the end)
    m2 = _model(enc_update=dict(hidden_size=64, stack_cons=3), feat=F, num_layers={"char": 2},
                dec_update=dict(hidden_size_dec=64, lm_hidden_size=64, emb_size=24, attention_vec_size=16))
    b = _batch(5, B, T, F, 7, 50, lens=lens)
    b["logmel"] = x
    m2.forward(b)
    xs = np.concatenate([x] + [np.concatenate([x[:, s:], np.zeros((B, s, F), np.float32)], 1) for s in (1, 2)], 2)
    assert tuple(m2.encoder_inputs.shape) == xs.shape
    np.testing.assert_array_equal(m2.encoder_inputs.cpu().numpy(), xs)
    b64 = dict(b); b64["logmel"] = xs.astype(np.float64)
    r = O.seq2seq_forward(b64, _f64(m2.variables.to_arrays()), num_layers={"char": 2}, is_training=True)
    np.testing.assert_allclose(m2.outputs["char"].cpu().numpy(), r["outputs"]["char"], rtol=0, atol=1e-4)


# ------------------------------------------------------------------ boundary: prepare_decoder_input
def test_prepare_decoder_input_signature_and_loop_functions():
    """decoder.py:84-115: prepare_decoder_input(decoder_inputs) -> (embedded_inp [T,B,E], loop_function); None under
    pure teacher forcing, the sampling closure with scheduled sampling, the arg-max closure in the inference graph
    (first maximum, as tf.argmax)."""
    rng = np.random.default_rng(2)
    ids = rng.integers(0, 50, (7, 3))
    for training, samp in ((True, 0.0), (True, 0.2), (False, 0.0)):
        m = _model(enc_update=dict(hidden_size=64), num_layers={"char": 2}, training=training,
                   dec_update=dict(hidden_size_dec=64, lm_hidden_size=64, emb_size=24, attention_vec_size=16, samp_prob=samp))
        dec = m.decoder["char"]
        emb = m.variables["model/rnn_decoder_char/decoder/embedding"]
        embedded, loop = dec.prepare_decoder_input(torch.from_numpy(ids).to(DEV))
        assert tuple(embedded.shape) == (7, 3, 24)
        assert torch.equal(embedded, emb[torch.from_numpy(ids).to(DEV)])
        if training and samp == 0.0:
            assert loop is None
            continue
        logits = torch.from_numpy(rng.standard_normal((3, 50)).astype(np.float32)).to(DEV)
        logits[1, 7] = logits[1, 30] = 9.0                                              # tie: the FIRST maximum wins
        out = loop(logits)
        assert tuple(out.shape) == (3, 24)
        if not training:
            want = logits.argmax(1); want[1] = 7
            assert torch.equal(out, emb[want])
        else:
            rows = [(emb == out[r]).all(1).nonzero().reshape(-1) for r in range(3)]
            assert all(len(r) == 1 for r in rows)                                       # each output row IS an embedding row
            out2 = loop(logits)                                                         # next call: fresh noise
            assert out2.shape == out.shape


# ------------------------------------------------------------------ Eval.beam_search_decode end to end
def test_eval_beam_search_decode_end_to_end(tmp_path):
    """eval_model.py:120-246: encoder pass over the dev batches (states cut to their lengths), batch-1 beam search per
    utterance, word-level scoring.  Ids must equal per-utterance BeamSearch.__call__ on the same states; the error rate
    and the insertion / deletion / substitution counts must equal the values computed by hand from those ids."""
    from e2e_asr_amd.beam_search import BeamSearch
    from e2e_asr_amd.eval_model import Eval, edit_ops
    from e2e_asr_amd.base_params import Bunch
    m = _model(enc_update=dict(hidden_size=64), num_layers={"char": 2}, training=False, vocab={"char": 14}, seed=41,
               dec_update=dict(hidden_size_dec=64, lm_hidden_size=64, emb_size=24, attention_vec_size=16),
               max_output={"char": 10})
    batches = [_batch(1, 3, 21, 20, 8, 14), _batch(2, 2, 16, 20, 8, 14, lens=[16, 9])]
    sp = BeamSearch.class_params(); sp.beam_size = 3
    bs = BeamSearch(m.variables.to_arrays(), sp, device=DEV)
    ev = Eval(m, params=Bunch(best_model_dir=str(tmp_path), vocab_dir=""))
    hidden, utt_ids, golds = ev.exec_encoder(batches)
    assert len(hidden) == 5 and [h.shape[0] for h in hidden] == [int(np.ceil(l / 2.0)) for bt in batches for l in bt["logmel_len"]]
    outs = [bs(h) for h in hidden]
    score, (ins, dele, sub) = ev.beam_search_decode(batches, bs, get_counts=True)
    err = n = i_ = d_ = s_ = 0
    for out, gold in zip(outs, golds):
        h, g = Eval.cut_at_eos(out), Eval.cut_at_eos(gold)
        d, a, b_, c = edit_ops(h, g)
        err += d; n += len(g); i_ += a; d_ += b_; s_ += c
    assert score == pytest.approx(err / float(n)) and (ins, dele, sub) == (i_, d_, s_)
    assert ins + dele + sub == err
    # with a vocabulary: sentences, word filtering and the reference's output files
    rev = [b"<pad>", b"<go>", b"<eos>"] + [(u"▁w%d" % i).encode("utf-8") for i in range(11)]
    ev2 = Eval(m, params=Bunch(best_model_dir=str(tmp_path), vocab_dir=""), rev_char_vocab=rev)
    score2 = ev2.beam_search_decode(batches, bs)
    assert 0.0 <= score2 and os.path.isfile(str(tmp_path / "gold.txt")) and os.path.isfile(str(tmp_path / "raw_3.txt"))
    assert len(open(str(tmp_path / "gold.txt")).read().strip().split("\n")) == 5


# ------------------------------------------------------------------ any -hsize; co-residency budget
@pytest.mark.parametrize("H,bi", [(96, True), (320, True), (40, False)])
def test_arbitrary_hidden_size_vs_oracle_and_autograd(H, bi):
    """encoder.py:188-189 takes any -hsize.  Widths the persistent recurrent kernels are not instantiated for run
    zero-padded to the next instantiated width (ops._pad_lstm_weights: exact -- padded units stay at h = c = 0): logits
    and loss vs the float64 oracle, every gradient vs float64 autograd."""
    from e2e_asr_amd import ops
    nl = {"char": 3}
    m = _model(enc_update=dict(hidden_size=H, bi_dir=bi), num_layers=nl, seed=37,
               dec_update=dict(hidden_size_dec=48, lm_hidden_size=40, emb_size=24, attention_vec_size=24))
    b = _batch(11 + H, 4, 29, 20, 8, 50)
    m.forward(b)
    ops.check_device_flag(torch.device(DEV))
    D = (2 if bi else 1) * H
    assert m.encoder_hidden_states[3].shape[2] == D
    w = _f64(m.variables.to_arrays())
    b64 = dict(b); b64["logmel"] = b["logmel"].astype(np.float64)
    r = O.seq2seq_forward(b64, w, num_layers=nl, bi_dir=bi, is_training=True)
    np.testing.assert_allclose(m.encoder_hidden_states[3].cpu().numpy(), r["enc"][3], rtol=0, atol=5e-5)
    np.testing.assert_allclose(m.outputs["char"].cpu().numpy(), r["outputs"]["char"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(m.total_loss.item(), r["total_loss"], rtol=2e-5)
    m.backward()
    ops.check_device_flag(torch.device(DEV))
    _grad_check(m, b, num_layers=nl, bi_dir=bi)


def test_persistent_grid_that_cannot_be_coresident_is_refused(monkeypatch):
    """The persistent kernels need every workgroup of a group resident at once.  The budget is the device's CU count
    (hipDeviceAttributeMultiprocessorCount), lowered here through ASR_LSTM_MAXWG to stand in for a partitioned device.  With
    8 workgroups a BiLSTM(256) still runs -- one row per launch in groups of four workgroups per direction (round 3; before,
    its 2 x 8-workgroup group had to be refused) -- and stays correct, as does the uni-directional layer; with 3 workgroups
    not even one group fits and both must be refused with ValueError (ASR_EUNSUPPORTED), not run into the 2-second exchange
    timeout; the decoder falls back to the per-step launch path below 16 workgroups."""
    from e2e_asr_amd import _lib, ops
    L = _lib.lib()
    full = L.asr_resident_wg_budget()
    assert full >= 64                                          # 256 on a whole MI355X
    rng = np.random.default_rng(3)
    B, T, IN, H = 3, 12, 16, 256
    x = torch.from_numpy(rng.standard_normal((B, T, IN)).astype(np.float32)).to(DEV)
    k = torch.from_numpy(rng.uniform(-0.1, 0.1, (IN + H, 4 * H)).astype(np.float32)).to(DEV)
    bz = torch.zeros(4 * H, device=DEV)
    ln = torch.tensor([12, 7, 1], dtype=torch.int32, device=DEV)
    want_uni = ops.lstm_layer_fwd(x, ln, k, bz)
    want_bi = ops.lstm_layer_fwd(x, ln, k, bz, k, bz)
    monkeypatch.setenv("ASR_LSTM_MAXWG", "8")
    monkeypatch.setenv("ASR_LSTM_G4", "1")                     # (whatever the environment of the run says)
    assert L.asr_resident_wg_budget() == 8
    got_bi = ops.lstm_layer_fwd(x, ln, k, bz, k, bz)           # 2 directions x 4 workgroups = one row per launch, three launches
    got_uni = ops.lstm_layer_fwd(x, ln, k, bz)
    ops.check_device_flag(torch.device(DEV))
    assert torch.allclose(got_uni, want_uni, atol=1e-6) and torch.allclose(got_bi, want_bi, atol=1e-6)
    monkeypatch.setenv("ASR_LSTM_G4", "0")                     # the eight-workgroup groups: the BiLSTM group does not fit
    with pytest.raises(ValueError):
        ops.lstm_layer_fwd(x, ln, k, bz, k, bz)
    monkeypatch.setenv("ASR_LSTM_G4", "1")
    monkeypatch.setenv("ASR_LSTM_MAXWG", "3")
    for args in ((x, ln, k, bz, k, bz), (x, ln, k, bz)):
        with pytest.raises(ValueError):
            ops.lstm_layer_fwd(*args)
    monkeypatch.setenv("ASR_LSTM_MAXWG", "8")
    assert L.asr_decoder_chain_supported(4, 10, 512, 128, 256) == 0 and L.asr_decoder_greedy_supported(4, 10, 512, 128, 256, 256, 256, 1000) == 0
    monkeypatch.delenv("ASR_LSTM_MAXWG")
    assert L.asr_resident_wg_budget() == full and L.asr_decoder_chain_supported(4, 10, 512, 128, 256) == 1


# ------------------------------------------------------------------ training graph in one persistent launch
@pytest.mark.parametrize("nb,T", [(32, 48), (37, 96), (3, 512), (3, 800), (6, 640), (32, 800)])
def test_training_decoder_one_launch_equals_segment_chain_path(monkeypatch, nb, T):
    """csrc/decoder_greedy.hip, TRAIN instantiation: the whole training-graph decoder (teacher forcing, scheduled-sampling
    feedback with Gumbel draws at the coin-selected steps, LM dropout, saved activations) in ONE persistent launch must give
    the segment-wise chain path's sampled tokens, logits, loss and every gradient (the backward consumes the activations
    the kernel saved).  32 utterances = 8 one-XCD groups = 256 workgroups; 37 = two launches; T=512 at depth 2 = 256 encoder
    positions (8 per workgroup); T=800 / 640 at depth 2 = 400 / 320 positions: the instantiation with 16 positions per
    workgroup (13 / 10 used: four / three quads of scores per source workgroup), config 4's phone memory, incl. its full batch
    of 32.  Repeated runs: race detector at full occupancy."""
    from e2e_asr_amd import ops
    kw = dict(feat=80, vocab={"char": 1000}, num_layers={"char": 3 if T < 512 else 2}, seed=13, max_output={"char": 14},
              enc_update=dict(hidden_size=256, out_prob=0.9),
              dec_update=dict(hidden_size_dec=256, lm_hidden_size=256, emb_size=256, attention_vec_size=128, samp_prob=0.3,
                              out_prob_dec=0.9))
    b = _batch(51 + nb, nb, T, 80, 15, 1000)

    def run(traink):
        monkeypatch.setenv("ASR_DEC_TRAINK", traink)
        m = _model(**kw)
        m.decoder["char"].coin_seed = 8
        m.global_step = 1
        m.forward(b)
        ws = m.decoder["char"].saved["ws"]
        assert (ws.get("greedy_ws") is not None) == (traink == "1") and ws.get("chain_ws") is not None
        out, tok, loss = m.outputs["char"].cpu().numpy().copy(), ws["tok"].cpu().numpy().copy(), m.total_loss.item()
        m.backward()
        ops.check_device_flag(torch.device(DEV))
        return out, tok, loss, {n: m.variables.grad_of(n).cpu().numpy().copy() for n in m.variables.names()}

    ref = run("0")
    assert (ref[1][1:] != np.asarray(b["char"]).T[1:ref[1].shape[0]]).any()            # some tokens really were sampled
    for rep in range(3 if nb >= 32 else 1):
        got = run("1")
        np.testing.assert_array_equal(got[1], ref[1])
        np.testing.assert_allclose(got[0], ref[0], rtol=0, atol=5e-5)
        np.testing.assert_allclose(got[2], ref[2], rtol=1e-6)
        for n, g0 in ref[3].items():
            err = np.abs(got[3][n] - g0).max() / max(1e-3, np.abs(g0).max())
            assert err < 2e-4, (rep, n, err)


def test_training_decoder_one_launch_teacher_forced_vs_oracle_and_autograd():
    """The same kernel under pure teacher forcing (no feedback step at all) and ragged target / encoder lengths: logits
    and loss vs the float64 oracle, every gradient vs float64 autograd."""
    from e2e_asr_amd import ops
    m = _model(feat=80, vocab={"char": 1000}, seed=31, max_output={"char": 20})
    b = _batch(7, 7, 100, 80, 17, 1000)
    m.forward(b)
    assert m.decoder["char"].saved["ws"].get("greedy_ws") is not None
    w = _f64(m.variables.to_arrays())
    b64 = dict(b); b64["logmel"] = b["logmel"].astype(np.float64)
    r = O.seq2seq_forward(b64, w, is_training=True)
    np.testing.assert_allclose(m.outputs["char"].cpu().numpy(), r["outputs"]["char"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(m.total_loss.item(), r["total_loss"], rtol=1e-5)
    m.backward()
    ops.check_device_flag(torch.device(DEV))
    _grad_check(m, b)


# ------------------------------------------------------------------ MultiRNNCell decoder (num_layers_dec > 1)
@pytest.mark.parametrize("L,keep,lmH", [(2, 1.0, 32), (3, 0.8, 32), (2, 0.8, 24), (3, 1.0, 40)])
def test_multi_layer_decoder_vs_oracle_and_autograd(L, keep, lmH):
    """decoder.py:66-68, 77-78: `-num_layers_dec L` builds both decoder cells as MultiRNNCell stacks of DropoutWrapper(
    BasicLSTMCell) layers; the attention query is the TOP layer's c.  Logits and loss vs the float64 oracle and every gradient
    vs float64 autograd, ragged lengths; with dropout the per-layer masks of the counter-based generator are reproduced on
    the host and handed to the autograd twin."""
    from e2e_asr_amd import ops
    from e2e_asr_amd.multi_decoder import layer_seed
    from oracle import torch_ref as R
    from tests.test_gpu_model import _np_keep_scale
    nl = {"char": 2}
    m = _model(enc_update=dict(hidden_size=64), num_layers=nl, seed=43,
               dec_update=dict(hidden_size_dec=32, lm_hidden_size=lmH, emb_size=24, attention_vec_size=16, num_layers_dec=L,
                               out_prob_dec=keep))
    assert m.decoder["char"].cell.startswith("MultiRNNCell")
    # lmH != 32: the LM stack's top output passes through rnn/SimpleProjection (attn_decoder.py:149-151 with decoder.py:66-68)
    assert (lmH != 32) == any("SimpleProjection" in k for k in m.variables.names())
    b = _batch(61 + L, 5, 22, 20, 9, 50)
    m.global_step = 2
    m.forward(b)
    ops.check_device_flag(torch.device(DEV))
    B = 5
    T_out = m.decoder["char"].saved["t_out"]
    seed = m.decoder["char"].saved["seed"]
    w = _f64(m.variables.to_arrays())
    assert any("multi_rnn_cell_1/cell_%d/" % (L - 1) in k for k in w)
    lm_masks = dec_masks = None
    if keep < 1.0:
        def masks(stack, nlayers):
            out = []
            for k in range(nlayers):
                # (the oracle's raw_rnn restatement also runs the LM stack once past the last step: T_out + 1 rows)
                ii, bb, jj = np.meshgrid(np.arange(T_out + 1), np.arange(B), np.arange(lmH if stack == "lm" else 32), indexing="ij")
                out.append(_np_keep_scale(layer_seed(seed, stack, k), ii * B + bb, jj, keep))
            return out
        lm_masks, dec_masks = masks("lm", L), masks("dec", L - 1)
    b64 = dict(b); b64["logmel"] = b["logmel"].astype(np.float64)
    att, _, lens = O.encoder(b64["logmel"], b64["logmel_len"], w, nl)
    ref = O.attn_decoder(np.transpose(b["char"]), b["char_len"], att[2], lens[2], w, is_training=True,
                         lm_keep_masks=lm_masks, dec_keep_masks=dec_masks)
    np.testing.assert_allclose(m.outputs["char"].cpu().numpy(), ref, rtol=0, atol=1e-4)
    m.backward()
    ops.check_device_flag(torch.device(DEV))
    W = R.weights_to_torch(w)
    t = lambda ms: None if ms is None else [torch.tensor(x) for x in ms]
    total, _, _ = R.seq2seq_loss(b64, W, num_layers=nl, lm_keep_masks=None if lm_masks is None else {"char": t(lm_masks)},
                                 dec_keep_masks=None if dec_masks is None else {"char": t(dec_masks)})
    np.testing.assert_allclose(m.total_loss.item(), total.item(), rtol=2e-5)
    total.backward()
    for name in m.variables.names():
        ref_g = W[name].grad.numpy()
        err = np.abs(m.variables.grad_of(name).cpu().numpy() - ref_g).max() / max(1e-3, np.abs(ref_g).max())
        assert err < 2e-3, (name, err)


def test_multi_layer_decoder_inference_and_sampling_modes():
    """The same stacks in the inference graph (argmax feedback every step: ids equal to the float64 oracle's) and under
    scheduled sampling (runs, feeds drawn tokens, stays finite)."""
    from e2e_asr_amd import ops
    dec = dict(hidden_size_dec=32, lm_hidden_size=32, emb_size=24, attention_vec_size=16, num_layers_dec=2)
    m = _model(enc_update=dict(hidden_size=64), num_layers={"char": 2}, seed=47, training=False, dec_update=dec,
               max_output={"char": 9})
    b = _batch(71, 4, 18, 20, 8, 50)
    out = m.forward(b)["char"].cpu().numpy()
    w = _f64(m.variables.to_arrays())
    b64 = dict(b); b64["logmel"] = b["logmel"].astype(np.float64)
    r = O.seq2seq_forward(b64, w, num_layers={"char": 2}, is_training=False, max_output={"char": 9})["outputs"]["char"]
    np.testing.assert_allclose(out, r, rtol=0, atol=1e-4)
    np.testing.assert_array_equal(m.greedy_ids().cpu().numpy(), O.greedy_decode_ids(r, 4))
    m2 = _model(enc_update=dict(hidden_size=64), num_layers={"char": 2}, seed=47, dec_update=dict(dec, samp_prob=0.5))
    m2.decoder["char"].coin_seed = 3
    m2.step(b)
    ops.check_device_flag(torch.device(DEV))
    fed = m2.decoder["char"]  # saved was consumed by backward; the step ran end to end
    assert torch.isfinite(m2.variables.flat).all() and m2.global_step == 1
