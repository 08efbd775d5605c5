"""GPU parity, third set (round 3).

* the bench workload's BACKWARD at its own size: config 2, B = 32, T = 800, 120 target tokens, ragged lengths -- every
  gradient against float64 autograd through the torch twin of the oracle (the paths that exist only at this size meet the
  twin here: 512-workgroup split-K weight gradients at K = 25 600, 16 BPTT groups x 800 steps, side-stream joins);
* `-ind_softmax` (attn_decoder.py:119-125): the decoder's own softmax `rnn/OutputProjection2`, logits + loss vs the oracle,
  every gradient vs autograd, training and inference graphs, and an LM step that must leave it untouched.
"""
import numpy as np
import pytest
import torch

from oracle import asr_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _f64(w):
    return {k: np.asarray(v, np.float64) for k, v in w.items()}


def _model(params_update=None, enc_update=None, dec_update=None, tasks=("char",), num_layers=None,
           feat=20, training=True, vocab=None, seed=3, max_output=None, variables=None):
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
    return Seq2SeqModel(None, isTraining=training, params=p, device=DEV, feat_length=feat, seed=seed, variables=variables)


def _grad_check(m, b, tol=2e-3, unused=(), **kw):
    """Every variable's gradient vs float64 autograd; variables in `unused` must be absent from the twin's graph AND carry an
    all-zero gradient on the device."""
    from oracle import torch_ref as R
    W = R.weights_to_torch(_f64(m.variables.to_arrays()))
    b64 = dict(b); b64["logmel"] = b["logmel"].astype(np.float64)
    total, _, _ = R.seq2seq_loss(b64, W, **kw)
    total.backward()
    worst, worst_name = 0.0, None
    for name in m.variables.names():
        got = m.variables.grad_of(name).cpu().numpy()
        if name in unused:
            assert W[name].grad is None and not got.any(), name
            continue
        ref = W[name].grad.numpy()
        err = float(np.abs(got - ref).max()) / max(1e-3, float(np.abs(ref).max()))
        if err > worst:
            worst, worst_name = err, name
        assert err < tol, (name, err)
    return float(total.item()), worst, worst_name


# ------------------------------------------------------------------ the bench's backward at the bench's size
def test_config2_full_size_gradients_vs_autograd():
    """BASELINE config 2 at FULL size -- B = 32, T = 800, F = 80, 120 target tokens, ragged input and target lengths: every
    gradient of the HIP backward (seq2seq_model.py:137-151 `tf.gradients`) against float64 autograd of the oracle's torch
    twin, relative to each variable's largest gradient entry.  Needs ~1 min of host CPU for the twin."""
    import time
    from e2e_asr_amd import ops
    from e2e_asr_amd.weights import synthetic_batch
    m = _model(feat=80, vocab={"char": 1000}, max_output={"char": 120}, seed=17)
    b = synthetic_batch(B=32, T=800, F=80, t_dec=121, vocab=1000, variable_len=True, seed=4321)
    m.forward(b)
    m.backward()
    ops.check_device_flag(torch.device(DEV))
    t0 = time.time()
    nthr = torch.get_num_threads()
    try:
        import os
        torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
        loss, worst, name = _grad_check(m, b)
    finally:
        torch.set_num_threads(nthr)
    np.testing.assert_allclose(m.total_loss.item(), loss, rtol=1e-5)
    print("config-2 full size: worst relative gradient error %.2e (%s), twin took %.0f s" % (worst, name, time.time() - t0))


# ------------------------------------------------------------------ -ind_softmax
IND = dict(hidden_size_dec=64, lm_hidden_size=64, emb_size=24, attention_vec_size=16, ind_softmax=True)
PRE = "model/rnn_decoder_char/"


@pytest.mark.parametrize("chain", ["1", "0"])
def test_ind_softmax_logits_loss_and_gradients(monkeypatch, chain):
    """attn_decoder.py:119-125: with -ind_softmax the decoder projects through `rnn/OutputProjection2`; the variable
    `rnn/OutputProjection` (the char LM's softmax, lm_encoder.py:108-109) exists next to it, is not read, and receives a
    zero gradient.  Logits and loss vs the float64 oracle, every gradient vs autograd; persistent chains and launch path."""
    from e2e_asr_amd import ops
    from e2e_asr_amd.weights import synthetic_batch
    monkeypatch.setenv("ASR_DEC_CHAIN", chain)
    m = _model(enc_update=dict(hidden_size=64), dec_update=IND, num_layers={"char": 3}, seed=41)
    names = m.variables.names()
    assert PRE + "rnn/OutputProjection2/kernel" in names and PRE + "rnn/OutputProjection/kernel" in names
    w32 = m.variables.to_arrays()
    assert not np.array_equal(w32[PRE + "rnn/OutputProjection2/kernel"], w32[PRE + "rnn/OutputProjection/kernel"])
    b = synthetic_batch(B=5, T=37, F=20, t_dec=11, vocab=50, variable_len=True, seed=77)
    m.forward(b)
    ops.check_device_flag(torch.device(DEV))
    b64 = dict(b); b64["logmel"] = b["logmel"].astype(np.float64)
    r = O.seq2seq_forward(b64, _f64(w32), num_layers={"char": 3}, is_training=True, ind_softmax=True)
    err = np.abs(m.outputs["char"].cpu().numpy() - r["outputs"]["char"]).max()
    assert err < 1e-4, err
    np.testing.assert_allclose(m.total_loss.item(), r["total_loss"], rtol=2e-5)
    # the other reading (shared softmax) must NOT match: the flag really switches the variable
    r_shared = O.seq2seq_forward(b64, _f64(w32), num_layers={"char": 3}, is_training=True)
    assert np.abs(m.outputs["char"].cpu().numpy() - r_shared["outputs"]["char"]).max() > 1e-2
    m.backward()
    ops.check_device_flag(torch.device(DEV))
    loss, worst, _ = _grad_check(m, b, num_layers={"char": 3}, ind_softmax={"char": True},
                                 unused=(PRE + "rnn/OutputProjection/kernel", PRE + "rnn/OutputProjection/bias"))
    np.testing.assert_allclose(m.total_loss.item(), loss, rtol=2e-5)
    before = m.variables.to_arrays()
    m.apply_gradients()
    after = m.variables.to_arrays()
    for leaf in ("kernel", "bias"):
        np.testing.assert_array_equal(after[PRE + "rnn/OutputProjection/" + leaf], before[PRE + "rnn/OutputProjection/" + leaf])
    assert not np.array_equal(after[PRE + "rnn/OutputProjection2/kernel"], before[PRE + "rnn/OutputProjection2/kernel"])


def test_ind_softmax_inference_graph_vs_oracle():
    """The inference graph (argmax feedback, eval_model.py:56-118) under -ind_softmax: tokens identical to the oracle's."""
    from e2e_asr_amd.weights import synthetic_batch
    m = _model(enc_update=dict(hidden_size=64), dec_update=IND, num_layers={"char": 3}, seed=43, training=False)
    b = synthetic_batch(B=4, T=33, F=20, t_dec=11, vocab=50, variable_len=True, seed=78)
    m.forward(b)
    b64 = dict(b); b64["logmel"] = b["logmel"].astype(np.float64)
    r = O.seq2seq_forward(b64, _f64(m.variables.to_arrays()), num_layers={"char": 3}, is_training=False,
                          max_output={"char": 12}, ind_softmax=True)
    got = m.outputs["char"].cpu().numpy()
    assert np.abs(got - r["outputs"]["char"]).max() < 1e-4
    np.testing.assert_array_equal(O.greedy_decode_ids(got, 4), O.greedy_decode_ids(r["outputs"]["char"], 4))


def test_lm_step_leaves_independent_softmax_untouched():
    """lm_model.py:102-103 + attn_decoder.py:119-125: the char LM trains `rnn/OutputProjection`; the decoder's
    `rnn/OutputProjection2` is not in its graph -- zero gradient, bit-identical after the AdamLM step."""
    from e2e_asr_amd.lm_encoder import LMEncoder
    from e2e_asr_amd.lm_model import LMModel
    rng = np.random.default_rng(5)
    m = _model(enc_update=dict(hidden_size=64), dec_update=IND, num_layers={"char": 2}, vocab={"char": 31}, seed=47)
    ep = LMEncoder.class_params()
    ep.out_prob = 1.0; ep.lm_hidden_size = 64; ep.proj_size = 64; ep.emb_size = 24; ep.vocab_size = 31
    lm = LMModel(LMEncoder(isTraining=True, params=ep, variables=m.variables))
    B, T = 6, 9
    lens = np.array([9, 3, 7, 1, 9, 5])
    ids = np.zeros((B, T + 1), np.int64)
    for bb in range(B):
        ids[bb, :lens[bb] + 1] = rng.integers(1, 31, lens[bb] + 1)
    before = m.variables.to_arrays()
    lm.step({"char": ids, "char_len": lens})
    after = m.variables.to_arrays()
    changed = sorted(k for k in after if not np.array_equal(after[k], before[k]))
    assert changed == sorted(PRE + l for l in ("decoder/embedding", "rnn/basic_lstm_cell/kernel", "rnn/basic_lstm_cell/bias",
                                               "rnn/OutputProjection/kernel", "rnn/OutputProjection/bias"))
    for leaf in ("kernel", "bias"):
        assert not m.variables.grad_of(PRE + "rnn/OutputProjection2/" + leaf).any()


# ------------------------------------------------------------------ run-to-run stability at full occupancy
@pytest.mark.parametrize("slabs", [True, False])
def test_train_step_is_reproducible_run_to_run(slabs):
    """Three forward + backward passes from identical weights at config-2 widths and B = 32 (256-workgroup persistent launches,
    side-stream GEMMs co-running): logits bit-identical; every gradient BIT-IDENTICAL in the deterministic mode (ops.set_wgrad_mode(True): split-K weight
    gradients through slabs + a fixed-order reduce, two-stage bias sums, ordered embedding scatter: csrc/splitk.hip), equal to
    1e-5 of its largest entry with float atomics (the default), and the two modes equal to that
    tolerance.  A variant of the BPTT that prefetched by LDS-DMA was bit-stable ALONE and 1 % off run to run inside the step:
    only a test at this level sees that class of race.  (tf.gradients on one CPU thread is deterministic: seq2seq_model.py:148.)"""
    from e2e_asr_amd import ops
    from e2e_asr_amd.weights import synthetic_batch
    b = synthetic_batch(B=32, T=160, F=80, t_dec=21, vocab=1000, variable_len=True, seed=100)
    runs = []
    ops.set_wgrad_mode(slabs)
    try:
        for _ in range(3):
            m = _model(feat=80, vocab={"char": 1000}, max_output={"char": 30}, seed=6)
            m.forward(b); m.backward()
            torch.cuda.synchronize()
            ops.check_device_flag(torch.device(DEV))
            runs.append((m.outputs["char"].cpu().numpy().copy(), {n: m.variables.grad_of(n).cpu().numpy().copy() for n in m.variables.names()}))
        ops.set_wgrad_mode(not slabs)
        m = _model(feat=80, vocab={"char": 1000}, max_output={"char": 30}, seed=6)
        m.forward(b); m.backward()
        torch.cuda.synchronize()
        other = {n: m.variables.grad_of(n).cpu().numpy().copy() for n in m.variables.names()}
    finally:
        ops.set_wgrad_mode(False)
    for out, grads in runs[1:]:
        np.testing.assert_array_equal(out, runs[0][0])
        for n, g in grads.items():
            ref = runs[0][1][n]
            if slabs:
                np.testing.assert_array_equal(g, ref, err_msg=n)
            else:
                assert np.abs(g - ref).max() <= 1e-5 * max(1e-30, np.abs(ref).max()), n
    for n, g in other.items():
        ref = runs[0][1][n]
        assert np.abs(g - ref).max() <= 1e-5 * max(1e-30, np.abs(ref).max()), n
