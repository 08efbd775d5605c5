"""CPU: the torch-autograd twin (oracle/torch_ref.py) must reproduce the pinned NumPy oracle's
forward, so its gradients are gradients of the pinned function.  Also encoder property tests
that pin the restated TF semantics."""
import numpy as np
import pytest
import torch

from e2e_asr_amd.weights import init_weights, synthetic_batch
from oracle import asr_oracle as O
from oracle import torch_ref as R


def _small(tasks=("char",), bi_dir=True, depth=3, lm_hidden=12):
    w = init_weights(feat=10, hidden=8, bi_dir=bi_dir, depth=depth, tasks=tasks, vocab={"char": 17, "phone": 9},
                     emb=12, hidden_dec=12, lm_hidden=lm_hidden, attn_vec=8, seed=1)
    for k in w:      # non-zero biases so bias paths are exercised
        if k.endswith("bias"):
            w[k] = np.random.default_rng(len(k)).uniform(-0.2, 0.2, w[k].shape).astype(np.float32)
    return {k: v.astype(np.float64) for k, v in w.items()}


@pytest.mark.parametrize("tasks,bi,lmh", [(("char",), True, 12), (("char", "phone"), True, 12), (("char",), False, 7)])
def test_torch_ref_matches_numpy_oracle(tasks, bi, lmh):
    w = _small(tasks, bi, lm_hidden=lmh)
    b = synthetic_batch(B=3, T=13, F=10, t_dec=7, vocab=9, variable_len=True, seed=3, tasks=tasks)
    b["logmel"] = b["logmel"].astype(np.float64)
    nl = {"char": 3, "phone": 2}
    ref = O.seq2seq_forward(b, w, tasks=tasks, num_layers=nl, bi_dir=bi, is_training=True)
    W = R.weights_to_torch(w)
    total, losses, outs = R.seq2seq_loss(b, W, tasks=tasks, num_layers=nl, bi_dir=bi)
    for t in tasks:
        np.testing.assert_allclose(outs[t].detach().numpy(), ref["outputs"][t], rtol=0, atol=1e-12)
    np.testing.assert_allclose(total.item(), ref["total_loss"], rtol=1e-12)
    total.backward()
    assert all(W[k].grad is not None and torch.isfinite(W[k].grad).all() for k in W)


def test_all_equal_length_equals_unmasked():
    """dynamic_rnn with len == T must equal the plain recurrence (no masking effects)."""
    rng = np.random.default_rng(0)
    T, B, I, H = 9, 3, 5, 4
    x = rng.standard_normal((T, B, I)); w = rng.uniform(-0.5, 0.5, (I + H, 4 * H)); b = rng.uniform(-0.5, 0.5, 4 * H)
    out, _ = O.lstm_layer(x, [T] * B, w, b)
    c = np.zeros((B, H)); h = np.zeros((B, H))
    for t in range(T):
        c, h = O.lstm_cell(x[t], c, h, w, b)
        np.testing.assert_array_equal(out[t], h)


def test_backward_direction_is_reverse_sequence():
    """bw(x) == reverse_sequence(fw(reverse_sequence(x))) per utterance, zeros past the length."""
    rng = np.random.default_rng(1)
    T, B, I, H = 11, 4, 3, 5
    x = rng.standard_normal((T, B, I)); w = rng.uniform(-0.5, 0.5, (I + H, 4 * H)); b = rng.uniform(-0.5, 0.5, 4 * H)
    lens = np.array([11, 4, 1, 7])
    bw, _ = O.lstm_layer(x, lens, w, b, reverse=True)
    xr = np.zeros_like(x)
    for i, L in enumerate(lens):
        xr[:L, i] = x[:L, i][::-1]
    fw, _ = O.lstm_layer(xr, lens, w, b)
    for i, L in enumerate(lens):
        np.testing.assert_array_equal(bw[:L, i], fw[:L, i][::-1])
        assert not bw[L:, i].any()


def test_pyramid_odd_even_and_lengths():
    x = np.arange(2 * 5 * 3, dtype=np.float64).reshape(2, 5, 3)
    out, ln = O.pyramid(x, [5, 2])
    assert out.shape == (2, 3, 6) and list(ln) == [3, 1]
    np.testing.assert_array_equal(out[0, 2], np.concatenate((x[0, 4], np.zeros(3))))    # [h_last, 0]
    out, ln = O.pyramid(x[:, :4], [4, 3])
    assert out.shape == (2, 2, 6) and list(ln) == [2, 2]
    with pytest.raises(ValueError):
        O.pyramid(x, [4, 2])          # max_len even but T odd: tf.reshape would fail


def test_loss_matches_torch_composition():
    rng = np.random.default_rng(2)
    T, B, V = 6, 4, 11
    lg = rng.standard_normal((T * B, V)); tg = rng.integers(0, V, (T, B)); ln = np.array([6, 1, 3, 5])
    ref = O.cross_entropy_loss(lg, tg, ln)
    got = R.cross_entropy_loss(torch.tensor(lg), tg, ln).item()
    np.testing.assert_allclose(got, ref, rtol=1e-12)


def test_adam_and_clip_tf_semantics():
    g = [np.array([3.0, 4.0]), np.array([12.0])]            # global norm 13
    cl, gn = O.clip_by_global_norm(g, 5.0)
    assert abs(gn - 13.0) < 1e-12
    np.testing.assert_allclose(np.concatenate(cl), np.array([3.0, 4.0, 12.0]) * 5.0 / 13.0)
    cl, _ = O.clip_by_global_norm(g, 20.0)                  # below the threshold: untouched
    np.testing.assert_allclose(np.concatenate(cl), [3.0, 4.0, 12.0])
    var, m, v = O.adam_step(np.array([1.0]), np.zeros(1), np.zeros(1), np.array([0.5]), 1, 1e-3)
    # step 1: m = 0.05, v = 2.5e-4, lr_t = 1e-3*sqrt(1e-3)/0.1 -> update = lr_t*m/(sqrt(v)+eps) ~= 1e-3
    np.testing.assert_allclose(var, 1.0 - 1e-3 * np.sqrt(1e-3) / 0.1 * 0.05 / (np.sqrt(2.5e-4) + 1e-8))


def test_gru_layer_oracle_matches_the_torch_twin_and_dynamic_rnn_properties():
    """tf.nn.rnn_cell.GRUCell under dynamic_rnn (encoder.py:45-48; restated from the published TF-1.x cell -- parity unpinned like
    the rest of the TF-graph half): the NumPy oracle equals its autograd twin; outputs past each length are zero; the bw
    direction equals reverse(fw(reverse(x))) per utterance; equal lengths == no masking."""
    import torch
    from oracle import asr_oracle as O, torch_ref as R
    rng = np.random.default_rng(5)
    T, B, IN, H = 11, 4, 6, 5
    x = rng.standard_normal((T, B, IN))
    lens = [11, 7, 2, 1]
    wg = rng.uniform(-0.5, 0.5, (IN + H, 2 * H)); bg = np.ones(2 * H)
    wc = rng.uniform(-0.5, 0.5, (IN + H, H)); bc = rng.uniform(-0.1, 0.1, H)
    tt = lambda a: torch.tensor(a)
    for rev in (False, True):
        o, hlast = O.gru_layer(x, lens, wg, bg, wc, bc, rev)
        t = R.gru_layer(tt(x), lens, tt(wg), tt(bg), tt(wc), tt(bc), rev).numpy()
        np.testing.assert_allclose(o, t, atol=1e-14)
        for b, l in enumerate(lens):
            assert np.abs(o[l:, b]).max(initial=0.0) == 0.0
    o_bw, _ = O.gru_layer(x, lens, wg, bg, wc, bc, True)
    for b, l in enumerate(lens):
        o_f, _ = O.gru_layer(x[:l, b:b + 1][::-1], [l], wg, bg, wc, bc, False)
        np.testing.assert_allclose(o_f[::-1, 0], o_bw[:l, b], atol=1e-14)
    # one step by hand: h' = u*h + (1-u)*tanh([x, r*h].Wc + bc), r|u = sigmoid([x,h].Wg + bg) with h = 0
    h1 = O.gru_cell(x[0], np.zeros((B, H)), wg, bg, wc, bc)
    u = 1.0 / (1.0 + np.exp(-(x[0] @ wg[:IN] + bg)))[:, H:]
    np.testing.assert_allclose(h1, (1 - u) * np.tanh(x[0] @ wc[:IN] + bc), atol=1e-14)


def test_gru_encoder_oracle_equals_twin_through_the_pyramid():
    import torch
    from oracle import asr_oracle as O, torch_ref as R
    from e2e_asr_amd.weights import init_weights
    w = {k: v.astype(np.float64) for k, v in init_weights(feat=6, hidden=5, depth=3, seed=2, use_lstm=False, vocab={"char": 20},
                                                          emb=4, hidden_dec=5, lm_hidden=5, attn_vec=3).items()}
    assert any("gru_cell/gates/kernel" in k for k in w) and not any("encoder" in k and "basic_lstm_cell" in k for k in w)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 13, 6)); lens = [13, 9, 4]
    att, _, ln = O.encoder(x, lens, w, {"char": 3})
    W = R.weights_to_torch(w)
    att_t, _ = R.encoder(torch.tensor(x), lens, W, {"char": 3})
    np.testing.assert_allclose(att[3], att_t[3].detach().numpy(), atol=1e-13)
    assert att[3].shape == (3, 4, 10)


def test_gru_decoder_oracle_equals_twin_and_queries_the_state():
    """decoder.py:56-59, 79-80 with use_lstm False: both decoder cells are GRUCells and the attention query is the GRU state itself.
    The NumPy oracle (cell_stack with 4-entry GRU cells, state kept as (h, h)) equals its autograd twin, teacher-forced, ragged."""
    import torch
    from oracle import asr_oracle as O, torch_ref as R
    from e2e_asr_amd.weights import init_weights, synthetic_batch
    w = {k: v.astype(np.float64) for k, v in init_weights(feat=6, hidden=5, depth=2, seed=4, use_lstm=False, dec_use_lstm=False,
                                                          vocab={"char": 17}, emb=4, hidden_dec=6, lm_hidden=5, attn_vec=3).items()}
    assert any("rnn/gru_cell_1/candidate/kernel" in k for k in w) and not any("basic_lstm_cell" in k for k in w)
    assert any("SimpleProjection" in k for k in w)                  # lm_hidden 5 != hidden_dec 6
    b = synthetic_batch(B=4, T=14, F=6, t_dec=7, vocab=17, variable_len=True, seed=3)
    b64 = dict(b); b64["logmel"] = b["logmel"].astype(np.float64)
    r = O.seq2seq_forward(b64, w, num_layers={"char": 2}, is_training=True)
    W = R.weights_to_torch(w)
    total, losses, outs = R.seq2seq_loss(b64, W, num_layers={"char": 2})
    np.testing.assert_allclose(r["outputs"]["char"], outs["char"].detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(r["losses"]["char"], float(losses["char"]), rtol=1e-12)
    total.backward()
    assert all(W[k].grad is not None and np.isfinite(W[k].grad.numpy()).all() for k in W if "rnn_decoder_char" in k)
