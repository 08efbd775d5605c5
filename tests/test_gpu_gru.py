"""GPU: the GRUCell encoder (encoder.py:42-53 with use_lstm False -- the `Encoder.class_params()` default, encoder.py:27; the
reference CLI always sets use_lstm, encoder.py:187, so this is completeness, not the measured path).  csrc/gru.hip against the
float64 oracle (oracle/asr_oracle.py gru_layer: tf.nn.rnn_cell.GRUCell under dynamic_rnn as published -- parity unpinned like the
rest of the TF-graph half) and against float64 autograd of its torch twin."""
import numpy as np
import pytest
import torch

from oracle import asr_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cells(rng, IN, H, ndir):
    return [(rng.uniform(-0.3, 0.3, (IN + H, 2 * H)).astype(np.float32), rng.uniform(0.5, 1.5, 2 * H).astype(np.float32),
             rng.uniform(-0.3, 0.3, (IN + H, H)).astype(np.float32), rng.uniform(-0.2, 0.2, H).astype(np.float32))
            for _ in range(ndir)]


def _dev(cells):
    return [tuple(torch.from_numpy(a).to(DEV) for a in c) for c in cells]


def _masks(seed, B, T, t_out, H, ndir, keep):
    from tests.test_gpu_model import _np_keep_scale
    out = []
    for d in range(ndir):
        tt, bb, jj = np.meshgrid(np.arange(T), np.arange(B), np.arange(H), indexing="ij")
        out.append(_np_keep_scale(seed, bb * t_out + tt, d * H + jj, keep))
    return out


@pytest.mark.parametrize("B,T,IN,H,ndir,lens,tout,keep", [
    (5, 23, 12, 16, 2, [23, 17, 9, 2, 1], 24, 1.0),
    (3, 40, 80, 256, 2, [40, 31, 5], 40, 1.0),
    (4, 19, 10, 300, 1, [19, 19, 8, 3], 19, 1.0),          # H > 256: two units per thread
    (4, 21, 14, 64, 2, [21, 20, 11, 1], 22, 0.8),          # output-only dropout
])
def test_gru_layer_forward_vs_oracle(B, T, IN, H, ndir, lens, tout, keep):
    from e2e_asr_amd import ops
    rng = np.random.default_rng(B * 100 + T + H)
    x = rng.standard_normal((B, T, IN)).astype(np.float32)
    cells = _cells(rng, IN, H, ndir)
    seed = 77
    out = ops.gru_layer_fwd(torch.from_numpy(x).to(DEV), torch.tensor(lens, dtype=torch.int32, device=DEV), _dev(cells),
                            t_out=tout, keep_prob=keep, seed=seed).cpu().numpy()
    km = _masks(seed, B, T, tout, H, ndir, keep) if keep < 1.0 else [None] * ndir
    x_tm = np.transpose(x, (1, 0, 2)).astype(np.float64)
    halves = [O.gru_layer(x_tm, lens, *[a.astype(np.float64) for a in cells[d]], reverse=d == 1, keep_mask=km[d])[0] for d in range(ndir)]
    ref = np.transpose(np.concatenate(halves, 2), (1, 0, 2))
    assert out.shape == (B, tout, ndir * H)
    np.testing.assert_allclose(out[:, :T], ref, rtol=0, atol=2e-5)
    assert not out[:, T:].any()
    for b, l in enumerate(lens):
        assert not out[b, l:].any()


@pytest.mark.parametrize("B,T,IN,H,ndir,lens,keep", [
    (4, 17, 9, 16, 2, [17, 12, 5, 1], 1.0),
    (3, 25, 20, 64, 2, [25, 14, 3], 0.8),
    (2, 12, 7, 300, 1, [12, 6], 1.0),
])
def test_gru_layer_backward_vs_autograd(B, T, IN, H, ndir, lens, keep):
    """tf.gradients through the layer (seq2seq_model.py:148): dx and every weight / bias gradient against float64 autograd."""
    from e2e_asr_amd import ops
    from oracle import torch_ref as R
    rng = np.random.default_rng(B + T + IN + H)
    x = rng.standard_normal((B, T, IN)).astype(np.float32)
    cells = _cells(rng, IN, H, ndir)
    dout = rng.standard_normal((B, T, ndir * H)).astype(np.float32)
    seed = 31
    ln = torch.tensor(lens, dtype=torch.int32, device=DEV)
    cd = _dev(cells)
    xd = torch.from_numpy(x).to(DEV)
    out, gx, cx, hprev, rh = ops.gru_layer_fwd(xd, ln, cd, save=True, keep_prob=keep, seed=seed)
    grads = [tuple(torch.randn_like(t) for t in c) for c in cd]              # accumulated into: start from non-zero
    g0 = [tuple(t.clone() for t in c) for c in grads]
    dx = ops.gru_layer_bwd(xd, ln, cd, torch.from_numpy(dout).to(DEV), gx, cx, hprev, rh, grads, need_dx=True, keep_prob=keep, seed=seed)
    torch.cuda.synchronize()
    km = _masks(seed, B, T, T, H, ndir, keep) if keep < 1.0 else [None] * ndir
    xt = torch.tensor(x.astype(np.float64), requires_grad=True)
    wt = [tuple(torch.tensor(a.astype(np.float64), requires_grad=True) for a in c) for c in cells]
    halves = [R.gru_layer(xt.transpose(0, 1), lens, *wt[d], reverse=d == 1,
                          keep_mask=None if km[d] is None else torch.tensor(km[d])) for d in range(ndir)]
    o = torch.cat(halves, 2).transpose(0, 1)
    np.testing.assert_allclose(out.cpu().numpy(), o.detach().numpy(), rtol=0, atol=2e-5)
    (o * torch.tensor(dout.astype(np.float64))).sum().backward()
    rel = lambda got, ref: float(np.abs(got - ref).max()) / max(1e-3, float(np.abs(ref).max()))
    assert rel(dx.cpu().numpy(), xt.grad.numpy()) < 1e-4
    for d in range(ndir):
        for k in range(4):
            got = (grads[d][k] - g0[d][k]).cpu().numpy()
            assert rel(got, wt[d][k].grad.numpy()) < 2e-4, (d, k)


def test_default_encoder_params_train_end_to_end_vs_oracle_and_autograd():
    """`Encoder.class_params()` as the reference ships it (use_lstm False, encoder.py:27): a Seq2SeqModel with that encoder -- three
    pyramidal BiGRU layers, dropout -- gives the oracle's logits and loss, every gradient equals float64 autograd, and train
    steps run and lower the loss."""
    from tests.test_gpu_parity3 import _model, _f64, _grad_check
    from tests.test_gpu_model import _np_keep_scale
    from e2e_asr_amd import ops
    from e2e_asr_amd.weights import synthetic_batch
    m = _model(enc_update=dict(use_lstm=False, hidden_size=48, out_prob=1.0), num_layers={"char": 3}, seed=9,
               dec_update=dict(hidden_size_dec=32, lm_hidden_size=32, emb_size=24, attention_vec_size=16))
    assert m.encoder.get_cell() == "GRUCell(48)"
    assert any("gru_cell/candidate/kernel" in n for n in m.variables.names())
    b = synthetic_batch(B=6, T=41, F=20, t_dec=9, vocab=50, variable_len=True, seed=5)
    m.forward(b)
    ops.check_device_flag(torch.device(DEV))
    w = _f64(m.variables.to_arrays())
    b64 = dict(b); b64["logmel"] = b["logmel"].astype(np.float64)
    r = O.seq2seq_forward(b64, w, num_layers={"char": 3}, is_training=True)
    np.testing.assert_allclose(m.outputs["char"].cpu().numpy(), r["outputs"]["char"], rtol=0, atol=1e-4)
    m.backward()
    ops.check_device_flag(torch.device(DEV))
    total, worst, name = _grad_check(m, b, num_layers={"char": 3})
    np.testing.assert_allclose(m.total_loss.item(), total, rtol=2e-5)
    losses = []
    for _ in range(12):
        losses.append(float(m.step(b)["char"]))
    ops.check_device_flag(torch.device(DEV))
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


# ------------------------------------------------------------------ the GRUCell attention decoder (decoder.py:56-59, 79-80)
@pytest.mark.parametrize("keep,lmH", [(1.0, 32), (0.8, 32), (0.8, 24)])
def test_gru_decoder_logits_loss_and_gradients_vs_oracle_and_autograd(keep, lmH):
    """use_lstm False in the decoder's params: both decoder cells are GRUCells, the attention query is the GRU state itself
    (e2e_asr_amd/gru_decoder.py: a host-composed per-step path over csrc/gru.hip with T = 1, asr_attn_bwd, the step kernels).
    Logits and loss vs the float64 oracle, every gradient vs float64 autograd; ragged lengths; the LM cell's output dropout masks
    are reproduced from the counter-based generator; lmH != 32: + SimpleProjection.  With the encoder's default (GRU) cell too."""
    from tests.test_gpu_parity3 import _model, _f64
    from tests.test_gpu_model import _np_keep_scale
    from e2e_asr_amd import ops
    from e2e_asr_amd.gru_decoder import step_seed
    from e2e_asr_amd.weights import synthetic_batch
    from oracle import torch_ref as R
    nl = {"char": 2}
    m = _model(enc_update=dict(use_lstm=False, hidden_size=40), num_layers=nl, seed=21,
               dec_update=dict(use_lstm=False, hidden_size_dec=32, lm_hidden_size=lmH, emb_size=24, attention_vec_size=16,
                               out_prob_dec=keep))
    assert m.decoder["char"].cell == "GRUCell(32)"
    names = m.variables.names()
    assert any("rnn/gru_cell_1/gates/kernel" in n for n in names) and not any("basic_lstm_cell" in n for n in names)
    B = 5
    b = synthetic_batch(B=B, T=22, F=20, t_dec=9, vocab=50, variable_len=True, seed=61)
    m.global_step = 3
    m.forward(b)
    ops.check_device_flag(torch.device(DEV))
    T_out = m.decoder["char"].saved["t_out"]
    seed = m.decoder["char"].saved["seed"]
    lm_masks = None
    if keep < 1.0:          # (the oracle's raw_rnn restatement also runs the LM cell once past the last step: T_out + 1 rows)
        bb, jj = np.meshgrid(np.arange(B), np.arange(lmH), indexing="ij")
        lm_masks = np.stack([_np_keep_scale(step_seed(seed, "lm", i), bb, jj, keep) for i in range(T_out + 1)])
    w = _f64(m.variables.to_arrays())
    b64 = dict(b); b64["logmel"] = b["logmel"].astype(np.float64)
    att, _, lens = O.encoder(b64["logmel"], b64["logmel_len"], w, nl)
    ref = O.attn_decoder(np.transpose(b["char"]), b["char_len"], att[2], lens[2], w, is_training=True, lm_keep_masks=lm_masks)
    np.testing.assert_allclose(m.outputs["char"].cpu().numpy(), ref, rtol=0, atol=1e-4)
    m.backward()
    ops.check_device_flag(torch.device(DEV))
    W = R.weights_to_torch(w)
    total, _, _ = R.seq2seq_loss(b64, W, num_layers=nl, lm_keep_masks=None if lm_masks is None else {"char": torch.tensor(lm_masks)})
    np.testing.assert_allclose(m.total_loss.item(), total.item(), rtol=2e-5)
    total.backward()
    for name in names:
        ref_g = W[name].grad.numpy()
        err = np.abs(m.variables.grad_of(name).cpu().numpy() - ref_g).max() / max(1e-3, np.abs(ref_g).max())
        assert err < 2e-3, (name, err)


def test_gru_decoder_inference_and_sampling_modes():
    """The GRU decoder in the inference graph (argmax feedback every step: ids equal to the float64 oracle's) and under scheduled
    sampling (runs, feeds drawn tokens, stays finite, the loss falls over a few steps)."""
    from tests.test_gpu_parity3 import _model, _f64
    from e2e_asr_amd import ops
    from e2e_asr_amd.weights import synthetic_batch
    dec = dict(use_lstm=False, hidden_size_dec=32, lm_hidden_size=32, emb_size=24, attention_vec_size=16)
    m = _model(enc_update=dict(hidden_size=64), num_layers={"char": 2}, seed=47, training=False, dec_update=dec, max_output={"char": 9})
    b = synthetic_batch(B=4, T=18, F=20, t_dec=8, vocab=50, variable_len=True, seed=71)
    out = m.forward(b)["char"].cpu().numpy()
    w = _f64(m.variables.to_arrays())
    b64 = dict(b); b64["logmel"] = b["logmel"].astype(np.float64)
    r = O.seq2seq_forward(b64, w, num_layers={"char": 2}, is_training=False, max_output={"char": 9})["outputs"]["char"]
    np.testing.assert_allclose(out, r, rtol=0, atol=1e-4)
    np.testing.assert_array_equal(m.greedy_ids().cpu().numpy(), O.greedy_decode_ids(r, 4))
    m2 = _model(enc_update=dict(hidden_size=64), num_layers={"char": 2}, seed=47, dec_update=dict(dec, samp_prob=0.5))
    m2.decoder["char"].coin_seed = 3
    losses = [float(m2.step(b)["char"]) for _ in range(8)]
    ops.check_device_flag(torch.device(DEV))
    assert torch.isfinite(m2.variables.flat).all() and m2.global_step == 8 and losses[-1] < losses[0]
