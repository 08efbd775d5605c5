"""GPU: edge cases -- batch 1, single frame, length-1 rows, single encoder position,
argument validation.  The reference's own NumPy attention raises at T=1 (beam_search.py:154-157);
the TF graph handles it, and so does this path."""
import numpy as np
import pytest
import torch

from oracle import asr_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def T(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x))
    return (t if dtype is None else t.to(dtype)).to(DEV)


@pytest.mark.parametrize("B,Tn,lens", [(1, 1, [1]), (1, 9, [9]), (3, 1, [1, 1, 1]), (2, 5, [1, 5]), (33, 3, None)])
def test_lstm_tiny_shapes(B, Tn, lens):
    from e2e_asr_amd import ops
    rng = np.random.default_rng(B * 10 + Tn)
    IN, H = 8, 64
    lens = np.asarray(lens if lens is not None else rng.integers(1, Tn + 1, B))
    x = rng.standard_normal((B, Tn, IN)).astype(np.float32)
    k = rng.uniform(-0.3, 0.3, (IN + H, 4 * H)).astype(np.float32)
    bz = rng.uniform(-0.3, 0.3, 4 * H).astype(np.float32)
    out, gates, act, hp = ops.lstm_layer_fwd(T(x), T(lens, torch.int32), T(k), T(bz), T(k), T(bz), save=True)
    ops.check_device_flag(DEV)
    ref = O.bilstm_layer(np.transpose(x, (1, 0, 2)).astype(np.float64), lens, k.astype(np.float64), bz.astype(np.float64),
                         k.astype(np.float64), bz.astype(np.float64))
    np.testing.assert_allclose(out.cpu().numpy(), np.transpose(ref, (1, 0, 2)), rtol=0, atol=2e-5)
    dk = [torch.zeros_like(T(k)), torch.zeros_like(T(bz)), torch.zeros_like(T(k)), torch.zeros_like(T(bz))]
    dx = ops.lstm_layer_bwd(T(x), T(lens, torch.int32), T(k), T(k), torch.ones_like(out), gates, act, hp, *dk)
    ops.check_device_flag(DEV)
    assert torch.isfinite(dx).all() and all(torch.isfinite(d).all() for d in dk)
    for b in range(B):                       # no gradient flows into padding frames
        assert not dx[b, lens[b]:].any()


def test_attention_single_position_and_zero_length():
    from e2e_asr_amd import ops
    rng = np.random.default_rng(0)
    B, Te, H, A, D = 3, 4, 32, 16, 48
    enc = rng.standard_normal((B, Te, D)).astype(np.float32)
    q = rng.standard_normal((B, H)).astype(np.float32)
    w = rng.uniform(-0.3, 0.3, (H, A)).astype(np.float32); bb = np.zeros(A, np.float32)
    v = rng.uniform(-0.3, 0.3, A).astype(np.float32)
    hf = rng.standard_normal((B, Te, A)).astype(np.float32)
    lens = np.array([1, 4, 0])
    ctx, alpha = ops.attention(T(q), T(w), T(bb), T(v), T(hf), T(enc), T(lens, torch.int32))
    a = alpha.cpu().numpy(); c = ctx.cpu().numpy()
    np.testing.assert_allclose(a[0], [1, 0, 0, 0], atol=0)          # one live position: all the mass
    np.testing.assert_allclose(c[0], enc[0, 0], rtol=0, atol=1e-6)
    np.testing.assert_allclose(a[1].sum(), 1.0, rtol=1e-6)
    assert not a[2].any() and not c[2].any()                          # zero-length: zeros, never NaN


def test_model_batch1_single_target_token():
    """Whole train step at B=1 with a 1-token target and T=2 frames (every loop runs once or twice)."""
    from e2e_asr_amd.attn_decoder import AttnDecoder
    from e2e_asr_amd.seq2seq_model import Seq2SeqModel
    from oracle import torch_ref as R
    p = Seq2SeqModel.class_params()
    p.num_layers = {"char": 2}; p.max_output = {"char": 4}
    p.encoder_params.use_lstm = True; p.encoder_params.hidden_size = 64; p.encoder_params.out_prob = 1.0
    dp = AttnDecoder.class_params()
    dp.hidden_size_dec = 32; dp.lm_hidden_size = 32; dp.emb_size = 16; dp.attention_vec_size = 8; dp.vocab_size = 9
    dp.out_prob_dec = 1.0; dp.samp_prob = 0.0
    p.decoder_params = {"char": dp}
    m = Seq2SeqModel(None, True, p, device="cuda:0", feat_length=12, seed=5)
    batch = {"logmel": np.random.default_rng(1).standard_normal((1, 2, 12)).astype(np.float32), "logmel_len": np.array([2]),
             "char": np.array([[1, 2]]), "char_len": np.array([1])}
    w = {k: v.astype(np.float64) for k, v in m.variables.to_arrays().items()}
    m.forward(batch); loss = m.total_loss.item(); m.backward()
    W = R.weights_to_torch(w)
    b64 = dict(batch); b64["logmel"] = batch["logmel"].astype(np.float64)
    total, _, _ = R.seq2seq_loss(b64, W, num_layers={"char": 2})
    np.testing.assert_allclose(loss, total.item(), rtol=1e-5)
    total.backward()
    for name in m.variables.names():
        ref = W[name].grad.numpy()
        np.testing.assert_allclose(m.variables.grad_of(name).cpu().numpy(), ref, rtol=0, atol=2e-3 * max(1e-3, np.abs(ref).max()))


def test_argument_validation_raises_valueerror():
    from e2e_asr_amd import ops
    x = torch.zeros(2, 4, 8, device=DEV); ln = torch.tensor([4, 4], dtype=torch.int32, device=DEV)
    k96 = torch.zeros(8 + 96, 4 * 96, device=DEV); b96 = torch.zeros(4 * 96, device=DEV)
    assert tuple(ops.lstm_layer_fwd(x, ln, k96, b96).shape) == (2, 4, 96)   # H=96 runs zero-padded to the 128-wide kernel
    k600 = torch.zeros(8 + 600, 4 * 600, device=DEV)
    with pytest.raises(ValueError, match="up to 512"):              # beyond the widest instantiated recurrent kernel
        ops.lstm_layer_fwd(x, ln, k600, torch.zeros(4 * 600, device=DEV))
    from e2e_asr_amd import _lib
    import ctypes as C
    hx = torch.zeros(1 << 20, dtype=torch.uint8, device=DEV); flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    out = torch.zeros(2, 4, 96, device=DEV); gates = torch.zeros(2, 4, 1, 4 * 96, device=DEV)
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = _lib.lib().asr_lstm_layer_fwd(None, p(x), 2, 4, 8, 8, p(ln), 96, 1, p(k96), p(b96), None, None, p(out), 4, p(gates),
                                       None, None, p(hx), hx.numel(), p(flag), 1.0, 0, None, None)
    assert rc == -3                                                  # the C ABI itself: ASR_EUNSUPPORTED for H = 96
    k = torch.zeros(9 + 64, 256, device=DEV)
    with pytest.raises(ValueError, match="kernel rows"):
        ops.lstm_layer_fwd(x, ln, k, torch.zeros(256, device=DEV))
    with pytest.raises(ValueError, match="inner dimensions"):
        ops.gemm(torch.zeros(3, 4, device=DEV), torch.zeros(5, 6, device=DEV))
    with pytest.raises(ValueError):                                  # K1 not a multiple of 4
        ops.linear(torch.zeros(2, 6, device=DEV), torch.zeros(6, 8, device=DEV))
    with pytest.raises(ValueError, match="contiguous float32"):
        ops.gemm(torch.zeros(4, 4, device=DEV).t(), torch.zeros(4, 4, device=DEV))


def test_gradients_finite_when_recycled_memory_holds_nan():
    """Workspaces come from torch.empty: rows past an utterance's length that a kernel does not write must never
    reach a product (found by the bf16 test: hprev rows past the length held NaN bit patterns from a freed tensor and
    poisoned dK_h = Hprev^T.dG although dG is zero there).  Poison the allocator's free blocks, then take a ragged step."""
    from tests.test_gpu_model import _model, _batch
    rng = np.random.default_rng(77)
    for _ in range(2):
        junk = [torch.full((n,), float("nan"), device=DEV) for n in (1 << 22, 1 << 20, 1 << 18, 1 << 16, 3 << 14)]
        del junk
        m = _model(enc_update=dict(hidden_size=64), num_layers={"char": 3}, seed=9,
                   dec_update=dict(hidden_size_dec=64, lm_hidden_size=64, emb_size=24, attention_vec_size=16))
        b = _batch(rng, 6, 29, 20, 11, 50, lens=[29, 3, 17, 1, 22, 8])
        m.forward(b)
        assert np.isfinite(m.total_loss.item())
        m.backward()
        for n in m.variables.names():
            assert torch.isfinite(m.variables.grad_of(n)).all(), n


@pytest.mark.parametrize("tdec", [2, 3])
def test_persistent_decoder_paths_with_one_and_two_output_steps(monkeypatch, tdec):
    """T_out = 1 and 2: the persistent chains run a single step (no exchange of a previous step, nothing published for a
    next one) -- must equal the per-step launch path, forward and backward."""
    from tests.test_gpu_model import _model, _batch
    rng = np.random.default_rng(91)
    kw = dict(enc_update=dict(hidden_size=64), num_layers={"char": 2}, seed=4,
              dec_update=dict(hidden_size_dec=64, lm_hidden_size=64, emb_size=24, attention_vec_size=16))
    b = _batch(rng, 5, 19, 20, tdec, 50)
    res = []
    for chain in ("1", "0"):
        monkeypatch.setenv("ASR_DEC_CHAIN", chain)
        monkeypatch.setenv("ASR_LM_CHAIN", chain)
        m = _model(**kw)
        m.forward(b)
        assert (m.decoder["char"].saved["ws"].get("chain_ws") is not None) == (chain == "1")
        out, loss = m.outputs["char"].cpu().numpy().copy(), m.total_loss.item()
        m.backward()
        from e2e_asr_amd import ops
        ops.check_device_flag(torch.device(DEV))
        res.append((out, loss, {n: m.variables.grad_of(n).cpu().numpy().copy() for n in m.variables.names()}))
    assert res[0][0].shape[0] == (tdec - 1) * 5
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=0, atol=2e-5)
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-6)
    for n, g0 in res[1][2].items():
        err = np.abs(res[0][2][n] - g0).max() / max(1e-3, np.abs(g0).max())
        assert err < 1e-4, (n, err)


def test_device_prefetcher_stages_batches_in_order():
    """e2e_asr_amd/prefetch.py: batches come out in order, `logmel` already on the device (copied from pinned host memory on a
    copy stream one batch ahead), every other key untouched; device-resident batches pass through."""
    from e2e_asr_amd.prefetch import DevicePrefetcher
    rng = np.random.default_rng(0)
    src = [{"logmel": rng.standard_normal((3, 5 + i, 4)).astype(np.float32), "logmel_len": np.array([5 + i] * 3), "utt_id": i}
           for i in range(13)]                         # more batches than the ring of pinned buffers holds
    src[3]["logmel"] = torch.from_numpy(src[3]["logmel"]).to(DEV)
    got = list(DevicePrefetcher(src, DEV))
    assert [b["utt_id"] for b in got] == list(range(13))
    for b, s in zip(got, src):
        assert b["logmel"].is_cuda and b["logmel_len"] is s["logmel_len"]
        ref = s["logmel"].cpu().numpy() if torch.is_tensor(s["logmel"]) else s["logmel"]
        np.testing.assert_array_equal((b["logmel"] * 1.0).cpu().numpy(), ref)
    assert list(DevicePrefetcher([], DEV)) == []
    primed = DevicePrefetcher(src[:4], DEV).primed()                # reader thread already running before the first next()
    assert [b["utt_id"] for b in primed] == [0, 1, 2, 3]

    def broken():
        yield src[0]
        raise RuntimeError("reader failed")
    with pytest.raises(RuntimeError, match="reader failed"):      # a failing reader surfaces in the consumer
        list(DevicePrefetcher(broken(), DEV))
    it = iter(DevicePrefetcher(iter(src), DEV))                   # an abandoned iterator stops its worker
    next(it)
    it.close()
    # reader_finished(): false before a reader exists and while it still has batches to read, true once it has read the last
    # one -- the moment the single-bucket training loop starts the next epoch's reader over the same dataset (train.py)
    import time
    pf = DevicePrefetcher(src, DEV, depth=1)
    assert not pf.reader_finished()
    it = pf.primed()
    first = next(it)
    assert first["utt_id"] == 0 and not pf.reader_finished()      # 6+ batches, queue depth 2 + ring: the reader is still blocked
    rest = list(it)
    assert [b["utt_id"] for b in rest] == list(range(1, len(src)))
    for _ in range(100):
        if pf.reader_finished():
            break
        time.sleep(0.01)
    assert pf.reader_finished()


def test_concat_kx_layers_equals_torch_cat():
    """ops.concat_kx_layers (asr_concat2_multi: every layer's [in,8H] kernel and [8H] bias concatenation in one launch) is
    bit-equal to the two torch.cat calls per layer it replaces."""
    from e2e_asr_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    layers = []
    for IN, H in ((80, 256), (1024, 256), (96, 64), (512, 128)):
        mk = lambda *shape: torch.randn(*shape, generator=g).to(DEV)
        layers.append((mk(IN + H, 4 * H), mk(4 * H), mk(IN + H, 4 * H), mk(4 * H)))
    got = ops.concat_kx_layers(layers)
    for (kf, bf, kb, bb), (kx, bc) in zip(layers, got):
        IN = kf.shape[0] - kf.shape[1] // 4
        assert torch.equal(kx, torch.cat([kf[:IN], kb[:IN]], 1)) and torch.equal(bc, torch.cat([bf, bb]))
