"""BASELINE config 3 ("same model bf16"): GEMMs with bf16 MFMA operands (csrc/gemm.hip gemm_planes_kernel, one plane),
fp32 accumulation and fp32 everything else -- and "bf16x2" (two planes, three products).  Tolerances are stated here.
Kernel: equal to an fp64 product of the bf16-ROUNDED operands (resp. of their two-plane roundings, minus the dropped
lo.lo term) to fp32 accumulation error.  Model, against the float64 oracle on the same fp32 weights and inputs:
  * bf16 (one plane): the north star's 1e-3 logit tolerance is an fp32 statement and one bf16 plane (2^-9 relative per operand,
    through four encoder layers and 120 decoder steps) does NOT meet it at full size: measured 1.3e-3; the bound asserted
    here is 5e-3 absolute on the logits and 1 % on the loss;
  * bf16x2: meets the 1e-3 (asserted < 1e-3 at the full config-2 size; measured ~1e-5)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _restore_process_wide_precision():
    """Every test here changes process-wide state (GEMM precision, decoder plane count, bf16-pipe recurrences); a failing
    assertion must not leave it set for the rest of the session."""
    yield
    try:
        from e2e_asr_amd import ops
        ops.set_gemm_precision("f32")
        ops.set_decoder_bf16_planes(2)
        ops.set_lstm_mfma(False)
    except Exception:
        pass


@pytest.fixture(autouse=True)
def _restore_precision():
    from e2e_asr_amd import ops
    yield
    ops.set_gemm_precision("f32")


def _bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float64)       # round-to-nearest-even, like v_cvt_pk_bf16_f32


@pytest.mark.parametrize("ta,tb,M,N,K,acc", [(0, 0, 256, 128, 96, 0), (0, 1, 128, 256, 64, 0), (1, 0, 128, 128, 4096, 0),
                                             (1, 0, 256, 256, 8192, 1), (0, 0, 384, 1024, 1024, 1)])
def test_gemm_bf16_equals_product_of_rounded_operands(ta, tb, M, N, K, acc):
    from e2e_asr_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = torch.randn((K, M) if ta else (M, K), generator=g).to(DEV)
    b = torch.randn((N, K) if tb else (K, N), generator=g).to(DEV)
    bias = None if ta else torch.randn(N, generator=g).to(DEV)
    c0 = torch.randn(M, N, generator=g).to(DEV)
    ops.set_gemm_precision("bf16")
    assert ops.get_gemm_precision() == "bf16"
    out = ops.gemm(a, b, bias, bool(ta), bool(tb), out=c0.clone(), accumulate=bool(acc))
    ra, rb = _bf16_round(a), _bf16_round(b)
    ref = (ra.t() if ta else ra) @ (rb.t() if tb else rb)
    if bias is not None:
        ref = ref + bias.double()
    if acc:
        ref = ref + c0.double()
    err = (out.double() - ref).abs().max().item()
    assert err <= 2e-6 * K ** 0.5 * 4 + 1e-5, err            # fp32 accumulation of K products of magnitude ~1
    # ... and it really is the bf16 path: the exact fp32 kernel gives a visibly different (more accurate) result
    ops.set_gemm_precision("f32")
    exact = ops.gemm(a, b, bias, bool(ta), bool(tb), out=c0.clone(), accumulate=bool(acc))
    assert (exact - out).abs().max().item() > 1e-3


def _two_planes(x):
    h1 = x.to(torch.bfloat16).to(torch.float32)
    h2 = (x - h1).to(torch.bfloat16).to(torch.float32)
    return h1.double(), h2.double()


@pytest.mark.parametrize("ta,tb,M,N,K", [(0, 0, 256, 128, 96), (0, 1, 128, 256, 80), (1, 0, 128, 128, 4096), (0, 0, 384, 1024, 1024)])
def test_gemm_bf16x2_equals_three_products_of_the_two_plane_split(ta, tb, M, N, K):
    """bf16x2: x ~ h1 + h2 (two bf16 terms, 16 significand bits); the product keeps a1b1 + a1b2 + a2b1."""
    from e2e_asr_amd import ops
    g = torch.Generator(device="cpu").manual_seed(7 * M + N + K)
    a = torch.randn((K, M) if ta else (M, K), generator=g).to(DEV)
    b = torch.randn((N, K) if tb else (K, N), generator=g).to(DEV)
    ops.set_gemm_precision("bf16x2")
    assert ops.get_gemm_precision() == "bf16x2"
    out = ops.gemm(a, b, None, bool(ta), bool(tb))
    (a1, a2), (b1, b2) = _two_planes(a), _two_planes(b)
    op = lambda x, t: x.t() if t else x
    ref = op(a1, ta) @ op(b1, tb) + op(a1, ta) @ op(b2, tb) + op(a2, ta) @ op(b1, tb)
    assert (out.double() - ref).abs().max().item() <= 2e-6 * K ** 0.5 * 4 + 1e-5
    exact = op(a.double(), ta) @ op(b.double(), tb)
    rel = ((out.double() - exact).abs() / (op(a.double().abs(), ta) @ op(b.double().abs(), tb))).max().item()
    assert 2.0 ** -24 < rel < 2.0 ** -14, rel                  # ~2^-16: between fp32 and one bf16 plane


def test_partial_tiles_fall_back_to_exact_fp32():
    from e2e_asr_amd import ops
    a = torch.randn(100, 80, device=DEV); b = torch.randn(80, 1000, device=DEV)
    ops.set_gemm_precision("bf16")
    out = ops.gemm(a, b)
    ref = a.double() @ b.double()
    assert (out.double() - ref).abs().max().item() < 1e-4


def test_model_bf16_logits_and_gradients_close_to_fp64_oracle():
    """Config-2 architecture at reduced time extent (H=256 so that the products are whole tiles and take the bf16
    path): logits within 5e-2 absolute (fp32 path: 2e-7) and loss within 1e-2 relative of the float64 oracle; every
    gradient has cosine similarity >= 0.995 with the fp32 path's gradient."""
    from tests.test_gpu_model import _model, _batch, _f64
    from oracle import asr_oracle as O
    from e2e_asr_amd import ops
    rng = np.random.default_rng(41)
    kw = dict(enc_update=dict(hidden_size=256), num_layers={"char": 3}, seed=5,
              dec_update=dict(hidden_size_dec=256, lm_hidden_size=256, emb_size=256, attention_vec_size=128))
    b = _batch(rng, 8, 64, 20, 12, 50)
    res = {}
    for prec in ("f32", "bf16"):
        ops.set_gemm_precision(prec)
        m = _model(**kw)
        m.forward(b)
        out = m.outputs["char"].cpu().numpy().copy()
        loss = m.total_loss.item()
        m.backward()
        ops.check_device_flag(torch.device(DEV))
        res[prec] = (out, loss, {n: m.variables.grad_of(n).cpu().numpy().copy() for n in m.variables.names()}, m)
    m = res["f32"][3]
    w = _f64(m.variables.to_arrays())
    b64 = dict(b); b64["logmel"] = b["logmel"].astype(np.float64)
    r = O.seq2seq_forward(b64, w, num_layers={"char": 3}, is_training=True)
    ref = r["outputs"]["char"]
    assert np.abs(res["f32"][0] - ref).max() < 1e-4
    d = np.abs(res["bf16"][0] - ref).max()
    assert 1e-5 < d < 5e-3, d                                  # moved by the operand rounding, but bounded
    assert abs(res["bf16"][1] - r["total_loss"]) < 1e-2 * abs(r["total_loss"])
    for n, g32 in res["f32"][2].items():
        g16 = res["bf16"][2][n]
        cos = float((g32 * g16).sum() / (np.linalg.norm(g32) * np.linalg.norm(g16) + 1e-30))
        assert cos > 0.995, (n, cos)


@pytest.mark.parametrize("prec,tol,seed,wscale", [("bf16", 1e-3, 17, 1.0), ("bf16", 1e-3, 23, 1.0), ("bf16", 1e-3, 31, 1.0),
                                                   ("bf16", 5e-3, 17, 1.5),
                                                   ("bf16-one-plane-everywhere", 5e-3, 17, 1.0), ("bf16x2", 1e-3, 17, 1.0)])
def test_config3_full_size_logits_vs_oracle(prec, tol, seed, wscale):
    """BASELINE config 3's per-GPU workload at FULL size (B = 32, T = 800): logits against the float64 oracle.
    "bf16" = the mode `bench.py --config 3` runs: one bf16 operand plane in the encoder's products (91 % of the GEMM FLOPs), two
    in the decoder's (ops._decoder_precision), fp32 recurrences: WITHIN THE NORTH STAR'S 1e-3 (measured 0.82e-3), loss within
    1 %.  One plane everywhere: 1.22e-3 (asserted < 5e-3; each half of the model alone carries ~0.85e-3).  bf16x2 everywhere
    (two planes, three products): 2.4e-6."""
    from tests.test_gpu_model import _model, _f64
    from e2e_asr_amd import ops
    from e2e_asr_amd.weights import synthetic_batch
    from oracle import asr_oracle as O
    ops.set_decoder_bf16_planes(1 if prec == "bf16-one-plane-everywhere" else 2)
    prec = prec.split("-")[0]
    ops.set_gemm_precision(prec)
    m = _model(feat=80, vocab={"char": 1000}, params_update=dict(max_output={"char": 120}), seed=seed)
    if wscale != 1.0:
        # "trained-like" weights: every LSTM kernel 1.5 x the init scale (larger pre-activations, logits 3 x larger) -- how the
        # bf16 margin behaves away from a random init (round-3 review).  Measured (scripts/exp_bf16_scale.py, round 4): the
        # error of bf16 mode AND of the fp32 path grow together with the scale -- x1 0.68e-3 / 2.8e-7, x1.5 3.2e-3 / 8.1e-7,
        # x2 1.1e-2 / 1.7e-6, x3 0.12 / 2.7e-5 (max |logit| 0.27 -> 2.3): the recurrence amplifies any rounding, the ratio
        # bf16 : fp32 stays 2 400 - 6 800.  So the 1e-3 of the north star is met by bf16 mode at init scale only; stated bound
        # here 5e-3 at x 1.5 (two planes everywhere, `bf16x2`, stay at 1e-5).
        with torch.no_grad():
            for n in m.variables.names():
                if n.endswith("/kernel") and ("basic_lstm_cell" in n):
                    m.variables[n].mul_(wscale)
    b = synthetic_batch(B=32, T=800, F=80, t_dec=121, vocab=1000, variable_len=True, seed=4321 + seed - 17)
    m.forward(b)
    ops.check_device_flag(torch.device(DEV))
    out = m.outputs["char"].cpu().numpy()
    w = _f64(m.variables.to_arrays())
    b64 = dict(b); b64["logmel"] = b["logmel"].astype(np.float64)
    r = O.seq2seq_forward(b64, w, is_training=True)
    err = np.abs(out - r["outputs"]["char"]).max()
    ops.set_decoder_bf16_planes(2)
    assert (1e-5 if prec == "bf16" else 0.0) < err < tol, err
    assert abs(m.total_loss.item() - r["total_loss"]) < 1e-2 * abs(r["total_loss"])
    print("config-3 (%s operands, weight seed %d, LSTM kernels x %g) full size: max |logit diff| = %.3g (bound %.0e, margin x%.2f)" % (
        prec, seed, wscale, err, tol, tol / max(err, 1e-30)))


def _lstm_ref_bf16(x, lens, k, b, reverse, rb):
    """float64 BasicLSTM layer (basic_lstm.py:14-23, dynamic_rnn masking) with the operand roundings of the library's bf16 mode:
    x, K_x, K_h and the h fed back into the recurrent product go through `rb` (bf16 round-to-nearest-even, or identity)."""
    B, T, IN = x.shape
    H = k.shape[1] // 4
    gx = rb(x).reshape(B * T, IN) @ rb(k[:IN]) + b
    gx = gx.reshape(B, T, 4 * H)
    kh = rb(k[IN:])
    out = torch.zeros(B, T, H, dtype=torch.float64)
    for bq in range(B):
        c = torch.zeros(H, dtype=torch.float64); h = torch.zeros(H, dtype=torch.float64)
        n = int(lens[bq])
        for s in range(n):
            t = n - 1 - s if reverse else s
            g = gx[bq, t] + rb(h.float()) @ kh
            i, j, f, o = g[:H], g[H:2 * H], g[2 * H:3 * H], g[3 * H:]
            c = c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
            h = torch.sigmoid(o) * torch.tanh(c)
            out[bq, t] = h
    return out


def test_recurrent_product_on_the_bf16_matrix_pipe():
    """In the library's bf16 mode the persistent forward recurrence (H = 256) forms h.K_h with v_mfma_f32_16x16x32_bf16: K_h and
    the fed-back h rounded to bf16, fp32 accumulation, fp32 cell.  Against a float64 layer with the SAME operand roundings the
    output agrees to fp32-level error; against the unrounded layer it differs by the bf16 operand rounding."""
    from e2e_asr_amd import ops
    rng = np.random.default_rng(9)
    B, T, IN, H = 4, 32, 64, 256        # B*T = 128: the input projection is a whole-tile product, i.e. on the bf16 path too
    x = torch.from_numpy(rng.standard_normal((B, T, IN)).astype(np.float32))
    kf = torch.from_numpy(rng.uniform(-0.075, 0.075, (IN + H, 4 * H)).astype(np.float32))
    kb = torch.from_numpy(rng.uniform(-0.075, 0.075, (IN + H, 4 * H)).astype(np.float32))
    bz = torch.zeros(4 * H)
    lens = np.array([32, 20, 1, 31])
    ops.set_gemm_precision("bf16")
    ops.set_lstm_mfma(True)          # (opt-in since round 3: by default bf16 mode keeps the fp32 version-2 recurrences)
    try:
        got = ops.lstm_layer_fwd(x.to(DEV), torch.from_numpy(lens.astype(np.int32)).to(DEV), kf.to(DEV), bz.to(DEV), kb.to(DEV),
                                 bz.to(DEV)).cpu().double()
    finally:
        ops.set_lstm_mfma(False)
    ops.check_device_flag(torch.device(DEV))
    rb = lambda v: v.to(torch.bfloat16).double()
    ident = lambda v: v.double()
    for d, k, rev in ((0, kf, False), (1, kb, True)):
        ref_b = _lstm_ref_bf16(x, lens, k, bz.double(), rev, rb)
        ref_e = _lstm_ref_bf16(x, lens, k, bz.double(), rev, ident)
        e_b = (got[:, :, d * H:(d + 1) * H] - ref_b).abs().max().item()
        e_e = (got[:, :, d * H:(d + 1) * H] - ref_e).abs().max().item()
        assert e_b < 5e-4 and e_e > 4 * e_b, (d, e_b, e_e)      # (a tie flipping under fp32 vs float64 h costs ~2^-9 of one element)
        for bq in range(B):
            assert not got[bq, lens[bq]:].any()


def test_bptt_contraction_on_the_bf16_matrix_pipe_close_to_fp32_path():
    """bf16 mode: the BPTT's contraction dG_{s-1}.K_h^T on v_mfma_f32_16x16x32_bf16 (K_h and the exchanged dG rounded to bf16,
    fp32 accumulation and fp32 pointwise backward).  Every gradient of the layer stays within the operand rounding of the fp32
    path: cosine >= 0.999, max error <= 3 % of the gradient's largest entry; both recurrence variants (R = 1, 2 rows per group)."""
    from e2e_asr_amd import ops
    rng = np.random.default_rng(19)
    for B in (4, 40):                      # 40 rows bidirectional = R 2 groups; 4 rows = R 1
        T, IN, H = 32, 64, 256
        x = torch.from_numpy(rng.standard_normal((B, T, IN)).astype(np.float32)).to(DEV)
        k = [torch.from_numpy(rng.uniform(-0.075, 0.075, (IN + H, 4 * H)).astype(np.float32)).to(DEV) for _ in range(2)]
        bz = torch.zeros(4 * H, device=DEV)
        lens = rng.integers(1, T + 1, B); lens[0] = T
        ln = torch.from_numpy(lens.astype(np.int32)).to(DEV)
        dout = torch.from_numpy(rng.standard_normal((B, T, 2 * H)).astype(np.float32)).to(DEV)
        res = {}
        for prec in ("f32", "bf16"):
            ops.set_gemm_precision(prec)
            ops.set_lstm_mfma(prec == "bf16")
            out, gates, act, hp = ops.lstm_layer_fwd(x, ln, k[0], bz, k[1], bz, save=True)
            dk = [torch.zeros_like(k[0]) for _ in range(2)]; db = [torch.zeros_like(bz) for _ in range(2)]
            dx = ops.lstm_layer_bwd(x, ln, k[0], k[1], dout, gates, act, hp, dk[0], db[0], dk[1], db[1], need_dx=True,
                                    kx_cat=getattr(gates, "kx_cat", None))
            ops.check_device_flag(torch.device(DEV))
            res[prec] = [dx.cpu().double()] + [t.cpu().double() for t in dk + db]
        ops.set_lstm_mfma(False)
        for g32, g16 in zip(res["f32"], res["bf16"]):
            cos = float((g32 * g16).sum() / (g32.norm() * g16.norm() + 1e-30))
            err = float((g32 - g16).abs().max() / g32.abs().max())
            assert cos > 0.999 and err < 3e-2, (B, cos, err)
