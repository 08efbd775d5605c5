"""GPU: the fp32 GEMM on the bf16 matrix pipe (exact 3-way operand split, six MFMA products -- csrc/gemm.hip
gemm_split3_kernel) must be an fp32-accurate GEMM: its error against a float64 product is held to the error of the
exact-fp32 MFMA kernel (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain) on the same operands, for every operand layout the
path uses (NN input projections, NT data gradients, TN split-K weight gradients), and it is exact where fp32 is."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(a, b, ta, tb, split, bias=None, accumulate=False, c0=None):
    from e2e_asr_amd import ops
    prev = ops.get_gemm_split()
    ops.set_gemm_split(split)
    try:
        out = None if c0 is None else c0.clone()
        return ops.gemm(a, b, bias, ta, tb, out=out, accumulate=accumulate)
    finally:
        ops.set_gemm_split(prev)


@pytest.mark.parametrize("form,M,N,K", [("NN", 1024, 1024, 1024), ("NN", 12800, 1024, 80 * 16), ("NT", 2048, 1024, 2048),
                                        ("TN", 1024, 1024, 12800), ("TN", 256, 1024, 25600),
                                        ("TN", 80, 1024, 25600), ("TN", 200, 256, 12800),
                                        # the 64x64-tile kernel (gemm_split3s_kernel): the decoder's shapes, K = V = 1000 (not a
                                        # multiple of the 16-wide k-tile), one- and two-tile contractions
                                        ("NN", 3840, 256, 512), ("NT", 3840, 256, 1000), ("NN", 768, 1024, 256),
                                        ("TN", 512, 640, 200), ("NT", 256, 256, 24), ("NN", 256, 320, 16),
                                        ("NN", 3808, 256, 512), ("NN", 3840, 1000, 256), ("NT", 250, 1000, 36), ("TN", 100, 1000, 256)])
@pytest.mark.parametrize("dist", ["normal", "wide"])
def test_split_gemm_error_is_fp32_class(form, M, N, K, dist):
    rng = np.random.default_rng(hash((form, M, N, K, dist)) & 0xFFFF)
    ta, tb = form[0] == "T", form[1] == "T"
    def draw(shape):
        x = rng.standard_normal(shape)
        if dist == "wide":                      # eight decades of magnitude, signs mixed: cancellation + tiny residual planes
            x = x * np.exp(rng.uniform(-9, 9, shape))
        return x.astype(np.float32)
    a = draw((K, M) if ta else (M, K)); b = draw((N, K) if tb else (K, N))
    bias = draw((N,))
    A64 = (a.T if ta else a).astype(np.float64); B64 = (b.T if tb else b).astype(np.float64)
    ref = A64 @ B64 + bias
    scale = np.abs(A64) @ np.abs(B64) + np.abs(bias)          # the quantity fp32 error bounds are relative to
    at, bt, biast = (torch.from_numpy(x).to(DEV) for x in (a, b, bias))
    got_s = _run(at, bt, ta, tb, True, biast).cpu().numpy().astype(np.float64)
    got_e = _run(at, bt, ta, tb, False, biast).cpu().numpy().astype(np.float64)
    err_s = (np.abs(got_s - ref) / scale).max()
    err_e = (np.abs(got_e - ref) / scale).max()
    eps = 2.0 ** -24
    # an fp32 dot product of length K in any order: |error| <= ~K eps sum|a||b| worst case, ~sqrt(K) eps typically
    assert err_e < 8 * np.sqrt(K) * eps and err_s < 8 * np.sqrt(K) * eps, (err_s, err_e)
    assert err_s < 3.0 * err_e + 4 * eps, (err_s, err_e)      # no worse than the exact-fp32 kernel, up to order effects
    rms_s = np.sqrt((((got_s - ref) / scale) ** 2).mean()); rms_e = np.sqrt((((got_e - ref) / scale) ** 2).mean())
    assert rms_s < 2.0 * rms_e + eps, (rms_s, rms_e)
    print("%s %dx%dx%d %s: split max %.2e rms %.2e | exact max %.2e rms %.2e (units of sum|a||b|)" % (
        form, M, N, K, dist, err_s, rms_s, err_e, rms_e))


def test_split_gemm_is_exact_where_fp32_is():
    """Integer-valued operands whose products and sums fit 24 bits: every fp32 summation order gives the exact result, and
    so must the split (the three planes reproduce every operand bit); also accumulate=True and the operand split of values
    that need all 24 significand bits."""
    rng = np.random.default_rng(7)
    M = N = 256; K = 512
    a = rng.integers(-2047, 2048, (M, K)).astype(np.float32)       # 12-bit integers: need two bf16 planes
    b = rng.integers(-3, 4, (K, N)).astype(np.float32)
    c0 = rng.integers(-100, 100, (M, N)).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64) + c0
    assert np.abs(ref).max() < 2 ** 24
    got = _run(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), False, False, True, accumulate=True,
               c0=torch.from_numpy(c0).to(DEV)).cpu().numpy()
    np.testing.assert_array_equal(got, ref.astype(np.float32))
    # 24-bit operands x powers of two: the product is the operand itself, bit for bit, only if all three planes are right
    a = (rng.integers(2 ** 23, 2 ** 24, (256, 128)) * rng.choice([-1.0, 1.0], (256, 128))).astype(np.float32)
    b = np.zeros((128, 128), np.float32); b[np.arange(128), np.arange(128)] = 2.0 ** rng.integers(-20, 20, 128)
    got = _run(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), False, False, True).cpu().numpy()
    np.testing.assert_array_equal(got, a * np.diag(b)[None, :])


def test_split_is_the_default_and_small_products_stay_on_the_exact_kernel():
    from e2e_asr_amd import ops
    assert ops.get_gemm_split() is True
    rng = np.random.default_rng(3)
    # K not a multiple of 4 (no 16-byte rows) / fewer than 8 tiles: the fp32-input MFMA kernel either way
    a = torch.from_numpy(rng.standard_normal((200, 98)).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rng.standard_normal((98, 72)).astype(np.float32)).to(DEV)
    assert torch.equal(_run(a, b, False, False, True), _run(a, b, False, False, False))
    a = torch.from_numpy(rng.standard_normal((100, 96)).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rng.standard_normal((96, 60)).astype(np.float32)).to(DEV)
    assert torch.equal(_run(a, b, False, False, True), _run(a, b, False, False, False))


@pytest.mark.parametrize("form,M,N,K,prec", [
    ("NN", 12800, 2048, 1024, 0),       # config-2 layer-2 input projection: 1 600 tiles, 3.125 per workgroup slot
    ("NN", 6400, 1024, 1024, 0),        # 400 tiles: less than one round, runs shorter than a tile (middle contributors)
    ("NT", 3200, 1024, 2048, 0),        # data gradient of the top layer: 200 tiles, 128 k-tiles each, three or four workgroups per tile
    ("NT", 12800, 1024, 2048, 0),       # ... of layer 2: 800 tiles
    ("NN", 6400, 1024, 1024, 1),        # one bf16 plane (BK = 32)
    ("NN", 3200, 2048, 1024, 2),        # two planes
    ("NN", 1024 + 128, 1024, 256, 0),   # 72 tiles: 9 per XCD, 16 k-tiles -> at most 18 workgroups per XCD ... few units per run
])
def test_stream_k_equals_whole_tile_kernel_and_is_reproducible(monkeypatch, form, M, N, K, prec):
    """gemm_planes_kernel<..., SK> (every workgroup an equal run of k-tiles, partial tiles summed in a fixed order) against the
    whole-tile launch of the same kernel (ASR_GEMM_SK=0) and float64: same error class, bit-identical from run to run, bias and
    accumulation applied once."""
    from e2e_asr_amd import ops
    rng = np.random.default_rng(M + N + K + prec)
    ta, tb = form[0] == "T", form[1] == "T"
    a = rng.standard_normal((M, K)).astype(np.float32); b = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    bias = rng.standard_normal((N,)).astype(np.float32)
    c0 = rng.standard_normal((M, N)).astype(np.float32)
    at, bt, biast, c0t = (torch.from_numpy(x).to(DEV) for x in (a, b, bias, c0))
    prev = ops.get_gemm_precision()
    ops.set_gemm_precision({0: "f32", 1: "bf16", 2: "bf16x2"}[prec])
    try:
        monkeypatch.setenv("ASR_GEMM_SK", "1")
        sk1 = _run(at, bt, ta, tb, True, biast)
        sk2 = _run(at, bt, ta, tb, True, biast)
        sk_acc = _run(at, bt, ta, tb, True, biast, accumulate=True, c0=c0t)
        monkeypatch.setenv("ASR_GEMM_SK", "0")
        wt = _run(at, bt, ta, tb, True, biast)
    finally:
        ops.set_gemm_precision(prev)
    torch.cuda.synchronize()
    assert torch.equal(sk1, sk2)
    B64 = (b.T if tb else b).astype(np.float64)
    ref = a.astype(np.float64) @ B64 + bias
    scale = np.abs(a).astype(np.float64) @ np.abs(B64) + np.abs(bias)
    e_sk = (np.abs(sk1.cpu().numpy() - ref) / scale).max(); e_wt = (np.abs(wt.cpu().numpy() - ref) / scale).max()
    tol = {0: 8 * np.sqrt(K) * 2.0 ** -24, 1: 2.0 ** -7, 2: 2.0 ** -14}[prec]
    assert e_sk < tol and e_wt < tol, (e_sk, e_wt)
    assert e_sk < 3 * e_wt + 2.0 ** -22, (e_sk, e_wt)
    # the two differ by summation order in the tiles several workgroups share, nowhere by more than fp32 rounding of the terms
    np.testing.assert_allclose(sk1.cpu().numpy(), wt.cpu().numpy(), rtol=0, atol=float(16 * 2.0 ** -24 * scale.max()))
    np.testing.assert_allclose(sk_acc.cpu().numpy(), sk1.cpu().numpy() + c0, rtol=0, atol=float(4 * 2.0 ** -24 * (scale.max() + np.abs(c0).max())))
