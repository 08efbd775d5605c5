"""GPU: the GEMMs on operands pre-split into bf16 planes (csrc/gemm_p3.hip, "P3" operands) -- the products of encoder.py:78-81
(input projections) and of tf.gradients through them (seq2seq_model.py:148) with the fp32 -> bf16-plane split taken out of the
k-loop.  Held to the error of the exact-fp32 MFMA kernel against float64 on the same operands (three planes), and to the product
of the rounded planes (one / two planes); the encoder layer run on plane operands end to end (ASR_P3=1: the recurrent kernels
write h, h_prev and dG as planes) must reproduce the default path's outputs and gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _err(c, ref, den):
    return ((c.double() - ref).abs() / den).max().item()


@pytest.mark.parametrize("M,N,K", [(128, 256, 16), (256, 512, 48), (1280, 256, 1024), (384, 768, 2048)])
@pytest.mark.parametrize("scale", ["normal", "wide"])
def test_kk_three_planes_is_fp32_accurate(M, N, K, scale):
    from e2e_asr_amd import ops
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = torch.randn(M, K, device=DEV, generator=g)
    b = torch.randn(N, K, device=DEV, generator=g)
    if scale == "wide":          # eight decades of magnitudes: every plane carries bits
        a = a * torch.pow(10.0, torch.rand(M, K, device=DEV, generator=g) * 8 - 4)
        b = b * torch.pow(10.0, torch.rand(N, K, device=DEV, generator=g) * 8 - 4)
    bias = torch.randn(N, device=DEV, generator=g)
    ref = a.double() @ b.double().t()
    den = a.double().abs() @ b.double().abs().t()
    c = ops.gemm_p3_kk(ops.p3_split(a, 3), ops.p3_split(b, 3), bias)
    e_p3 = _err(c - bias, ref, den)
    ops.set_gemm_split(False)
    try:
        c32 = ops.gemm(a, b, None, trans_b=True)
    finally:
        ops.set_gemm_split(True)
    e_32 = _err(c32, ref, den)
    assert e_p3 <= max(1.5 * e_32, 2e-7), (e_p3, e_32)
    # transposed split (the weights of the forward projection arrive as [K, N])
    c2 = ops.gemm_p3_kk(ops.p3_split(a, 3), ops.p3_split(b.t().contiguous(), 3, transpose=True), bias)
    assert torch.equal(c, c2)
    # accumulate, and K split over two workgroups per tile (float atomics into a live C)
    c3 = c.clone()
    ops.gemm_p3_kk(ops.p3_split(a, 3), ops.p3_split(b, 3), None, out=c3, accumulate=True)
    assert _err(c3 - c - bias * 0, ref, den) <= max(2.5 * e_32, 4e-7)
    if K >= 32:
        c4 = torch.empty_like(c)
        ops.gemm_p3_kk(ops.p3_split(a, 3), ops.p3_split(b, 3), bias, out=c4, splits=2)
        assert _err(c4 - bias, ref, den) <= max(1.5 * e_32, 2e-7)


@pytest.mark.parametrize("M,N,K", [(128, 256, 16), (256, 256, 400), (1280, 1024, 1600)])
def test_rr_three_planes_is_fp32_accurate(M, N, K):
    from e2e_asr_amd import ops
    g = torch.Generator(device=DEV).manual_seed(7 * M + N + K)
    a = torch.randn(K, M, device=DEV, generator=g)
    b = torch.randn(K, N, device=DEV, generator=g)
    ref = a.double().t() @ b.double()
    den = a.double().abs().t() @ b.double().abs()
    ops.set_gemm_split(False)
    try:
        c32 = ops.gemm(a, b, None, trans_a=True)
    finally:
        ops.set_gemm_split(True)
    e_32 = _err(c32, ref, den)
    for splits in (1, 0):
        c = torch.zeros(M, N, device=DEV)
        ops.gemm_p3_rr(ops.p3_split(a, 3), ops.p3_split(b, 3), out=c, splits=splits)
        # (fp32 accumulation over up to K terms in one chain when splits = 1: the exact-fp32 MFMA chain itself is at
        # 0.75-3.5e-7 of sum |a||b| for K = 1024 .. 4096, MI355X_MICROARCH.md)
        assert _err(c, ref, den) <= max(3.0 * e_32, 4e-7), splits
    # output column map (a producer that writes its columns unit-major for a gate-major C)
    perm = torch.randperm(N, device=DEV, generator=g).to(torch.int32)
    c = torch.zeros(M, N, device=DEV)
    ops.gemm_p3_rr(ops.p3_split(a, 3), ops.p3_split(b, 3), out=c, splits=1, colmap=perm)
    full = torch.zeros(M, N, device=DEV)
    ops.gemm_p3_rr(ops.p3_split(a, 3), ops.p3_split(b, 3), out=full, splits=1)
    assert torch.equal(c[:, perm.long()], full)


@pytest.mark.parametrize("np_", [1, 2])
def test_reduced_plane_counts_equal_the_product_of_the_rounded_planes(np_):
    from e2e_asr_amd import ops
    g = torch.Generator(device=DEV).manual_seed(11)
    M, N, K = 256, 512, 512
    a = torch.randn(M, K, device=DEV, generator=g)
    b = torch.randn(N, K, device=DEV, generator=g)

    def planes(x):
        out, r = [], x.clone()
        for _ in range(np_):
            h = r.to(torch.bfloat16).float()
            out.append(h.double())
            r = r - h
        return out
    pa, pb = planes(a), planes(b)
    ref = sum(pa[i] @ pb[j].t() for i in range(np_) for j in range(np_) if i + j <= np_ - 1)
    c = ops.gemm_p3_kk(ops.p3_split(a, np_), ops.p3_split(b, np_))
    den = a.double().abs() @ b.double().abs().t()
    assert _err(c, ref, den) < 3e-7
    a2, b2 = a.t().contiguous(), b.t().contiguous()           # RR: [K, M], [K, N]
    c = torch.zeros(M, N, device=DEV)
    ops.gemm_p3_rr(ops.p3_split(a2, np_), ops.p3_split(b2, np_), out=c, splits=1)
    assert _err(c, ref, den) < 3e-7


@pytest.mark.parametrize("K", [16, 48, 400, 512])
def test_rr_one_plane_keeps_every_contraction_row(K):
    """One plane runs two MFMA k-steps per stage (32 contraction rows); K % 32 == 16 must not drop the last 16 rows (advisor,
    round 4: asr_gemm_p3_rr accepted K % 16 == 0 and launched the 32-row kernel).  The last 16 rows carry the largest values."""
    from e2e_asr_amd import ops
    g = torch.Generator(device=DEV).manual_seed(K)
    M, N = 128, 256
    a = torch.randn(K, M, device=DEV, generator=g)
    b = torch.randn(K, N, device=DEV, generator=g)
    a[-16:] *= 8.0
    ar, br = a.to(torch.bfloat16).double(), b.to(torch.bfloat16).double()
    ref = ar.t() @ br
    den = ar.abs().t() @ br.abs()
    for splits in (1, 0):
        c = torch.zeros(M, N, device=DEV)
        ops.gemm_p3_rr(ops.p3_split(a, 1), ops.p3_split(b, 1), out=c, splits=splits)
        assert _err(c, ref, den) < 3e-7, (K, splits)


def test_split_layouts():
    """asr_p3_split_ex: padded image, transposed image, unit-major column permutation -- against the layout formula."""
    from e2e_asr_amd import ops
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(24, 80, device=DEV, generator=g)

    def decode(p3):       # P3 image -> float64 [rows, cols] (sum of the planes)
        raw = p3.buf.view(torch.int16).view(p3.rows, p3.cols // 8, p3.np, 8)
        f = (raw.to(torch.int32) << 16).view(torch.float32)
        return f.double().sum(2).reshape(p3.rows, p3.cols)
    d = decode(ops.p3_split(x, 3, cols=128))
    assert torch.equal(d[:, :80].float(), x) and (d[:, 80:] == 0).all()
    for cols in (83, 39, 43):          # rows that do not start 16-byte aligned (fbank + pitch, MFCC): the scalar-load branch
        xo = torch.randn(24, cols, device=DEV, generator=g)
        d = decode(ops.p3_split(xo, 3, cols=128))
        assert torch.equal(d[:, :cols].float(), xo) and (d[:, cols:] == 0).all()
    assert torch.equal(decode(ops.p3_split(x, 3, transpose=True)).float(), x.t())
    H = 8
    w = torch.randn(16, 2 * 4 * H, device=DEV, generator=g)
    du = decode(ops.p3_split(w, 3, unit_major_h=H)).float()
    for dd in range(2):
        for u in range(H):
            for gg in range(4):
                assert torch.equal(du[:, dd * 4 * H + 4 * u + gg], w[:, dd * 4 * H + gg * H + u])
    cm = ops.p3_colmap(H, torch.device(DEV)).cpu().numpy()
    assert [int(cm[4 * u + gg]) for u in range(H) for gg in range(4)] == [gg * H + u for u in range(H) for gg in range(4)]


def test_encoder_on_plane_operands_equals_default_path(monkeypatch):
    """ASR_P3=1: the groups-of-four recurrent kernels write h / h_prev / dG as planes and every encoder GEMM of layers >= 2 (and
    the first layer's weight gradients) runs on plane operands.  Same arithmetic as the default path's split3 products, other
    summation order: logits to 2e-5, every gradient to 2e-4 of its largest entry, with dropout and ragged lengths."""
    from tests.test_gpu_parity3 import _model
    from e2e_asr_amd import ops
    from e2e_asr_amd.weights import synthetic_batch
    b = synthetic_batch(B=32, T=160, F=80, t_dec=21, vocab=1000, variable_len=True, seed=100)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("ASR_P3", flag)
        m = _model(feat=80, vocab={"char": 1000}, max_output={"char": 30}, seed=6, enc_update=dict(out_prob=0.9))
        m.forward(b); m.backward()
        torch.cuda.synchronize()
        ops.check_device_flag(torch.device(DEV))
        res[flag] = (m.outputs["char"].cpu().numpy().copy(), {n: m.variables.grad_of(n).cpu().numpy().copy() for n in m.variables.names()})
    assert np.abs(res["1"][0] - res["0"][0]).max() < 2e-5
    for n, g0 in res["0"][1].items():
        assert np.abs(res["1"][1][n] - g0).max() <= 2e-4 * max(1e-30, np.abs(g0).max()), n


@pytest.mark.parametrize("B,T", [(64, 64), (40, 64), (32, 96)])
def test_bf16_mode_plane_path_equals_fp32_operand_path(monkeypatch, B, T):
    """bf16 mode (BASELINE config 3) runs the encoder's GEMMs on bf16 activations written by the recurrent kernels by default
    (one plane).  Same roundings as the path that rounds fp32 operands inside the GEMM (ASR_P3=0) -- other summation order: logits
    to 1e-4, gradients to 1e-3 of their largest entry.  B = 64 / 40: two launches of the recurrent kernels per layer (row ranges,
    plane pointers offset per range; 40 rows: a partial second range); ragged lengths, dropout on."""
    from tests.test_gpu_parity3 import _model
    from e2e_asr_amd import ops
    from e2e_asr_amd.weights import synthetic_batch
    b = synthetic_batch(B=B, T=T, F=80, t_dec=13, vocab=1000, variable_len=True, seed=7)
    ops.set_gemm_precision("bf16")
    try:
        res = {}
        for flag in ("0", None):
            if flag is None:
                monkeypatch.delenv("ASR_P3", raising=False)
            else:
                monkeypatch.setenv("ASR_P3", flag)
            assert ops.p3_planes() == (0 if flag == "0" else 1)
            m = _model(feat=80, vocab={"char": 1000}, max_output={"char": 20}, seed=9, enc_update=dict(out_prob=0.9))
            m.forward(b); m.backward()
            torch.cuda.synchronize()
            ops.check_device_flag(torch.device(DEV))
            res[flag] = (m.outputs["char"].cpu().numpy().copy(), {n: m.variables.grad_of(n).cpu().numpy().copy() for n in m.variables.names()})
    finally:
        ops.set_gemm_precision("f32")
    assert np.abs(res[None][0] - res["0"][0]).max() < 1e-4
    for n, g0 in res["0"][1].items():
        assert np.abs(res[None][1][n] - g0).max() <= 1e-3 * max(1e-30, np.abs(g0).max()), n


@pytest.mark.parametrize("feat", [83, 39])
def test_bf16_mode_takes_any_feat_length(monkeypatch, feat):
    """bf16 mode builds a 128-column plane image of the frames for the first layer's weight gradient; feat_length 83 (fbank +
    pitch) or 39 (MFCC) gives rows that are not 16-byte aligned (advisor, round 4: the split refused them and the train step
    raised).  The reference takes any --feat_length.  Held to the fp32-operand bf16 path as above."""
    from tests.test_gpu_parity3 import _model
    from e2e_asr_amd import ops
    from e2e_asr_amd.weights import synthetic_batch
    b = synthetic_batch(B=32, T=64, F=feat, t_dec=13, vocab=1000, variable_len=True, seed=17)
    ops.set_gemm_precision("bf16")
    try:
        res = {}
        for flag in ("0", None):
            if flag is None:
                monkeypatch.delenv("ASR_P3", raising=False)
            else:
                monkeypatch.setenv("ASR_P3", flag)
            m = _model(feat=feat, vocab={"char": 1000}, max_output={"char": 20}, seed=9, enc_update=dict(out_prob=0.9))
            m.forward(b); m.backward()
            torch.cuda.synchronize()
            ops.check_device_flag(torch.device(DEV))
            res[flag] = (m.outputs["char"].cpu().numpy().copy(), {n: m.variables.grad_of(n).cpu().numpy().copy() for n in m.variables.names()})
            m.apply_gradients()
            m.step(b)                      # and a whole step through the arena path
            torch.cuda.synchronize()
            ops.check_device_flag(torch.device(DEV))
    finally:
        ops.set_gemm_precision("f32")
    assert np.abs(res[None][0] - res["0"][0]).max() < 1e-4
    # (a ragged input width sends the fp32-operand path's first-layer products through the edge kernels, which do not round
    # their operands to bf16: the two paths then differ by the bf16 rounding itself, 2^-8 of the largest entry -- measured 2e-3)
    for n, g0 in res["0"][1].items():
        assert np.abs(res[None][1][n] - g0).max() <= 4e-3 * max(1e-30, np.abs(g0).max()), n


def _busy_stream(n_launches=24):
    """Keep every CU busy from a second stream: split3 GEMMs (two resident workgroups per CU, LDS + MFMA + global loads) queued
    back to back -- ~4 ms of work the products under test have to share the chip with."""
    from e2e_asr_amd import ops
    side = torch.cuda.Stream()
    a = torch.randn(4096, 2048, device=DEV)
    b = torch.randn(2048, 4096, device=DEV)
    c = torch.empty(4096, 4096, device=DEV)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(n_launches):
            ops.gemm(a, b, out=c)
    return side, (a, b, c)


@pytest.mark.parametrize("np_,M,N,K", [(1, 12800, 1024, 1024),      # one plane, 256 x 256 tiles of eight waves (the config-3 projections)
                                       (1, 12928, 1024, 1024),      # one plane, 128 x 256 tiles (M % 256 != 0)
                                       (3, 12800, 1024, 1024),      # three planes, 128 x 256 tiles
                                       (1, 6400, 2048, 2048)])
def test_kk_products_are_bit_stable_at_full_occupancy(np_, M, N, K):
    """The KK form has no K split and no atomics: the same operands must give the same BITS on every call -- alone, and while
    another stream keeps all CUs busy (the condition under which round 3's LDS-DMA ring in the BPTT drifted run to run).  A
    fragment read that overtakes the LDS-DMA of its stage (the ordering scripts/check_p3_isa.py checks on the ISA) shows here as
    a run-to-run difference.  Arithmetic held: encoder.py:78-81 (the layer's input projection)."""
    from e2e_asr_amd import ops
    g = torch.Generator(device=DEV).manual_seed(M + np_)
    a = torch.randn(M, K, device=DEV, generator=g)
    b = torch.randn(N, K, device=DEV, generator=g)
    bias = torch.randn(N, device=DEV, generator=g)
    ap, bp = ops.p3_split(a, np_), ops.p3_split(b, np_)
    quiet = ops.gemm_p3_kk(ap, bp, bias).clone()
    torch.cuda.synchronize()
    side, keep = _busy_stream()
    outs = [ops.gemm_p3_kk(ap, bp, bias).clone() for _ in range(6)]       # interleaved with the other stream's workgroups
    torch.cuda.synchronize()
    del keep
    for i, o in enumerate(outs):
        assert torch.equal(o, quiet), "call %d differs from the quiet call in %d entries" % (i, int((o != quiet).sum()))
    ops.check_device_flag(torch.device(DEV))


def test_config3_forward_is_bit_stable_call_to_call():
    """BASELINE config 3 at its full size (32 x 800 x 80, bf16 mode: the encoder's projections on LDS-DMA-staged plane operands):
    two forward passes of the same step give identical logits bit for bit, the second one next to a busy second stream."""
    from tests.test_gpu_parity3 import _model
    from e2e_asr_amd import ops
    from e2e_asr_amd.weights import synthetic_batch
    b = synthetic_batch(B=32, T=800, F=80, t_dec=121, vocab=1000, variable_len=False, seed=1234)
    ops.set_gemm_precision("bf16")
    try:
        m = _model(feat=80, vocab={"char": 1000}, max_output={"char": 120}, seed=10, enc_update=dict(out_prob=0.9),
                   dec_update=dict(out_prob_dec=0.9, samp_prob=0.1))
        m.forward(b)
        first = m.outputs["char"].clone()
        loss0 = float(m.losses["char"])
        torch.cuda.synchronize()
        side, keep = _busy_stream()
        m.forward(b)
        second = m.outputs["char"].clone()
        loss1 = float(m.losses["char"])
        torch.cuda.synchronize()
        del keep
    finally:
        ops.set_gemm_precision("f32")
    ops.check_device_flag(torch.device(DEV))
    assert torch.equal(first, second), "%d of %d logits differ" % (int((first != second).sum()), first.numel())
    assert loss0 == loss1
