"""Thin torch-tensor wrappers over the C ABI (include/e2e_asr_hip.h).

torch is plumbing here: it owns device memory and the stream.  Every function
below launches hand-written gfx950 kernels through ctypes; none has a fallback.
Errors follow the reference's Python-exception convention: invalid arguments raise
ValueError, launch failures RuntimeError.
"""
import ctypes as C
import os

import torch

from . import _lib

_ERR = {-1: (ValueError, "invalid argument"), -2: (RuntimeError, "kernel launch failed"),
        -3: (ValueError, "unsupported shape")}


def _check(rc, what):
    if rc != 0:
        exc, msg = _ERR.get(rc, (RuntimeError, "error %d" % rc))
        raise exc("%s: %s" % (what, msg))


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def default_device():
    """Device of the module-level helpers that take NumPy arrays (num_utils, BasicLSTM): the current CUDA device.
    There is no CPU path."""
    if not torch.cuda.is_available():
        raise RuntimeError("e2e_asr_amd: no GPU visible; the HIP path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, name):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError("%s must be a contiguous float32 CUDA tensor" % name)
    return t


def _i32(t, name):
    if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
        raise ValueError("%s must be a contiguous int32 CUDA tensor" % name)
    return t


def gemm(a, b, bias=None, trans_a=False, trans_b=False, out=None, accumulate=False):
    """out[M,N] (+)= op(a) @ op(b) + bias.  2-D float32 CUDA tensors (row stride = size(1))."""
    _f32(a, "a"); _f32(b, "b"); _f32(bias, "bias")
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    Kb, N = (b.shape[1], b.shape[0]) if trans_b else b.shape
    if K != Kb:
        raise ValueError("gemm: inner dimensions differ (%d vs %d)" % (K, Kb))
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    _f32(out, "out")
    rc = _lib.lib().asr_gemm_f32(_stream(), int(trans_a), int(trans_b), M, N, K, _p(a), a.shape[1],
                                 _p(b), b.shape[1], _p(out), out.shape[1], _p(bias), int(accumulate))
    _check(rc, "asr_gemm_f32")
    return out


# ---- operands pre-split into bf16 planes ("P3", csrc/gemm_p3.hip) ------------------------------------------------------------
class P3(object):
    """P3 image of a logical [rows, cols] matrix: uint8 CUDA buffer + shape + plane count (include/e2e_asr_hip.h)."""

    def __init__(self, buf, rows, cols, np_):
        self.buf, self.rows, self.cols, self.np = buf, rows, cols, np_

    @property
    def ld8(self):
        return self.cols // 8


def p3_split(x, np_=3, transpose=False, out=None, cols=0, unit_major_h=0):
    """fp32 [R, C] -> P3 image of x (or of x^T).  cols > the logical column count: a zero-padded, tile-aligned image;
    unit_major_h = h: image column d*4h + 4u + g holds source column d*4h + g*h + u (asr_p3_split_ex)."""
    _f32(x, "x")
    R, Cc = x.shape
    rows, lc = (Cc, R) if transpose else (R, Cc)
    cols = cols if cols else (lc + 7) // 8 * 8
    if out is None:
        out = P3(torch.empty(int(_lib.lib().asr_p3_bytes(rows, cols, np_)), device=x.device, dtype=torch.uint8), rows, cols, np_)
    _check(_lib.lib().asr_p3_split_ex(_stream(), _p(x), R, Cc, x.shape[1], _p(out.buf), np_, int(transpose), cols, int(unit_major_h)),
           "asr_p3_split_ex")
    return out


def p3_split_many(jobs):
    """[(x, np, transpose, unit_major_h)] -> [P3]: up to 8 splits per launch (asr_p3_split_multi)."""
    outs = []
    for c0 in range(0, len(jobs), 8):
        chunk = jobs[c0:c0 + 8]
        arr = (_lib.P3SplitJob * len(chunk))()
        for i, (x, np_, tr, umh) in enumerate(chunk):
            _f32(x, "x")
            R, Cc = x.shape
            rows, cols = (Cc, R) if tr else (R, Cc)
            o = p3_alloc(rows, (cols + 7) // 8 * 8, np_, x.device)
            outs.append(o)
            arr[i] = _lib.P3SplitJob(_p(x), R, Cc, x.shape[1], _p(o.buf), np_, int(tr), o.cols, int(umh))
        _check(_lib.lib().asr_p3_split_multi(_stream(), len(chunk), arr), "asr_p3_split_multi")
    return outs


def p3_alloc(rows, cols, np_, dev):
    return P3(torch.empty(int(_lib.lib().asr_p3_bytes(rows, cols, np_)), device=dev, dtype=torch.uint8), rows, cols, np_)


def p3_planes():
    """Planes per value the encoder's P3 GEMMs run with, 0 = plane operands off (the fp32 operands are split inside the GEMM).
    Default (round 4, measured in the train step -- DESIGN section 4b): ON in bf16 mode (one plane = plain bf16 activations
    written by the recurrent kernels: BASELINE config 3, 6.97 -> 6.70 ms per step), OFF in fp32 mode (three planes: the
    producers' 6 bytes per value and the split passes cost what the faster k-loop gains; 8.14 against 7.85 ms) and in bf16x2.
    ASR_P3=1 / ASR_P3=0 force it on (3 / 2 / 1 planes by mode; not with the exact-fp32 MFMA kernels) / off."""
    e = os.environ.get("ASR_P3")
    mode = get_gemm_precision()
    if e == "0":
        return 0
    if mode == "bf16":
        return 1
    if e != "1":
        return 0
    if mode == "f32":
        return 3 if get_gemm_split() else 0
    return 2


_colmap_cache = {}


def p3_colmap(H, dev):
    """unit-major column 4u + g -> gate-major column g*H + u (one direction of a TF LSTM kernel)."""
    key = (H, dev)
    if key not in _colmap_cache:
        c = torch.arange(4 * H, dtype=torch.int32)
        _colmap_cache[key] = ((c % 4) * H + c // 4).to(torch.int32).to(dev)
    return _colmap_cache[key]


def lstm_p3_supported(B, T, IN, H, ndir):
    return bool(_lib.lib().asr_lstm_p3_supported(B, T, IN, H, ndir))


def _lstm_p3_struct(p3):
    """dict(np, x (P3), kxT (P3), out (P3), hprev (P3), dg (P3), kxu (P3), colmap) -> ctypes struct (kept alive by the caller)."""
    if p3 is None:
        return None
    g = lambda k: _p(p3[k].buf) if p3.get(k) is not None else None
    return _lib.LstmP3(int(p3["np"]), g("x"), int(p3["x"].cols) if p3.get("x") is not None else 0, g("kxT"), g("out"), g("hprev"),
                       g("dg"), g("kxu"), _p(p3.get("colmap")))


def gemm_p3_kk(a, b, bias=None, out=None, accumulate=False, splits=1):
    """out[M, N] (+)= A . B^T + bias with A = P3 [M, K], B = P3 [N, K]."""
    if a.cols != b.cols or a.np != b.np:
        raise ValueError("gemm_p3_kk: operands disagree (K %d vs %d, planes %d vs %d)" % (a.cols, b.cols, a.np, b.np))
    if out is None:
        out = torch.empty((a.rows, b.rows), device=a.buf.device, dtype=torch.float32)
    _f32(out, "out"); _f32(bias, "bias")
    _check(_lib.lib().asr_gemm_p3_kk(_stream(), a.rows, b.rows, a.cols, _p(a.buf), a.ld8, _p(b.buf), b.ld8, a.np,
                                     _p(out), out.shape[1], _p(bias), int(accumulate), int(splits)), "asr_gemm_p3_kk")
    return out


def gemm_p3_rr(a, b, out=None, accumulate=False, splits=0, colmap=None):
    """out[M, N] (+)= A^T . B with A = P3 [K, M], B = P3 [K, N] (contraction over the rows of both)."""
    if a.rows != b.rows or a.np != b.np:
        raise ValueError("gemm_p3_rr: operands disagree (K %d vs %d, planes %d vs %d)" % (a.rows, b.rows, a.np, b.np))
    if out is None:
        out = torch.zeros((a.cols, b.cols), device=a.buf.device, dtype=torch.float32)
        accumulate = True
    _f32(out, "out")
    _check(_lib.lib().asr_gemm_p3_rr(_stream(), a.cols, b.cols, a.rows, _p(a.buf), a.ld8, _p(b.buf), b.ld8, a.np,
                                     _p(out), out.shape[1], int(accumulate), int(splits), _p(colmap)), "asr_gemm_p3_rr")
    return out


class _Flag:
    """Device int that kernels set when an inter-workgroup wait times out."""
    _t = {}

    @classmethod
    def get(cls, dev):
        if dev not in cls._t:
            cls._t[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
        return cls._t[dev]


def check_device_flag(dev):
    """Raise if any persistent kernel on `dev` reported a timeout (synchronises)."""
    f = _Flag.get(dev)
    code = int(f.item())
    if code != 0:
        f.zero_()
        where = {11: "lstm_rec_fwd poll", 12: "lstm_rec_fwd pair poll", 13: "lstm_rec_bwd poll", 21: "decoder_chain_bwd poll",
                 41: "granule pair poll", 42: "quad poll", 51: "decoder_chain wait", 52: "decoder_chain_bwd wait a",
                 53: "decoder_chain_bwd wait b", 54: "decoder_chain_bwd wait c", 55: "lstm_rec_bwd wait",
                 61: "beam_persist grid barrier"}.get(code)
        if where is None and code % 100 == 31:
            where = "XCC agreement at the start of " + {0: "?", 1: "decoder_chain_fwd", 2: "decoder_chain_bwd", 3: "decoder_greedy / train",
                                                        4: "lstm_rec_fwd", 5: "lstm_rec_bwd (reduce-scatter)", 6: "lstm_rec_bwd (all-gather)"}.get(code // 100, "?")
        where = where or "code %d" % code
        raise RuntimeError("e2e_asr_amd: persistent LSTM kernel timed out waiting for a peer workgroup (%s)" % where)


_hx_cache = {}
_KXCAT = int(os.environ.get("ASR_KXCAT", "2"))      # EXPERIMENT: 0 = one GEMM per direction, 1 = fused forward projection, 2 = + fused dX


LSTM_KERNEL_H = (64, 128, 256, 512)     # instantiated widths of the persistent recurrent kernels (csrc/lstm.hip)


def _padded_h(H):
    for hp in LSTM_KERNEL_H:
        if H <= hp:
            return hp
    raise ValueError("hidden size %d: the recurrent kernels go up to %d units" % (H, LSTM_KERNEL_H[-1]))


def _pad_lstm_weights(kernel, bias, IN, H, Hp):
    """TF kernel [IN+H,4H] / bias [4H] (gates i,j,f,o) embedded in a width-Hp cell: the extra units have all-zero
    weights, so their state stays c = 0, h = sigmoid(0).tanh(0) = 0 for ever and they feed nothing into the real units
    (their K_h ROWS are zero too) -- the first H units compute exactly the width-H cell."""
    kp = kernel.new_zeros((IN + Hp, 4, Hp))
    k3 = kernel.view(IN + H, 4, H)
    kp[:IN, :, :H] = k3[:IN]
    kp[IN:IN + H, :, :H] = k3[IN:]
    bp = None
    if bias is not None:
        bp = bias.new_zeros((4, Hp))
        bp[:, :H] = bias.view(4, H)
        bp = bp.view(4 * Hp)
    return kp.view(IN + Hp, 4 * Hp), bp


def concat_kx_layers(layers):
    """layers: [(kernel_fw, bias_fw, kernel_bw, bias_bw)] of up to four BiLSTM layers.  Returns [(kx_cat [in, 8H], bias_cat [8H])]:
    the input rows of the two directions' kernels side by side (both directions' input projection, and dX, as ONE product)
    -- all of them built by ONE launch (asr_concat2_multi) instead of two torch.cat launches per layer."""
    assert 0 < len(layers) <= 4
    dev = layers[0][0].device
    a, b, dst, rows, wa, wb, lda, ldb, out = [], [], [], [], [], [], [], [], []
    for kf, bf, kb, bb in layers:
        H4 = kf.shape[1]
        IN = kf.shape[0] - H4 // 4
        kx = torch.empty((IN, 2 * H4), device=dev, dtype=torch.float32)
        bc = torch.empty((2 * H4,), device=dev, dtype=torch.float32)
        for (s0, s1, d, r, w0, w1, l0, l1) in ((_f32(kf, "kernel_fw"), _f32(kb, "kernel_bw"), kx, IN, H4, H4, H4, H4),
                                               (_f32(bf, "bias_fw"), _f32(bb, "bias_bw"), bc, 1, H4, H4, H4, H4)):
            a.append(s0.data_ptr()); b.append(s1.data_ptr()); dst.append(d.data_ptr())
            rows.append(r); wa.append(w0); wb.append(w1); lda.append(l0); ldb.append(l1)
        out.append((kx, bc))
    n = len(a)
    P, I = C.c_void_p * n, C.c_int * n
    rc = _lib.lib().asr_concat2_multi(_stream(), n, P(*a), P(*b), P(*dst), I(*rows), I(*wa), I(*wb), I(*lda), I(*ldb))
    _check(rc, "asr_concat2_multi")
    return out


def lstm_layer_fwd(x, seq_len, kernel_fw, bias_fw, kernel_bw=None, bias_bw=None, t_out=None,
                   save=False, keep_prob=1.0, seed=0, kx_cat=None, bias_cat=None, p3=None):
    """One (Bi)LSTM layer (encoder.py:55-91).  x [B,T,in] batch-major, seq_len int32 [B].

    Returns out [B,t_out,ndir*H] (zeros past each length) and, when save=True, the
    gates workspace, activation records [B,T,ndir,H,8] and hprev [B,T,ndir,H] for the backward pass.

    Any hidden size up to 512 (encoder.py:188-189 takes any -hsize): widths the persistent kernels are not instantiated
    for run zero-padded to the next instantiated width (exact, see _pad_lstm_weights; the recurrence is latency-bound, so
    the padding costs little), the saved tensors are then in the padded width.  Dropout there draws its mask over the
    padded column index, i.e. a different but equally distributed mask.
    """
    _f32(x, "x"); _i32(seq_len, "seq_len")
    B, T, IN = x.shape
    H = kernel_fw.shape[1] // 4
    ndir = 1 if kernel_bw is None else 2
    if kernel_fw.shape[0] != IN + H:
        raise ValueError("lstm kernel rows %d != in+H = %d" % (kernel_fw.shape[0], IN + H))
    if H not in LSTM_KERNEL_H:
        Hp = _padded_h(H)
        kf, bf = _pad_lstm_weights(kernel_fw, bias_fw, IN, H, Hp)
        kb, bb = _pad_lstm_weights(kernel_bw, bias_bw, IN, H, Hp) if ndir == 2 else (None, None)
        r = lstm_layer_fwd(x, seq_len, kf, bf, kb, bb, t_out=t_out, save=save, keep_prob=keep_prob, seed=seed)
        outp = r[0] if save else r
        out = torch.cat([outp[:, :, d * Hp:d * Hp + H] for d in range(ndir)], 2).contiguous()
        if save:
            r[1].kx_cat = None       # (the padded-width array is rebuilt per direction in the backward)
        return (out,) + tuple(r[1:]) if save else out
    t_out = T if t_out is None else t_out
    dev = x.device
    out = torch.empty((B, t_out, ndir * H), device=dev, dtype=torch.float32)
    gates = torch.empty((B, T, ndir, 4 * H), device=dev, dtype=torch.float32)
    act = torch.empty((B, T, ndir, H, 8), device=dev, dtype=torch.float32) if save else None
    hprev = torch.empty((B, T, ndir, H), device=dev, dtype=torch.float32) if save and not (p3 and p3.get("hprev") is not None) else None
    L = _lib.lib()
    nbytes = L.asr_lstm_ws_bytes(B, H, ndir)
    hx, nbytes = _hx_zeroed(dev, nbytes)
    if ndir == 2 and _KXCAT >= 1:       # input rows of the two kernels side by side: both directions' projection as ONE product (N = 8H)
        if kx_cat is None or bias_cat is None:      # (the encoder passes all its layers' concatenations, built by one launch)
            kx_cat = torch.cat([kernel_fw[:IN], kernel_bw[:IN]], 1)
            bias_cat = torch.cat([bias_fw, bias_bw])
    else:
        kx_cat = bias_cat = None
    st = _lstm_p3_struct(p3)
    rc = L.asr_lstm_layer_fwd_p3(_stream(), _p(x), B, T, IN, IN, _p(seq_len), H, ndir,
                                 _p(_f32(kernel_fw, "kernel_fw")), _p(_f32(bias_fw, "bias_fw")),
                                 _p(_f32(kernel_bw, "kernel_bw")), _p(_f32(bias_bw, "bias_bw")),
                                 _p(out), t_out, _p(gates), _p(act), _p(hprev), _p(hx), nbytes,
                                 _p(_Flag.get(dev)), float(keep_prob), int(seed) & 0xFFFFFFFF, _p(kx_cat), _p(bias_cat),
                                 C.byref(st) if st is not None else None)
    _check(rc, "asr_lstm_layer_fwd")
    gates.kx_cat = kx_cat        # rides along for lstm_layer_bwd(kx_cat=...): dX as one product over both directions
    return (out, gates, act, hprev) if save else out


def _hx(dev, nbytes):
    key = (dev, nbytes)
    if key not in _hx_cache:
        _hx_cache[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return _hx_cache[key]


WS_PREZEROED = 1 << 62          # include/e2e_asr_hip.h ASR_WS_PREZEROED
_arena = {}


_arena_need = {}                # {device: bytes the last step's workspaces asked for}: the next arena is sized from it


def ws_arena_begin(dev, nbytes=None):
    """One zero fill for all the exchange workspaces of a train step (Seq2SeqModel.step): the persistent launches then take
    slices of it (`_hx_zeroed`) instead of each running a memset launch in front of its kernel -- ~10 dependent launches of
    4-5 us per step.  Sized from what the previous step asked for (a few hundred KB at config 2; 24 MB the first time, and a
    step that outgrows its arena falls back to per-launch memsets and the next arena is larger).  ASR_WS_ARENA=0 turns it off."""
    if os.environ.get("ASR_WS_ARENA", "1") == "0" or dev.type != "cuda":
        _arena.pop(dev, None)
        return
    if nbytes is None:
        need = _arena_need.get(dev)
        nbytes = (24 << 20) if need is None else max(1 << 16, (need * 5 // 4 + 4095) // 4096 * 4096)
    _arena[dev] = [torch.zeros(nbytes, dtype=torch.uint8, device=dev), 0, 0]


def ws_arena_end(dev):
    a = _arena.pop(dev, None)
    if a is not None:
        _arena_need[dev] = a[2]


def _hx_zeroed(dev, nbytes):
    """(buffer, size argument): a slice of the step's zeroed arena + the PREZEROED flag, or the cached buffer the library zeroes.
    A slice is zero ONCE: it is handed out for one launch and never again within the step."""
    a = _arena.get(dev)
    if a is not None:
        a[2] += (nbytes + 255) // 256 * 256
        off = (a[1] + 255) // 256 * 256
        if off + nbytes <= a[0].numel():
            a[1] = off + nbytes
            return a[0][off:off + nbytes], nbytes | WS_PREZEROED
    return _hx(dev, nbytes), nbytes


def lstm_layer_bwd(x, seq_len, kernel_fw, kernel_bw, dout, gates, act, hprev, dk_fw, db_fw, dk_bw=None,
                   db_bw=None, need_dx=True, keep_prob=1.0, seed=0, join=True, kx_cat=None, p3=None):
    """Backward of lstm_layer_fwd.  `gates` is overwritten with dG; weight/bias gradients are
    ACCUMULATED into dk_*/db_* (views of the flat gradient buffer) on the library's side stream:
    join=False leaves them in flight (overlapping the next layer's BPTT) until ops.side_join().
    Returns dx [B,T,in] or None."""
    B, T, IN = x.shape
    H = kernel_fw.shape[1] // 4
    ndir = 1 if kernel_bw is None else 2
    dev = x.device
    if H not in LSTM_KERNEL_H:          # saved tensors are in the padded width (see lstm_layer_fwd)
        Hp = gates.shape[-1] // 4
        kf, _ = _pad_lstm_weights(kernel_fw, None, IN, H, Hp)
        kb = _pad_lstm_weights(kernel_bw, None, IN, H, Hp)[0] if ndir == 2 else None
        doutp = dout.new_zeros((B, dout.shape[1], ndir * Hp))
        for d in range(ndir):
            doutp[:, :, d * Hp:d * Hp + H] = dout[:, :, d * H:(d + 1) * H]
        gk = [kf.new_zeros(kf.shape) for _ in range(ndir)]
        gb = [kf.new_zeros(4 * Hp) for _ in range(ndir)]
        dx = lstm_layer_bwd(x, seq_len, kf, kb, doutp, gates, act, hprev, gk[0], gb[0], gk[1] if ndir == 2 else None,
                            gb[1] if ndir == 2 else None, need_dx=need_dx, keep_prob=keep_prob, seed=seed, join=True)
        for d, (dk, db) in enumerate(((dk_fw, db_fw), (dk_bw, db_bw))[:ndir]):
            g3 = gk[d].view(IN + Hp, 4, Hp)
            dk3 = dk.view(IN + H, 4, H)
            dk3[:IN] += g3[:IN, :, :H]
            dk3[IN:] += g3[IN:IN + H, :, :H]
            db.view(4, H).add_(gb[d].view(4, Hp)[:, :H])
        return dx
    dx = torch.empty_like(x) if need_dx else None
    L = _lib.lib()
    hxb, nbytes = _hx_zeroed(dev, L.asr_lstm_bwd_ws_bytes(B, H, ndir))
    st = _lstm_p3_struct(p3)
    rc = L.asr_lstm_layer_bwd_p3(_stream(), _p(x), B, T, IN, IN, _p(seq_len), H, ndir, _p(kernel_fw), _p(kernel_bw),
                                 _p(_f32(dout, "dout")), dout.shape[1], _p(gates), _p(act), _p(hprev), _p(dx),
                                 _p(dk_fw), _p(db_fw), _p(dk_bw), _p(db_bw), _p(hxb), nbytes,
                                 _p(_Flag.get(dev)), float(keep_prob), int(seed) & 0xFFFFFFFF,
                                 _p(kx_cat if (ndir == 2 and need_dx and _KXCAT >= 2) else None),
                                 C.byref(st) if st is not None else None)
    _check(rc, "asr_lstm_layer_bwd")
    keep_until_join(x, dout, gates, act, hprev, seq_len, kx_cat, p3, hxb)
    if join:          # weight/bias gradients are produced on the library's side stream
        side_join()
    return dx


def _ptr_array(ts):
    arr = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    return arr


def gru_layer_fwd(x, seq_len, cells, t_out=None, save=False, keep_prob=1.0, seed=0, h0=None, want_last=False):
    """One (Bi)GRU encoder layer (encoder.py:42-53 with use_lstm False).  cells: per direction (gates kernel [in+H, 2H], gates bias
    [2H], candidate kernel [in+H, H], candidate bias [H]).  Returns out [B, t_out, ndir*H] and, when save, (gx, cx, hprev, rh)."""
    _f32(x, "x"); _i32(seq_len, "seq_len")
    B, T, IN = x.shape
    ndir = len(cells)
    H = cells[0][2].shape[1]
    for wg, bg, wc, bc in cells:
        if wg.shape != (IN + H, 2 * H) or wc.shape != (IN + H, H) or bg.numel() != 2 * H or bc.numel() != H:
            raise ValueError("gru cell shapes do not match in=%d, H=%d" % (IN, H))
        for t_ in (wg, bg, wc, bc):
            _f32(t_, "gru weight")
    t_out = T if t_out is None else t_out
    dev = x.device
    f = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)
    out, gx, cx = f(B, t_out, ndir * H), f(B, T, ndir, 2 * H), f(B, T, ndir, H)
    hprev, rh = (f(B, T, ndir, H), f(B, T, ndir, H)) if save else (None, None)
    a = [_ptr_array([c[i] for c in cells]) for i in range(4)]
    h_last = f(B, ndir, H) if want_last else None
    rc = _lib.lib().asr_gru_layer_fwd(_stream(), _p(x), B, T, IN, IN, _p(seq_len), H, ndir, a[0], a[1], a[2], a[3],
                                      _p(out), t_out, _p(gx), _p(cx), _p(hprev), _p(rh), float(keep_prob), int(seed) & 0xFFFFFFFF,
                                      _p(_f32(h0, "h0")), _p(h_last))
    _check(rc, "asr_gru_layer_fwd")
    r = (out, gx, cx, hprev, rh) if save else out
    if want_last:
        return (r + (h_last,)) if save else (r, h_last)
    return r


def gru_layer_bwd(x, seq_len, cells, dout, gx, cx, hprev, rh, grads, need_dx=True, keep_prob=1.0, seed=0, dh_last=None,
                  want_dh0=False):
    """Backward of gru_layer_fwd.  grads: per direction (d gates kernel, d gates bias, d candidate kernel, d candidate bias) views
    of the flat gradient buffer (accumulated into).  Returns dx [B, T, in] or None."""
    B, T, IN = x.shape
    ndir = len(cells)
    H = cells[0][2].shape[1]
    dev = x.device
    dx = torch.empty_like(x) if need_dx else None
    wt = torch.empty(ndir * 3 * H * H, device=dev, dtype=torch.float32)
    dh0 = torch.empty((B, ndir, H), device=dev, dtype=torch.float32) if want_dh0 else None
    w = [_ptr_array([c[i] for c in cells]) for i in (0, 2)]
    g = [_ptr_array([c[i] for c in grads]) for i in range(4)]
    rc = _lib.lib().asr_gru_layer_bwd(_stream(), _p(x), B, T, IN, IN, _p(seq_len), H, ndir, w[0], w[1], _p(_f32(dout, "dout")),
                                      dout.shape[1], _p(gx), _p(cx), _p(hprev), _p(rh), _p(wt), g[0], g[1], g[2], g[3], _p(dx),
                                      float(keep_prob), int(seed) & 0xFFFFFFFF, _p(_f32(dh_last, "dh_last")), _p(dh0))
    _check(rc, "asr_gru_layer_bwd")
    return (dx, dh0) if want_dh0 else dx


def linear_wt(x, wt, out=None, accumulate=False, n=None, k=None, ldw=None):
    """out[M,N] (+)= x[M,K] @ wt[N,K]^T -- data-gradient products (wt = rows of a TF kernel)."""
    M = x.shape[0]
    K = x.shape[1] if k is None else k
    N = wt.shape[0] if n is None else n
    ldw = wt.shape[1] if ldw is None else ldw
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=torch.float32)
    rc = _lib.lib().asr_linear_wt_fwd(_stream(), _p(x), x.shape[1], K, _p(wt), ldw, _p(out), out.shape[1], M, N,
                                      int(accumulate))
    _check(rc, "asr_linear_wt_fwd")
    return out


def gemm_batched(a, b, out, M, N, K, lda, ldb, ldc, stride_a, stride_b, stride_c, batch, trans_a=False, trans_b=False,
                 accumulate=False):
    """out_i[M,N] (+)= op(a_i) @ op(b_i) for `batch` problems whose operands sit strides (in elements) apart."""
    rc = _lib.lib().asr_gemm_f32_batched(_stream(), int(trans_a), int(trans_b), M, N, K, _p(a), lda, stride_a, _p(b), ldb, stride_b,
                                         _p(out), ldc, stride_c, None, int(accumulate), batch)
    _check(rc, "asr_gemm_f32_batched")
    return out


def zero_finished_rows(logits, seq_len, T, B):
    """raw_rnn emits zeros for finished rows (attn_decoder.py:170): rows (t, b) with t >= seq_len[b] of logits [(T*B), V]."""
    _check(_lib.lib().asr_zero_finished_rows(_stream(), _p(logits), _p(_i32(seq_len, "seq_len")), T, B, logits.shape[1]),
           "asr_zero_finished_rows")
    return logits


def lstm_cell_bwd(gates, c, c_prev, dout, dh_carry, dc_carry, keep_prob=1.0, seed=0, step=0):
    """Pointwise backward of one lstm_cell step (see include/e2e_asr_hip.h asr_lstm_cell_bwd).  dout / dh_carry may be
    column slices of wider row-major buffers (their row stride is passed on); gates: activated in, dG out."""
    B, H = c.shape
    rc = _lib.lib().asr_lstm_cell_bwd(_stream(), _p(gates), _p(c), _p(c_prev), _p(dout), dout.stride(0),
                                      _p(dh_carry), 0 if dh_carry is None else dh_carry.stride(0), _p(dc_carry), B, H,
                                      float(keep_prob), int(seed) & 0xFFFFFFFF, int(step))
    _check(rc, "asr_lstm_cell_bwd")
    return gates


def attn_cell_bwd(q, w_att, b_att, v, hf, enc, enc_len, alpha, dqc, dctx_carry, dhf, dctx_out, dy, dv_part, gates, c_prev,
                  dh_carry, dc_carry):
    """One step of the attention + query-cell backward (include/e2e_asr_hip.h asr_attn_cell_bwd)."""
    B, Te, D = enc.shape
    H, A = w_att.shape
    rc = _lib.lib().asr_attn_cell_bwd(_stream(), _p(q), _p(w_att), _p(b_att), _p(v), _p(hf), _p(enc), _p(enc_len), _p(alpha), None,
                                      _p(dqc), _p(dctx_carry), 0 if dctx_carry is None else dctx_carry.stride(0), _p(dhf),
                                      _p(dctx_out), _p(dy), _p(dv_part), _p(gates), _p(c_prev), _p(dh_carry),
                                      0 if dh_carry is None else dh_carry.stride(0), _p(dc_carry), B, Te, H, A, D)
    _check(rc, "asr_attn_cell_bwd")


def attn_bwd(q, w_att, b_att, v, hf, enc, enc_len, alpha, dqc, dctx_carry, dhf, dctx_out, dy, dv_part):
    """Attention backward alone (include/e2e_asr_hip.h asr_attn_bwd): returns dq [B, H], the gradient w.r.t. the query."""
    B, Te, D = enc.shape
    H, A = w_att.shape
    dq = torch.empty((B, H), device=enc.device, dtype=torch.float32)
    rc = _lib.lib().asr_attn_bwd(_stream(), _p(q), _p(w_att), _p(b_att), _p(v), _p(hf), _p(enc), _p(enc_len), _p(alpha), _p(dqc),
                                 _p(dctx_carry), 0 if dctx_carry is None else dctx_carry.stride(0), _p(dhf), _p(dctx_out), _p(dy),
                                 _p(dv_part), _p(dq), B, Te, H, A, D)
    _check(rc, "asr_attn_bwd")
    return dq


def colsum(x, out, accumulate=True):
    M, N = x.shape
    _check(_lib.lib().asr_colsum_f32(_stream(), _p(x), x.stride(0), M, N, _p(out), int(accumulate)), "asr_colsum_f32")
    return out


def gather_rows(table, idx):
    out = torch.empty((idx.numel(), table.shape[1]), device=table.device, dtype=torch.float32)
    _check(_lib.lib().asr_gather_rows(_stream(), _p(table), _p(idx), _p(out), idx.numel(), table.shape[1]), "asr_gather_rows")
    return out


def scatter_add_rows(table_grad, idx, g):
    """table_grad[idx[r]] += g[r] (embedding gradient).  Deterministic wgrad mode: a row's occurrences in ascending order, no atomics."""
    L = _lib.lib()
    if L.asr_get_wgrad_mode() and table_grad.shape[1] <= 1024:
        _check(L.asr_scatter_add_rows_ordered(_stream(), _p(table_grad), table_grad.shape[0], _p(idx), _p(g), idx.numel(),
                                              table_grad.shape[1], g.shape[1]), "asr_scatter_add_rows_ordered")
        return
    _check(L.asr_scatter_add_rows(_stream(), _p(table_grad), _p(idx), _p(g), idx.numel(), table_grad.shape[1]),
           "asr_scatter_add_rows")


_sumsq_ws = {}


def sumsq(x, out=None):
    dev = x.device
    if dev not in _sumsq_ws:
        _sumsq_ws[dev] = torch.empty(1024, device=dev, dtype=torch.float32)
    if out is None:
        out = torch.empty(1, device=dev, dtype=torch.float32)
    _check(_lib.lib().asr_sumsq_f32(_stream(), _p(x), x.numel(), _p(_sumsq_ws[dev]), _p(out)), "asr_sumsq_f32")
    return out


def clip_adam(p, m, v, g, sumsq_t, grad_scale, clip_norm, lr_t, beta1=0.9, beta2=0.999, eps=1e-8):
    _check(_lib.lib().asr_clip_adam_f32(_stream(), _p(p), _p(m), _p(v), _p(g), p.numel(), _p(sumsq_t),
                                        float(grad_scale), float(clip_norm), float(lr_t), float(beta1),
                                        float(beta2), float(eps)), "asr_clip_adam_f32")


def linear(x1, w, bias=None, x2=None, gather=None, zero_from=None, zero_t=0, out=None):
    """[x1 | x2] @ w + bias for a skinny batch (attn_decoder.py `_linear` call sites)."""
    _f32(x1, "x1"); _f32(w, "w")
    M = x1.shape[0] if gather is None else gather.shape[0]
    K1 = x1.shape[1]
    K2 = 0 if x2 is None else x2.shape[1]
    N = w.shape[1]
    if w.shape[0] != K1 + K2:
        raise ValueError("linear: weight rows %d != %d" % (w.shape[0], K1 + K2))
    if out is None:
        out = torch.empty((M, N), device=x1.device, dtype=torch.float32)
    rc = _lib.lib().asr_linear_fwd(_stream(), _p(x1), x1.shape[1], K1, _p(gather), _p(_f32(x2, "x2")),
                                   K2, K2, _p(w), N, _p(_f32(bias, "bias")), _p(out), out.shape[1], M, N,
                                   _p(zero_from), int(zero_t))
    _check(rc, "asr_linear_fwd")
    return out


def lstm_cell(x, h_prev, c_prev, kernel, bias, gather=None, keep_prob=1.0, seed=0, step=0,
              save_gates=False):
    """One BasicLSTMCell step (basic_lstm.py:14-23) -> (c, h[, h_dropped][, gates])."""
    _f32(x, "x"); _f32(h_prev, "h_prev"); _f32(kernel, "kernel"); _f32(bias, "bias")
    H = kernel.shape[1] // 4
    M = x.shape[0] if gather is None else gather.shape[0]
    dev = x.device
    c = torch.empty((M, H), device=dev, dtype=torch.float32)
    h = torch.empty_like(c)
    hd = torch.empty_like(c) if keep_prob < 1.0 else None
    g = torch.empty((M, 4 * H), device=dev, dtype=torch.float32) if save_gates else None
    rc = _lib.lib().asr_lstm_cell_fwd(_stream(), _p(x), x.shape[1], x.shape[1], _p(gather), _p(h_prev),
                                      _p(_f32(c_prev, "c_prev")), _p(kernel), _p(bias), H, M, _p(c), _p(h),
                                      _p(hd), _p(g), float(keep_prob), int(seed) & 0xFFFFFFFF, int(step))
    _check(rc, "asr_lstm_cell_fwd")
    res = (c, h)
    if hd is not None:
        res += (hd,)
    if g is not None:
        res += (g,)
    return res


def attention(q, w_att, b_att, v, hf, enc, enc_len, shared=False):
    """Fused Bahdanau attention (attn_decoder.py:77-93) -> (ctx [B,D], alpha [B,Te]).
    shared=True: hf [Te,A] / enc [Te,D] / enc_len[0] describe one utterance attended by every
    query row (the hypotheses of a beam, beam_search.py:137-161)."""
    for n, t in (("q", q), ("w_att", w_att), ("b_att", b_att), ("v", v), ("hf", hf), ("enc", enc)):
        _f32(t, n)
    _i32(enc_len, "enc_len")
    if shared:
        B = q.shape[0]
        Te, D = enc.shape[-2], enc.shape[-1]
        H, A = w_att.shape
        alpha = torch.empty((B, Te), device=q.device, dtype=torch.float32)
        ctx = torch.empty((B, D), device=q.device, dtype=torch.float32)
        rc = _lib.lib().asr_attention_shared_fwd(_stream(), _p(q), q.shape[1], _p(w_att), _p(b_att), _p(v), _p(hf),
                                                 _p(enc), _p(enc_len), _p(alpha), _p(ctx), B, Te, H, A, D, 1)
        _check(rc, "asr_attention_shared_fwd")
        return ctx, alpha
    B, Te, D = enc.shape
    H, A = w_att.shape
    alpha = torch.empty((B, Te), device=q.device, dtype=torch.float32)
    ctx = torch.empty((B, D), device=q.device, dtype=torch.float32)
    rc = _lib.lib().asr_attention_fwd(_stream(), _p(q), q.shape[1], _p(w_att), _p(b_att), _p(v), _p(hf),
                                      _p(enc), _p(enc_len), _p(alpha), _p(ctx), B, Te, H, A, D)
    _check(rc, "asr_attention_fwd")
    return ctx, alpha


def masked_ce(logits, targets, seq_len):
    """losses.py:7-35 forward -> (loss scalar tensor, lse workspace)."""
    _f32(logits, "logits"); _i32(targets, "targets"); _i32(seq_len, "seq_len")
    T, B = targets.shape
    V = logits.shape[1]
    dev = logits.device
    nll = torch.empty(T * B, device=dev, dtype=torch.float32)
    lse = torch.empty(T * B, device=dev, dtype=torch.float32)
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    rc = _lib.lib().asr_masked_ce_fwd(_stream(), _p(logits), _p(targets), _p(seq_len), _p(nll), _p(lse),
                                      _p(loss), T, B, V)
    _check(rc, "asr_masked_ce_fwd")
    return loss, lse


def masked_ce_fwd_bwd(logits, targets, seq_len, grad_scale):
    """losses.py:7-35 forward AND its gradient in one pass over the logits -> (loss, lse, dlogits)."""
    _f32(logits, "logits"); _i32(targets, "targets"); _i32(seq_len, "seq_len"); _f32(grad_scale, "grad_scale")
    T, B = targets.shape
    V = logits.shape[1]
    dev = logits.device
    nll = torch.empty(T * B, device=dev, dtype=torch.float32)
    lse = torch.empty(T * B, device=dev, dtype=torch.float32)
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    d = torch.empty_like(logits)
    _check(_lib.lib().asr_masked_ce_fwd_bwd(_stream(), _p(logits), _p(targets), _p(seq_len), _p(grad_scale), _p(nll), _p(lse),
                                            _p(loss), _p(d), T, B, V), "asr_masked_ce_fwd_bwd")
    return loss, lse, d


def masked_ce_bwd(logits, targets, lse, seq_len, grad_scale):
    T, B = targets.shape
    V = logits.shape[1]
    d = torch.empty_like(logits)
    rc = _lib.lib().asr_masked_ce_bwd(_stream(), _p(logits), _p(targets), _p(lse), _p(seq_len),
                                      _p(_f32(grad_scale, "grad_scale")), _p(d), T, B, V)
    _check(rc, "asr_masked_ce_bwd")
    return d


def next_token(logits, sample=False, seed=0, step=0):
    B, V = logits.shape
    tok = torch.empty(B, device=logits.device, dtype=torch.int32)
    rc = _lib.lib().asr_next_token(_stream(), _p(_f32(logits, "logits")), B, V, V, _p(tok), int(sample),
                                   int(seed) & 0xFFFFFFFF, int(step))
    _check(rc, "asr_next_token")
    return tok


DEC_WEIGHT_LEAVES = {   # struct field -> variable leaf under model/rnn_decoder_<task>/
    "embedding": "decoder/embedding", "attn_enc_w": "AttnW", "attn_v": "AttnV",
    "attn_w": "rnn/Attention/kernel", "attn_b": "rnn/Attention/bias",
    "lm_kernel": "rnn/basic_lstm_cell/kernel", "lm_bias": "rnn/basic_lstm_cell/bias",
    "dec_kernel": "rnn/basic_lstm_cell_1/kernel", "dec_bias": "rnn/basic_lstm_cell_1/bias",
    "inp_w": "rnn/InputProjection/kernel", "inp_b": "rnn/InputProjection/bias",
    "ap_w": "rnn/AttnProjection/kernel", "ap_b": "rnn/AttnProjection/bias",
    "out_w": "rnn/OutputProjection/kernel", "out_b": "rnn/OutputProjection/bias",
    "simple_w": "rnn/SimpleProjection/kernel", "simple_b": "rnn/SimpleProjection/bias",
}


def _dec_struct(cls, tensors):
    s = cls()
    for f, ft in cls._fields_:
        if ft is not _lib.vp:
            continue                       # scalar members are set by the caller
        t = tensors.get(f)
        setattr(s, f, None if t is None else t.data_ptr())
    return s


class _decoder_precision(object):
    """bf16 mode (BASELINE config 3): the decoder's products (K <= 1024, V = 1000; 9 % of the model's GEMM FLOPs) run with TWO bf16
    operand planes while the encoder's stay on one.  Measured at the full config-2 size against the float64 oracle
    (scripts/exp_prec_map.py): one plane everywhere 1.22e-3 max logit error, one plane in the encoder only 0.82e-3, in the
    decoder only 0.89e-3 -- each half carries about as much as the north star's 1e-3 allows, so the cheap half is doubled.
    `set_decoder_bf16_planes(1)` restores one plane everywhere."""
    planes = 2

    def __enter__(self):
        self.prev = get_gemm_precision()
        if self.prev == "bf16" and _decoder_precision.planes == 2:
            set_gemm_precision("bf16x2")
        return self

    def __exit__(self, *exc):
        if get_gemm_precision() != self.prev:
            set_gemm_precision(self.prev)
        return False


def set_decoder_bf16_planes(n):
    if int(n) not in (1, 2):
        raise ValueError("decoder bf16 planes: 1 or 2")
    _decoder_precision.planes = int(n)


_const_cache = {}


def _const(kind, dev, n, value=0):
    """Read-only device constants the decoder entry points take (an all-zero initial state, a lengths vector of one value): built
    once per (device, size, value) instead of by a fill kernel on the launch stream of every step (two launches + their gaps
    between the encoder's last kernel and the decoder's first, round 5).  NEVER written by the library."""
    key = (kind, dev, int(n), int(value))
    t = _const_cache.get(key)
    if t is None:
        t = (torch.zeros(n, device=dev, dtype=torch.float32) if kind == "zeros" else
             torch.full((n,), int(value), device=dev, dtype=torch.int32))
        _const_cache[key] = t
    return t


def attn_decoder_fwd(wt, dec_inp, seq_len, enc, enc_len, mode=0, coin=None, samp_prob=0.0,
                     keep_lm=1.0, seed=0, t_out=None):
    """Whole attention decoder forward (attn_decoder.py:37-172).

    wt: dict struct-field -> float32 CUDA tensor (see DEC_WEIGHT_LEAVES).
    dec_inp int32 [T_dec,B]; seq_len int32 [B] (device) ; enc [B,Te,D]; enc_len int32 [B].
    Returns logits [(T_out*B),V] and the activation dict consumed by the backward.
    """
    B, Te, D = enc.shape
    V, E = wt["embedding"].shape
    H = wt["dec_kernel"].shape[1] // 4
    lmH = wt["lm_kernel"].shape[1] // 4
    A = wt["attn_w"].shape[1]
    if t_out is None:
        t_out = int(seq_len.max().item())
    T = t_out
    dev = enc.device
    f = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)
    ws = dict(hf=f(B, Te, A), tok=dec_inp[:T].contiguous().clone(), lm_gates=f(T, B, 4 * lmH),
              lm_c=f(T, B, lmH), lm_h=f(T, B, lmH), lm_hd=f(T, B, lmH) if keep_lm < 1.0 else None,
              sp=f(T, B, H) if wt.get("simple_w") is not None else None, x=f(T, B, E),
              dec_gates=f(T, B, 4 * H), dec_c=f(T, B, H), dec_h=f(T, B, H), alpha=f(T, B, Te),
              ctx=f(T, B, D), p=f(T, B, H),
              zeros=_const("zeros", dev, B * max(H, lmH, D)), y=f(T, B, A))
    L = _lib.lib()
    if mode != 1 and L.asr_decoder_chain_supported(B, Te, D, A, H):       # persistent decoder-chain path
        P_ = H if wt.get("simple_w") is not None else lmH
        ws["w2k"] = f(P_ + D + 1, 4 * H)          # W_inp.K_x [(P+D),4H] followed by the composed bias [4H]
        ws["chain_ws"] = _hx(dev, L.asr_decoder_chain_ws_bytes(B, D, A, H))
        ws["err"] = _Flag.get(dev)
        if L.asr_decoder_lm_chain_supported(B, lmH):                     # persistent LM cell chain
            ws["lm_act"] = f(T, B, lmH, 8)
            ws["lm_hprev"] = f(T, B, lmH)
            ws["lm_state"] = f(2, 2, B, lmH)
            ws["lm_len"] = _const("full_i32", dev, B, T)
            ws["lm_hx"] = _hx(dev, L.asr_lstm_ws_bytes(B, lmH, 1))
    if mode == 1 and wt.get("simple_w") is None and keep_lm >= 1.0 and L.asr_decoder_greedy_supported(B, Te, D, A, H, lmH, E, V):
        ws["w2k"] = f(lmH + D + 1, 4 * H)         # inference graph: the whole greedy loop in one persistent launch
        ws["greedy_ws"] = _hx(dev, L.asr_decoder_greedy_ws_bytes(B, D, A, H, lmH, V))
        ws["err"] = _Flag.get(dev)
    if (mode != 1 and T <= 256 and wt.get("simple_w") is None and ws.get("lm_act") is not None and
            os.environ.get("ASR_DEC_TRAINK", "1") != "0" and L.asr_decoder_greedy_supported(B, Te, D, A, H, lmH, E, V)):
        # training graph in one persistent launch (TRAIN instantiation of the same kernel): needs the chain path's buffers too
        ws["greedy_ws"] = _hx(dev, L.asr_decoder_greedy_ws_bytes(B, D, A, H, lmH, V))
    logits = f(T * B, V)
    cw = _dec_struct(_lib.DecWeights, wt)
    cd = _lib.DecDims(B, Te, D, A, H, lmH, E, V, T)
    cws = _dec_struct(_lib.DecWs, ws)
    coin_arr = None
    if coin is not None:
        coin_arr = (C.c_float * len(coin))(*[float(c) for c in coin])
    with _decoder_precision():
        rc = _lib.lib().asr_attn_decoder_fwd(_stream(), C.byref(cw), C.byref(cd), C.byref(cws), _p(enc),
                                             _p(_i32(enc_len, "enc_len")), _p(_i32(seq_len, "seq_len")),
                                             int(mode), coin_arr, float(samp_prob), float(keep_lm),
                                             int(seed) & 0xFFFFFFFF, _p(logits))
    _check(rc, "asr_attn_decoder_fwd")
    ws["_dims"] = (B, Te, D, A, H, lmH, E, V, T)
    return logits, ws


def attn_decoder_bwd(wt, gt, ws, enc, enc_len, dlogits, denc, keep_lm=1.0, seed=0, defer_lm=False, side_busy=False):
    """Backward of attn_decoder_fwd.  wt/gt: weight and gradient tensors by struct field (gradients
    are accumulated into gt, which are views of the flat gradient buffer); denc [B,Te,D] is
    accumulated into.  defer_lm: leave the LM cell chain's backward (its persistent BPTT, the embedding / LM-cell gradients)
    to a later attn_decoder_bwd_lm(returned dict) -- only honoured on the persistent LM-chain path.  side_busy: an earlier
    decoder's weight gradients are still queued on the side stream (second task of a multitask step)."""
    B, Te, D, A, H, lmH, E, V, T = ws["_dims"]
    dev = enc.device
    f = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)
    P = H if wt.get("simple_w") is not None else lmH
    bw = dict(dP=f(T, B, H), dQC=f(T, B, H + D), dY=f(T, B, A), dXH=f(T, B, E + H), dLC=f(T, B, P + D),
              dlm=f(T, B, lmH) if wt.get("simple_w") is not None else None, dEH=f(T, B, E + lmH),
              dc_dec=f(B, H), dc_lm=f(B, lmH), dhf=f(B, Te, A), dv_part=f(16 * B, A),
              dctx=f(T, B, D), emb_all=f(T, B, E))
    L = _lib.lib()
    if ws.get("err") is not None and L.asr_decoder_chain_supported(B, Te, D, A, H):   # persistent backward chain
        bw["chain_ws"] = _hx(dev, L.asr_decoder_chain_bwd_ws_bytes(B, D, A, H))
        bw["wc"] = f(D, 4 * H)
    if ws.get("lm_act") is not None:
        bw["lm_hx"] = _hx(dev, L.asr_lstm_bwd_ws_bytes(B, lmH, 1))
    cw = _dec_struct(_lib.DecWeights, wt)
    cg = _dec_struct(_lib.DecWeights, gt)
    cd = _lib.DecDims(B, Te, D, A, H, lmH, E, V, T)
    cws = _dec_struct(_lib.DecWs, {k: v for k, v in ws.items() if k != "_dims"})
    cbw = _dec_struct(_lib.DecBwdWs, bw)
    deferred = bool(defer_lm) and ws.get("lm_act") is not None and bw.get("lm_hx") is not None
    cbw.lm_deferred = int(deferred)
    cbw.side_busy = int(bool(side_busy))
    with _decoder_precision():
        rc = _lib.lib().asr_attn_decoder_bwd(_stream(), C.byref(cw), C.byref(cg), C.byref(cd), C.byref(cws),
                                             C.byref(cbw), _p(enc), _p(enc_len), _p(_f32(dlogits, "dlogits")),
                                             _p(_f32(denc, "denc")), float(keep_lm), int(seed) & 0xFFFFFFFF)
    _check(rc, "asr_attn_decoder_bwd")
    keep_until_join(bw, ws, wt, gt, enc, enc_len, dlogits)
    if deferred:
        bw["_lm_tail"] = (cw, cg, cd, cws, cbw, float(keep_lm), int(seed) & 0xFFFFFFFF)
    return bw


def attn_decoder_bwd_lm(bw):
    """The deferred LM-chain part of attn_decoder_bwd(defer_lm=True), on the current stream (asr_attn_decoder_bwd_lm): call it
    behind the encoder's backward pass and before side_join().  No-op when nothing was deferred."""
    t = bw.pop("_lm_tail", None)
    if t is None:
        return
    cw, cg, cd, cws, cbw, keep_lm, seed = t
    with _decoder_precision():
        rc = _lib.lib().asr_attn_decoder_bwd_lm(_stream(), C.byref(cw), C.byref(cg), C.byref(cd), C.byref(cws), C.byref(cbw),
                                                keep_lm, seed)
    _check(rc, "asr_attn_decoder_bwd_lm")


def pyramid_reduce(x, seq_len=None, skip=2):
    """encoder.py:94-119 as a standalone kernel: x [B,T,F] -> ([B,ceil(T/skip),skip*F], ceil(len/skip)); frames past T
    are zeros.  (Encoder itself never copies: its layer outputs are already in this layout.)"""
    _f32(x, "x")
    B, T, F = x.shape
    Tp = (T + skip - 1) // skip
    y = torch.empty((B, Tp, skip * F), device=x.device, dtype=torch.float32)
    lo = None
    if seq_len is not None:
        _i32(seq_len, "seq_len")
        lo = torch.empty_like(seq_len)
    _check(_lib.lib().asr_pyramid_reduce_fwd(_stream(), _p(x), _p(seq_len), _p(y), _p(lo), B, T, F, skip), "asr_pyramid_reduce_fwd")
    return (y, lo) if seq_len is not None else y


def pyramid_reduce_bwd(dy, T, skip=2):
    """Gradient of pyramid_reduce w.r.t. x: dy [B,Tp,skip*F] -> dx [B,T,F]."""
    _f32(dy, "dy")
    B, Tp, SF = dy.shape
    F = SF // skip
    dx = torch.empty((B, T, F), device=dy.device, dtype=torch.float32)
    _check(_lib.lib().asr_pyramid_reduce_bwd(_stream(), _p(dy), _p(dx), B, T, F, skip), "asr_pyramid_reduce_bwd")
    return dx


def set_gemm_precision(dtype):
    """"f32" (default, exact) or "bf16": bf16 MFMA operands with fp32 accumulation for the whole-tile GEMMs
    (BASELINE config 3).  Process-wide."""
    mode = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "bf16x2": 2}[str(dtype)]
    _check(_lib.lib().asr_set_gemm_precision(mode), "asr_set_gemm_precision")


def set_gemm_split(on):
    """fp32 products of whole tiles on the bf16 matrix pipe by exact 3-way operand splitting (default on) or on
    v_mfma_f32_32x32x2_f32 (off).  Both are fp32-accurate; see include/e2e_asr_hip.h.  Process-wide."""
    _check(_lib.lib().asr_set_gemm_split(int(bool(on))), "asr_set_gemm_split")


def get_gemm_split():
    return bool(_lib.lib().asr_get_gemm_split())


def get_gemm_precision():
    return ("f32", "bf16", "bf16x2")[_lib.lib().asr_get_gemm_precision()]


def set_wgrad_mode(slabs=True):
    """True: the deterministic mode -- split-K weight gradients through slabs + a fixed-order reduce, two-stage bias sums, ordered
    embedding scatter: every gradient bit-reproducible run to run (+0.1 ... 0.3 ms per config-2 step); False (default): float atomics."""
    _check(_lib.lib().asr_set_wgrad_mode(int(bool(slabs))), "asr_set_wgrad_mode")


def get_wgrad_mode():
    return bool(_lib.lib().asr_get_wgrad_mode())


PROF_TAGS = {"lstm_rec_fwd": 0, "lstm_rec_bwd": 1, "gemm": 2, "decoder_fwd": 3, "decoder_bwd": 4, "optim": 5, "side_tail": 6}


def prof_enable(on=True, only=None):
    """HIP-event timing of the persistent kernel families (PROF_TAGS).  only: names of the families to record (each event pair
    costs the stream a few microseconds: a timed run records the family it reports, bench.py)."""
    if on and only is not None:
        mask = 0
        for name in only:
            mask |= 1 << PROF_TAGS[name]
        _lib.lib().asr_prof_enable_mask(mask)
    else:
        _lib.lib().asr_prof_enable(int(bool(on)))


def prof_read(tag):
    """(total_ms, launches) of a profiled kernel family since prof_enable (synchronises)."""
    ms, n = C.c_double(0), C.c_int(0)
    _check(_lib.lib().asr_prof_read(PROF_TAGS[tag], C.byref(ms), C.byref(n)), "asr_prof_read")
    return ms.value, n.value


def prof_read_each(tag, cap=4096):
    """[elapsed ms] per recorded occurrence of a profiled family, in order (synchronises)."""
    buf, n = (C.c_double * cap)(), C.c_int(0)
    _check(_lib.lib().asr_prof_read_each(PROF_TAGS[tag], buf, cap, C.byref(n)), "asr_prof_read_each")
    return [buf[i] for i in range(n.value)]


_keepalive = []      # tensors still read by side-stream kernels: PyTorch's allocator only tracks the
                     # current stream, so they must not be freed (and re-used) before the join


def keep_until_join(*objs):
    _keepalive.append(objs)


def set_lstm_mfma(on):
    """bf16 mode only: recurrent products of the first-version recurrent kernels on the bf16 matrix pipe (off by default
    since round 3: the fp32 version-2 recurrences are faster and exact)."""
    _check(_lib.lib().asr_set_lstm_mfma(int(bool(on))), "asr_set_lstm_mfma")


def side_wait(stream):
    """`stream` (a torch.cuda.Stream other than the current one) waits for the side-stream work queued so far; the pending
    join is kept for the current stream's side_join() (parallel.py's tail overlap)."""
    _check(_lib.lib().asr_side_wait(C.c_void_p(stream.cuda_stream)), "asr_side_wait")


def side_join():
    """Order the library's side-stream work (weight/bias and LM-chain gradients) before the current
    stream; after it, buffers those kernels read may be released."""
    _check(_lib.lib().asr_side_join(_stream()), "asr_side_join")
    _keepalive.clear()
