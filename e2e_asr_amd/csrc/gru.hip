// GRU encoder layer (tf.nn.rnn_cell.GRUCell under dynamic_rnn / bidirectional_dynamic_rnn: encoder.py:42-53 with use_lstm False --
// the `Encoder.class_params()` default, encoder.py:27; the reference CLI always overrides it, encoder.py:187, so this cell is OFF
// the measured hot path).  Built for completeness of `Encoder.get_cell`, not for speed: no inter-workgroup exchange, no
// persistent-kernel machinery.
//   gate_inputs = [x, h] . W_gates + b_gates;  r, u = split(sigmoid(gate_inputs));  c = tanh([x, r*h] . W_cand + b_cand)
//   h' = u * h + (1 - u) * c;  t >= len: output 0, state copied through;  bw direction walks t = len-1 .. 0.
// Decomposition: the x parts of both products for all time steps are two GEMMs in front (the existing fp32-accurate MFMA
// kernels); the recurrent part is ONE workgroup per (utterance, direction) that keeps h in LDS and streams the recurrent
// weights W_gates[in:] [H, 2H] and W_cand[in:] [H, H] from L2 every step (768 KB per step at H = 256, shared by all workgroups
// of a direction: ~5 us per step), coalesced over the output unit.  The backward pass mirrors it with the transposed weights
// (one tiled transpose per call) and leaves dG_gates / dG_cand for the weight-gradient and dX GEMMs.
#include "common.h"
#include <algorithm>

extern "C" int asr_gemm_f32(void* stream, int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                            float* C, int ldc, const float* bias, int accumulate);
extern "C" int asr_colsum_f32(void* stream, const float* x, int ldx, int M, int N, float* out, int accumulate);

namespace asr {

struct GruArgs {
    const float* wgh; const float* wch;      // forward: W_gates[in:] [H][2H], W_cand[in:] [H][H];  backward: their transposes [2H][H], [H][H]
    float* gx;         // [B][T][ND][2H]  in: x.W_gates[:in] + b;  out (fwd): activated (r | u);  out (bwd): dG_gates
    float* cx;         // [B][T][ND][H]   in: x.W_cand[:in] + b;   out (fwd): c = tanh(.);       out (bwd): dG_cand
    float* hprev;      // [B][T][ND][H]   h_{t-1} (saved by the forward)
    float* rh;         // [B][T][ND][H]   r * h_{t-1} (saved by the forward: the A operand of dW_cand[in:])
    float* out;        // forward: [B][Tout][ND*H]
    const float* dout; // backward: [B][Tout][ND*H]
    const float* h0;   // forward: initial state [B][ND][H] or NULL (zeros);  backward: dh of the state BEHIND the last step, or NULL
    float* h_last;     // forward: final (plain) state [B][ND][H] or NULL;    backward: dh of the initial state [B][ND][H] or NULL
    const int* len;
    int B, T, Tout, ND, H, dir;      // dir: the direction this launch runs (arrays keep both)
    float keep; uint32_t seed;
};

// one workgroup per utterance b of direction a.dir; thread j owns units j, j + NT, ...
template <int NT, int UPT>
__global__ __launch_bounds__(NT) void gru_rec_fwd_kernel(GruArgs a) {
    extern __shared__ float sm[];
    const int H = a.H, H2 = 2 * H;
    float* hs = sm;            // h_{t-1} [H]
    float* rhs = sm + H;       // r * h_{t-1} [H]
    const int b = blockIdx.x, dir = a.dir;
    const int tid = threadIdx.x;
    const int S = min(max(a.len[b], 0), a.T);
    for (int j = tid; j < H; j += NT) hs[j] = a.h0 ? a.h0[((size_t)b * a.ND + dir) * H + j] : 0.f;
    __syncthreads();
    for (int s = 0; s < S; ++s) {
        const int t = dir ? (S - 1 - s) : s;
        const size_t row = ((size_t)b * a.T + t) * a.ND + dir;
        float* gxr = a.gx + row * H2;
        float* cxr = a.cx + row * H;
        float r_[UPT], u_[UPT];
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int j = tid + q * NT;
            if (j < H) {
                float ar = gxr[j], au = gxr[H + j];
                const float* w = a.wgh + j;
#pragma unroll 8
                for (int k = 0; k < H; ++k) { const float hk = hs[k]; ar = fmaf(hk, w[(size_t)k * H2], ar); au = fmaf(hk, w[(size_t)k * H2 + H], au); }
                r_[q] = fast_sigmoid(ar); u_[q] = fast_sigmoid(au);
                rhs[j] = r_[q] * hs[j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int j = tid + q * NT;
            if (j < H) {
                float ac = cxr[j];
                const float* w = a.wch + j;
#pragma unroll 8
                for (int k = 0; k < H; ++k) ac = fmaf(rhs[k], w[(size_t)k * H], ac);
                const float c = fast_tanh(ac);
                const float hp = hs[j];
                const float hn = u_[q] * hp + (1.f - u_[q]) * c;
                float o = hn;
                if (a.keep < 1.0f) o *= keep_scale(a.seed, (uint32_t)(b * a.Tout + t), (uint32_t)(dir * H + j), a.keep);
                a.out[((size_t)b * a.Tout + t) * (a.ND * H) + dir * H + j] = o;
                if (a.hprev) { a.hprev[row * H + j] = hp; a.rh[row * H + j] = rhs[j]; gxr[j] = r_[q]; gxr[H + j] = u_[q]; cxr[j] = c; }
                r_[q] = hn;            // (carried to the LDS update below)
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < UPT; ++q) { const int j = tid + q * NT; if (j < H) hs[j] = r_[q]; }
        __syncthreads();
    }
    if (a.h_last) for (int j = tid; j < H; j += NT) a.h_last[((size_t)b * a.ND + dir) * H + j] = hs[j];
    // zero output past the length (dynamic_rnn zero-fill; also the pyramid pad frame), zero saved rows there
    for (int idx = tid; idx < (a.Tout - S) * H; idx += NT) {
        const int t = S + idx / H, j = idx % H;
        a.out[((size_t)b * a.Tout + t) * (a.ND * H) + dir * H + j] = 0.f;
    }
    if (a.hprev)
        for (int idx = tid; idx < (a.T - S) * H; idx += NT) {
            const int t = S + idx / H, j = idx % H;
            const size_t row = ((size_t)b * a.T + t) * a.ND + dir;
            a.hprev[row * H + j] = 0.f; a.rh[row * H + j] = 0.f;
        }
}

// backward: wgh = W_gates[in:]^T [2H][H], wch = W_cand[in:]^T [H][H]; gx / cx hold the activated gates and are overwritten with dG
template <int NT, int UPT>
__global__ __launch_bounds__(NT) void gru_rec_bwd_kernel(GruArgs a) {
    extern __shared__ float sm[];
    const int H = a.H, H2 = 2 * H;
    float* dcs = sm;           // dG_cand of this step [H]
    float* dgs = sm + H;       // dG_gates of this step [2H]
    const int b = blockIdx.x, dir = a.dir;
    const int tid = threadIdx.x;
    const int S = min(max(a.len[b], 0), a.T);
    float dhc[UPT];            // dh carried to the earlier step, per owned unit
#pragma unroll
    for (int q = 0; q < UPT; ++q) { const int j = tid + q * NT; dhc[q] = (a.h0 && j < H) ? a.h0[((size_t)b * a.ND + dir) * H + j] : 0.f; }
    for (int s = 0; s < S; ++s) {
        const int t = dir ? s : (S - 1 - s);               // the forward's last step first
        const size_t row = ((size_t)b * a.T + t) * a.ND + dir;
        float* gxr = a.gx + row * H2;
        float* cxr = a.cx + row * H;
        float r_[UPT], u_[UPT], hp_[UPT], du_[UPT], carry[UPT];
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int j = tid + q * NT;
            if (j < H) {
                r_[q] = gxr[j]; u_[q] = gxr[H + j];
                const float c = cxr[j];
                hp_[q] = a.hprev[row * H + j];
                float dm = a.dout[((size_t)b * a.Tout + t) * (a.ND * H) + dir * H + j];
                if (a.keep < 1.0f) dm *= keep_scale(a.seed, (uint32_t)(b * a.Tout + t), (uint32_t)(dir * H + j), a.keep);
                const float dh = dm + dhc[q];
                du_[q] = dh * (hp_[q] - c);
                carry[q] = dh * u_[q];
                const float dcp = dh * (1.f - u_[q]) * (1.f - c * c);
                dcs[j] = dcp;
                cxr[j] = dcp;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int j = tid + q * NT;
            if (j < H) {
                float drh = 0.f;                             // d(r * h_prev)[j] = sum_k dcp[k] W_cand[in + j][k]
                const float* w = a.wch + j;
#pragma unroll 8
                for (int k = 0; k < H; ++k) drh = fmaf(dcs[k], w[(size_t)k * H], drh);
                carry[q] += drh * r_[q];
                const float drp = drh * hp_[q] * r_[q] * (1.f - r_[q]);
                const float dup = du_[q] * u_[q] * (1.f - u_[q]);
                dgs[j] = drp; dgs[H + j] = dup;
                gxr[j] = drp; gxr[H + j] = dup;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int j = tid + q * NT;
            if (j < H) {
                float acc = carry[q];                        // + sum_m dG_gates[m] W_gates[in + j][m]
                const float* w = a.wgh + j;
#pragma unroll 8
                for (int m = 0; m < H2; ++m) acc = fmaf(dgs[m], w[(size_t)m * H], acc);
                dhc[q] = acc;
            }
        }
        __syncthreads();
    }
    if (a.h_last) {
#pragma unroll
        for (int q = 0; q < UPT; ++q) { const int j = tid + q * NT; if (j < H) a.h_last[((size_t)b * a.ND + dir) * H + j] = dhc[q]; }
    }
    // dG = 0 past the row's length (the weight / input GEMMs read every row)
    for (int idx = tid; idx < (a.T - S) * H; idx += NT) {
        const int t = S + idx / H, j = idx % H;
        const size_t row = ((size_t)b * a.T + t) * a.ND + dir;
        a.gx[row * H2 + j] = 0.f; a.gx[row * H2 + H + j] = 0.f; a.cx[row * H + j] = 0.f;
    }
}

template <int NT, int UPT>
static int gru_launch(bool bwd, hipStream_t s, const GruArgs& a) {
    const size_t lds = (size_t)(bwd ? 3 : 2) * a.H * sizeof(float);
    if (bwd) hipLaunchKernelGGL((gru_rec_bwd_kernel<NT, UPT>), dim3(a.B), dim3(NT), lds, s, a);
    else     hipLaunchKernelGGL((gru_rec_fwd_kernel<NT, UPT>), dim3(a.B), dim3(NT), lds, s, a);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}
static int gru_dispatch(bool bwd, hipStream_t s, const GruArgs& a) {
    if (a.H <= 256) return gru_launch<256, 1>(bwd, s, a);
    if (a.H <= 512) return gru_launch<256, 2>(bwd, s, a);
    if (a.H <= 1024) return gru_launch<256, 4>(bwd, s, a);
    return ASR_EUNSUPPORTED;
}

}  // namespace asr

// One GRU layer, forward (encoder.py:55-91 with the cell of encoder.py:47-48).  x [B][T][in] (row pitch ldx), len [B]; per direction
// d: wg[d] [in+H][2H], bg[d] [2H], wc[d] [in+H][H], bc[d] [H] (TF layouts).  out [B][Tout][ndir*H] (zeros past each length).
// gx [B][T][ndir][2H], cx [B][T][ndir][H]: workspaces; hprev / rh [B][T][ndir][H] non-NULL = save for the backward pass (gx / cx
// then hold the activated r | u and c).  Dropout: output-only, mask keep_scale(seed, b*Tout + t, dir*H + unit).
// h0 [B][ndir][H] (NULL: zeros) = the initial state; h_last [B][ndir][H] (NULL: not wanted) = the final PLAIN state -- a caller that
// composes its own time loop (the GRU attention decoder, e2e_asr_amd/gru_decoder.py) runs T = 1 steps with them.
extern "C" int asr_gru_layer_fwd(void* stream, const float* x, int B, int T, int in_dim, int ldx, const int* len, int H, int ndir,
                                 const float* const* wg, const float* const* bg, const float* const* wc, const float* const* bc,
                                 float* out, int Tout, float* gx, float* cx, float* hprev, float* rh, float keep_prob, unsigned seed,
                                 const float* h0, float* h_last) {
    using namespace asr;
    if (!x || !len || !wg || !bg || !wc || !bc || !out || !gx || !cx || (hprev != nullptr) != (rh != nullptr)) return ASR_EINVAL;
    if (B <= 0 || T <= 0 || in_dim <= 0 || H <= 0 || Tout < T || ldx < in_dim || (ndir != 1 && ndir != 2)) return ASR_EINVAL;
    if (H > 1024) return ASR_EUNSUPPORTED;
    int rc;
    const int M = B * T;
    for (int d = 0; d < ndir; ++d) {
        if (!wg[d] || !bg[d] || !wc[d] || !bc[d]) return ASR_EINVAL;
        // x parts of both products for all time steps: gx[:, d] = x . W_gates[:in] + b_gates, cx[:, d] = x . W_cand[:in] + b_cand
        if ((rc = asr_gemm_f32(stream, 0, 0, M, 2 * H, in_dim, x, ldx, wg[d], 2 * H, gx + (size_t)d * 2 * H, ndir * 2 * H, bg[d], 0))) return rc;
        if ((rc = asr_gemm_f32(stream, 0, 0, M, H, in_dim, x, ldx, wc[d], H, cx + (size_t)d * H, ndir * H, bc[d], 0))) return rc;
    }
    for (int d = 0; d < ndir; ++d) {          // one launch per direction (its own weights); B workgroups each
        GruArgs a;
        a.wgh = wg[d] + (size_t)in_dim * 2 * H; a.wch = wc[d] + (size_t)in_dim * H;
        a.gx = gx; a.cx = cx; a.hprev = hprev; a.rh = rh; a.out = out; a.dout = nullptr; a.len = len;
        a.h0 = h0; a.h_last = h_last;
        a.B = B; a.T = T; a.Tout = Tout; a.ND = ndir; a.H = H; a.dir = d; a.keep = keep_prob; a.seed = seed;
        if ((rc = gru_dispatch(false, static_cast<hipStream_t>(stream), a))) return rc;
    }
    return ASR_OK;
}

// Backward of asr_gru_layer_fwd (tf.gradients through the layer, seq2seq_model.py:148).  gx / cx / hprev / rh: what the forward
// saved; gx / cx are overwritten with dG_gates / dG_cand.  dout [B][Tout][ndir*H].  wt_ws: >= ndir * 3*H*H floats (the transposed
// recurrent weights).  Weight / bias gradients are ACCUMULATED into dwg / dbg / dwc / dbc (same layouts as the weights); dx
// [B][T][in] (NULL: not needed) is overwritten.  dh_last [B][ndir][H] (NULL: zero) = gradient w.r.t. the final state (what the
// forward returned in h_last); dh0 [B][ndir][H] (NULL: not wanted) = gradient w.r.t. h0.  Everything on `stream`.
extern "C" int asr_gru_layer_bwd(void* stream, const float* x, int B, int T, int in_dim, int ldx, const int* len, int H, int ndir,
                                 const float* const* wg, const float* const* wc, const float* dout, int Tout,
                                 float* gx, float* cx, const float* hprev, const float* rh, float* wt_ws,
                                 float* const* dwg, float* const* dbg, float* const* dwc, float* const* dbc, float* dx,
                                 float keep_prob, unsigned seed, const float* dh_last, float* dh0) {
    using namespace asr;
    if (!x || !len || !wg || !wc || !dout || !gx || !cx || !hprev || !rh || !wt_ws || !dwg || !dbg || !dwc || !dbc) return ASR_EINVAL;
    if (B <= 0 || T <= 0 || in_dim <= 0 || H <= 0 || Tout < T || ldx < in_dim || (ndir != 1 && ndir != 2)) return ASR_EINVAL;
    if (H > 1024) return ASR_EUNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc;
    const int M = B * T, H2 = 2 * H;
    for (int d = 0; d < ndir; ++d) {
        if (!wg[d] || !wc[d] || !dwg[d] || !dbg[d] || !dwc[d] || !dbc[d]) return ASR_EINVAL;
        float* wgT = wt_ws + (size_t)d * 3 * H * H;        // [2H][H] = W_gates[in:]^T
        float* wcT = wgT + (size_t)2 * H * H;              // [H][H]  = W_cand[in:]^T
        if ((rc = transpose_add(s, wgT, H, wg[d] + (size_t)in_dim * H2, H2, H, 0))) return rc;      // C[m][n] = T[n][m], T [N = H][M = 2H]
        if ((rc = transpose_add(s, wcT, H, wc[d] + (size_t)in_dim * H, H, H, 0))) return rc;
        GruArgs a;
        a.wgh = wgT; a.wch = wcT;
        a.gx = gx; a.cx = cx; a.hprev = const_cast<float*>(hprev); a.rh = const_cast<float*>(rh); a.out = nullptr; a.dout = dout; a.len = len;
        a.h0 = dh_last; a.h_last = dh0;
        a.B = B; a.T = T; a.Tout = Tout; a.ND = ndir; a.H = H; a.dir = d; a.keep = keep_prob; a.seed = seed;
        if ((rc = gru_dispatch(true, s, a))) return rc;
    }
    for (int d = 0; d < ndir; ++d) {
        const float* dGg = gx + (size_t)d * H2; const int ldg = ndir * H2;
        const float* dGc = cx + (size_t)d * H;  const int ldc_ = ndir * H;
        // dW_gates = [x | h_prev]^T . dG_gates;  dW_cand = [x | r*h_prev]^T . dG_cand;  biases = column sums
        if ((rc = asr_gemm_f32(stream, 1, 0, in_dim, H2, M, x, ldx, dGg, ldg, dwg[d], H2, nullptr, 1))) return rc;
        if ((rc = asr_gemm_f32(stream, 1, 0, H, H2, M, hprev + (size_t)d * H, ndir * H, dGg, ldg, dwg[d] + (size_t)in_dim * H2, H2, nullptr, 1))) return rc;
        if ((rc = asr_colsum_f32(stream, dGg, ldg, M, H2, dbg[d], 1))) return rc;
        if ((rc = asr_gemm_f32(stream, 1, 0, in_dim, H, M, x, ldx, dGc, ldc_, dwc[d], H, nullptr, 1))) return rc;
        if ((rc = asr_gemm_f32(stream, 1, 0, H, H, M, rh + (size_t)d * H, ndir * H, dGc, ldc_, dwc[d] + (size_t)in_dim * H, H, nullptr, 1))) return rc;
        if ((rc = asr_colsum_f32(stream, dGc, ldc_, M, H, dbc[d], 1))) return rc;
        if (dx) {       // dx = dG_gates . W_gates[:in]^T + dG_cand . W_cand[:in]^T, summed over the directions
            if ((rc = asr_gemm_f32(stream, 0, 1, M, in_dim, H2, dGg, ldg, wg[d], H2, dx, in_dim, nullptr, d > 0))) return rc;
            if ((rc = asr_gemm_f32(stream, 0, 1, M, in_dim, H, dGc, ldc_, wc[d], H, dx, in_dim, nullptr, 1))) return rc;
        }
    }
    return ASR_OK;
}
