// Skinny (M = batch <= a few 16-row tiles) dense layers of the attention-decoder step.
//
// Reference ops (attn_decoder.py): `_linear([a,b], n, True)` = concat(a,b).W + bias at
// :117 (AttnProjection), :122/125 (OutputProjection), :151 (SimpleProjection), :158
// (InputProjection); and the two BasicLSTMCell calls per step (:148 lm cell, :166 outer
// cell through raw_rnn) whose arithmetic is basic_lstm.py:14-23.
//
// One workgroup = one 16(columns) x 16(batch rows) output tile on v_mfma_f32_16x16x4_f32
// (exact f32).  W is streamed ONCE straight from global/L2 into the MFMA A-operand
// registers (no LDS: it is not shared between waves -- each of the 4 waves owns a quarter
// of K); X rows are read as 16-byte vectors.  The 4 K-partials meet in LDS and wave 0
// runs the epilogue.  The concat of the reference is never materialised: the K loop
// switches source pointer at K1.  In LSTM mode the 16 tile columns are 4 hidden units x
// their 4 gates (column n_local = 4*unit + gate -> TF column gate*H + j), so that the
// MFMA result layout (row = 4*(lane>>4) + reg) hands each lane the i,j,f,o of ONE unit for
// ONE batch row and the cell runs in registers.
#include "common.h"
#include "skinny.h"

namespace asr {


// WT: the weight is given transposed, W[n][k] (row stride ldw along n) -- the data-gradient
// products dY.W^T of the backward pass; the lane then reads 4 consecutive k as one float4.
template <bool LSTM, bool WT>
__device__ __forceinline__ void skinny_body(const SkinnyArgs& a, const int bx, const int by, float (*red)[4][64]) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nl = lane & 15, g4 = lane >> 4;
    const int K = a.K1 + a.K2;
    // column of W feeding output row nl of the MFMA tile
    int colW; bool colok;
    if (LSTM) { const int j = bx * 4 + (nl >> 2); colW = (nl & 3) * a.H + j; colok = j < a.H; }
    else      { colW = bx * 16 + nl; colok = colW < a.N; }
    const int brow = by * 16 + nl;      // batch row this lane supplies as MFMA B operand
    const bool rowok = brow < a.M;
    const float* xr1 = nullptr; const float* xr2 = nullptr;
    if (rowok) {
        xr1 = a.x1 + (size_t)(a.gather1 ? a.gather1[brow] : brow) * a.ld1;
        if (a.K2 > 0) xr2 = a.x2 + (size_t)(a.gather2 ? a.gather2[brow] : brow) * a.ld2;
    }
    const int chunk = ((K + 63) / 64) * 16;
    const int kbeg = w * chunk, kend = min(K, kbeg + chunk);
    // epilogue operands (wave 0 only) are requested NOW so that their latency hides under the
    // weight stream instead of forming a second dependent memory round trip after the reduction
    float eb0 = 0.f, eb1 = 0.f, eb2 = 0.f, eb3 = 0.f, ecp = 0.f;
    bool ez = false;
    if (w == 0 && rowok) {
        if (LSTM) {
            const int j = bx * 4 + g4;
            if (j < a.H) {
                eb0 = a.bias[j]; eb1 = a.bias[a.H + j]; eb2 = a.bias[2 * a.H + j]; eb3 = a.bias[3 * a.H + j];
                if (a.c_prev) ecp = a.c_prev[(size_t)(a.gather2 ? a.gather2[brow] : brow) * a.H + j];
            }
        } else {
            const int n = bx * 16 + 4 * g4;
            if (a.bias) {
                if (n < a.N) eb0 = a.bias[n];
                if (n + 1 < a.N) eb1 = a.bias[n + 1];
                if (n + 2 < a.N) eb2 = a.bias[n + 2];
                if (n + 3 < a.N) eb3 = a.bias[n + 3];
            }
            ez = a.zero_from && a.zero_t >= a.zero_from[brow];
        }
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};   // two independent MFMA chains
    // All loads of a group of IT k-steps are issued before the first MFMA consumes one, so
    // a wave pays ~one memory round trip per group instead of one per k-step.
    constexpr int IT = 12;
    for (int kb0 = kbeg; kb0 < kend; kb0 += 16 * IT) {
        float4 xv[IT]; float4 wv[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int k = kb0 + 16 * it + 4 * g4;
            xv[it] = make_float4(0.f, 0.f, 0.f, 0.f); wv[it] = xv[it];
            if (k < kend) {
                if (rowok) xv[it] = (k < a.K1) ? *reinterpret_cast<const float4*>(xr1 + k)
                                               : *reinterpret_cast<const float4*>(xr2 + (k - a.K1));
                if (colok) {
                    if (WT) {
                        wv[it] = *reinterpret_cast<const float4*>(a.W + (size_t)colW * a.ldw + k);
                    } else {
                        const float* wp = a.W + (size_t)k * a.ldw + colW;
                        wv[it].x = wp[0]; wv[it].y = wp[a.ldw];
                        wv[it].z = wp[2 * (size_t)a.ldw]; wv[it].w = wp[3 * (size_t)a.ldw];
                    }
                }
            }
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[it].x, xv[it].x, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[it].y, xv[it].y, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[it].z, xv[it].z, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[it].w, xv[it].w, acc2, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
    if (w > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[w - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (w != 0) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += red[0][r][lane] + red[1][r][lane] + red[2][r][lane];
    // lane holds D[n_local = 4*g4 + r][batch = nl]
    const int b = by * 16 + nl;
    if (b >= a.M) return;
    if (LSTM) {
        const int j = bx * 4 + g4;
        if (j >= a.H) return;
        const int H = a.H;
        const float gi = fast_sigmoid(acc[0] + eb0);
        const float gj = fast_tanh(acc[1] + eb1);
        const float gf = fast_sigmoid(acc[2] + eb2 + 1.0f);
        const float go = fast_sigmoid(acc[3] + eb3);
        const float cp = ecp;
        const float c = cp * gf + gi * gj;
        const float h = go * fast_tanh(c);
        a.c_out[(size_t)b * H + j] = c;
        a.h_out[(size_t)b * H + j] = h;
        if (a.hdrop_out)
            a.hdrop_out[(size_t)b * H + j] = h * keep_scale(a.seed, a.step * (uint32_t)a.M + (uint32_t)b, (uint32_t)j, a.keep);
        if (a.gates_out) {
            float* gp = a.gates_out + (size_t)b * 4 * H + j;
            gp[0] = gi; gp[H] = gj; gp[2 * H] = gf; gp[3 * H] = go;
        }
    } else {
        const int n = bx * 16 + 4 * g4;
        const bool z = ez;
        const float eb[4] = {eb0, eb1, eb2, eb3};
        float* op = a.out + (size_t)b * a.ldo + n;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < a.N) {
                float v = z ? 0.f : acc[r] + eb[r];
                if (a.accumulate) v += op[r];
                op[r] = v;
            }
    }
}

template <bool LSTM, bool WT>
__global__ __launch_bounds__(256) void skinny_kernel(SkinnyArgs a) {
    __shared__ float red[3][4][64];
    skinny_body<LSTM, WT>(a, blockIdx.x, blockIdx.y, red);
}
template <bool LSTM>
__global__ __launch_bounds__(256) void skinny_pair_kernel(SkinnyArgs a0, SkinnyArgs a1, int nbx0) {
    __shared__ float red[3][4][64];
    if ((int)blockIdx.x < nbx0) skinny_body<LSTM, false>(a0, blockIdx.x, blockIdx.y, red);
    else skinny_body<LSTM, false>(a1, blockIdx.x - nbx0, blockIdx.y, red);
}

}  // namespace asr
static int skinny_check(const asr::SkinnyArgs& a);
namespace asr {

int skinny_launch_pair(hipStream_t s, bool lstm, const SkinnyArgs& a0, const SkinnyArgs& a1) {
    if (int rc = skinny_check(a0)) return rc;
    if (int rc = skinny_check(a1)) return rc;
    if (lstm ? (a0.H <= 0 || a1.H <= 0 || (a0.H & 3) || (a1.H & 3)) : (a0.N <= 0 || a1.N <= 0 || !a0.out || !a1.out)) return ASR_EINVAL;
    const int nb0 = lstm ? (a0.H + 3) / 4 : (a0.N + 15) / 16, nb1 = lstm ? (a1.H + 3) / 4 : (a1.N + 15) / 16;
    const int M = a0.M > a1.M ? a0.M : a1.M;
    dim3 grid(nb0 + nb1, (M + 15) / 16);
    if (lstm) hipLaunchKernelGGL((skinny_pair_kernel<true>), grid, dim3(256), 0, s, a0, a1, nb0);
    else hipLaunchKernelGGL((skinny_pair_kernel<false>), grid, dim3(256), 0, s, a0, a1, nb0);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

int skinny_launch(hipStream_t s, bool lstm, const SkinnyArgs& a) {
    if (int rc = skinny_check(a)) return rc;
    if (lstm ? (a.H <= 0 || (a.H & 3) || !a.c_out || !a.h_out) : (a.N <= 0 || !a.out)) return ASR_EINVAL;
    dim3 grid(lstm ? (a.H + 3) / 4 : (a.N + 15) / 16, (a.M + 15) / 16);
    if (lstm) hipLaunchKernelGGL((skinny_kernel<true, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((skinny_kernel<false, false>), grid, dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

}  // namespace asr

static int skinny_check(const asr::SkinnyArgs& a) {
    if (!a.x1 || !a.W || a.M <= 0 || a.K1 <= 0 || a.K2 < 0) return ASR_EINVAL;
    if ((a.K1 & 3) || (a.K2 & 3) || (a.ld1 & 3) || (a.K2 > 0 && (!a.x2 || (a.ld2 & 3)))) return ASR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(a.x1) & 15) || (a.x2 && (reinterpret_cast<uintptr_t>(a.x2) & 15))) return ASR_EINVAL;
    return ASR_OK;
}

// out[M,N] = [X1 | X2] . W + bias   (rows b with zero_t >= zero_from[b] are emitted as 0)
extern "C" int asr_linear_fwd(void* stream, const float* x1, int ld1, int K1, const int* gather1,
                              const float* x2, int ld2, int K2, const float* W, int ldw,
                              const float* bias, float* out, int ldo, int M, int N,
                              const int* zero_from, int zero_t) {
    asr::SkinnyArgs a{};
    a.x1 = x1; a.ld1 = ld1; a.K1 = K1; a.gather1 = gather1; a.x2 = x2; a.ld2 = ld2; a.K2 = K2;
    a.W = W; a.ldw = ldw; a.bias = bias; a.M = M; a.N = N; a.out = out; a.ldo = ldo;
    a.zero_from = zero_from; a.zero_t = zero_t;
    if (int rc = skinny_check(a)) return rc;
    if (!out || N <= 0 || ldw < N || ldo < N) return ASR_EINVAL;
    dim3 grid((N + 15) / 16, (M + 15) / 16);
    hipLaunchKernelGGL((asr::skinny_kernel<false, false>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// BasicLSTMCell step: (c,h) = cell([X1 | h_prev], c_prev) with TF kernel [K1+H, 4H], gates i,j,f,o.
extern "C" int asr_lstm_cell_fwd(void* stream, const float* x1, int ld1, int K1, const int* gather1,
                                 const float* h_prev, const float* c_prev, const float* kernel,
                                 const float* bias, int H, int M, float* c_out, float* h_out,
                                 float* hdrop_out, float* gates_out, float keep, unsigned seed,
                                 unsigned step) {
    asr::SkinnyArgs a{};
    a.x1 = x1; a.ld1 = ld1; a.K1 = K1; a.gather1 = gather1; a.x2 = h_prev; a.ld2 = H; a.K2 = H;
    a.W = kernel; a.ldw = 4 * H; a.bias = bias; a.M = M; a.N = 4 * H; a.H = H;
    a.c_prev = c_prev; a.c_out = c_out; a.h_out = h_out; a.hdrop_out = hdrop_out; a.gates_out = gates_out;
    a.keep = keep; a.seed = seed; a.step = step;
    if (int rc = skinny_check(a)) return rc;
    if (!h_prev || !bias || !c_out || !h_out || H <= 0 || (H & 3)) return ASR_EINVAL;
    dim3 grid((H + 3) / 4, (M + 15) / 16);
    hipLaunchKernelGGL((asr::skinny_kernel<true, false>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// out[M,N] (+)= X[M,K] . Wt^T + bias, Wt given as [N,K] (row stride ldw): the data-gradient
// products of the backward pass, e.g. [dx | dh_prev] = dGates . K^T with K the TF kernel [in+H, 4H].
extern "C" int asr_linear_wt_fwd(void* stream, const float* x, int ldx, int K, const float* Wt, int ldw,
                                 float* out, int ldo, int M, int N, int accumulate) {
    asr::SkinnyArgs a{};
    a.x1 = x; a.ld1 = ldx; a.K1 = K; a.W = Wt; a.ldw = ldw; a.M = M; a.N = N; a.out = out; a.ldo = ldo;
    a.accumulate = accumulate;
    if (int rc = skinny_check(a)) return rc;
    if (!out || N <= 0 || ldw < K || ldo < N || (ldw & 3) || (reinterpret_cast<uintptr_t>(Wt) & 15)) return ASR_EINVAL;
    dim3 grid((N + 15) / 16, (M + 15) / 16);
    hipLaunchKernelGGL((asr::skinny_kernel<false, true>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
