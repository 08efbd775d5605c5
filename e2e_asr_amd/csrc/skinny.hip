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
#include "skinny_body.h"

namespace asr {


template <bool LSTM, bool WT>
__global__ __launch_bounds__(256) void skinny_kernel(SkinnyArgs a) {
    __shared__ float red[3][4][64];
    skinny_body<LSTM, WT>(a, blockIdx.x, blockIdx.y, red, threadIdx.x);
}
template <bool LSTM>
__global__ __launch_bounds__(256) void skinny_pair_kernel(SkinnyArgs a0, SkinnyArgs a1, int nbx0) {
    __shared__ float red[3][4][64];
    if ((int)blockIdx.x < nbx0) skinny_body<LSTM, false>(a0, blockIdx.x, blockIdx.y, red, threadIdx.x);
    else skinny_body<LSTM, false>(a1, blockIdx.x - nbx0, blockIdx.y, red, threadIdx.x);
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
