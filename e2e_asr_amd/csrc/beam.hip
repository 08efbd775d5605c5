// One beam-search step for all live hypotheses of ONE utterance -- the arithmetic of BeamSearch.get_top_k
// (beam_search.py:163-221) up to the two logit vectors: decoder LM cell on the fed token, [SimpleProjection],
// InputProjection with the previous context, outer cell, attention over the shared encoder states (query = cell
// state c), AttnProjection, OutputProjection; and the external LM cell + its output projection for shallow fusion
// (:200-207).  Scoring (float64 log-softmax, lm_weight, argpartition) stays on the host as the reference defines it.
// Ten stream-ordered launches of the step kernels behind ONE call (skinny.hip, attention.hip).
#include "common.h"
#include "skinny.h"
#include "skinny_body.h"
#include "attention_body.h"
#include <type_traits>
#include <cstdlib>
#include "../../include/e2e_asr_hip.h"

extern "C" int asr_beam_step_perm(void* stream, const asr_dec_weights* w, const asr_lm_weights* lm, const asr_dec_dims* d,
                                  const float* hf, const float* enc, const int* enc_len, const int* tokens, const int* sel,
                                  const asr_beam_state* in, const asr_beam_state* out, float* scratch,
                                  float* logits, float* logits_lm, const float* lm_kernel_t, const float* ext_kernel_t,
                                  const float* dec_kernel_t);
extern "C" int asr_beam_step_sel(void* stream, const asr_dec_weights* w, const asr_lm_weights* lm, const asr_dec_dims* d,
                                 const float* hf, const float* enc, const int* enc_len, const int* tokens, const int* sel,
                                 const asr_beam_state* in, const asr_beam_state* out, float* scratch,
                                 float* logits, float* logits_lm);
extern "C" int asr_beam_step(void* stream, const asr_dec_weights* w, const asr_lm_weights* lm, const asr_dec_dims* d,
                             const float* hf, const float* enc, const int* enc_len, const int* tokens,
                             const asr_beam_state* in, const asr_beam_state* out, float* scratch,
                             float* logits, float* logits_lm) {
    return asr_beam_step_sel(stream, w, lm, d, hf, enc, enc_len, tokens, nullptr, in, out, scratch, logits, logits_lm);
}
// sel (device, [k] ints, or NULL): row r of the step's input state is row sel[r] of `in` -- the parents of the surviving
// hypotheses (beam_search.py:306-318) read in place by the step kernels instead of being gathered by a launch of their own.
// With sel the SimpleProjections must be absent (ASR_EUNSUPPORTED otherwise: gather with asr_beam_gather, then sel = NULL).
extern "C" int asr_beam_step_sel(void* stream, const asr_dec_weights* w, const asr_lm_weights* lm, const asr_dec_dims* d,
                                 const float* hf, const float* enc, const int* enc_len, const int* tokens, const int* sel,
                                 const asr_beam_state* in, const asr_beam_state* out, float* scratch,
                                 float* logits, float* logits_lm) {
    return asr_beam_step_perm(stream, w, lm, d, hf, enc, enc_len, tokens, sel, in, out, scratch, logits, logits_lm, nullptr, nullptr, nullptr);
}
// ... with the three LSTM kernels ALSO given in tile order (asr_lstm_kernel_tile_order: column 16*tile + 4*unit + gate next to each
// other, the order the skinny LSTM tiles read them in), or NULL: a tile's 16 weight columns of a K row are then ONE 64-byte
// segment instead of four 16-byte pieces a gate block apart -- the two LM cells 9.8 -> 5.9 us, the outer cell 7.7 -> 5.0 us per
// token (same values, same arithmetic: results are bit-identical).  The weights are constants of a decode: permute once.
extern "C" int asr_beam_step_perm(void* stream, const asr_dec_weights* w, const asr_lm_weights* lm, const asr_dec_dims* d,
                                  const float* hf, const float* enc, const int* enc_len, const int* tokens, const int* sel,
                                  const asr_beam_state* in, const asr_beam_state* out, float* scratch,
                                  float* logits, float* logits_lm, const float* lm_kernel_t, const float* ext_kernel_t,
                                  const float* dec_kernel_t) {
    if (!w || !lm || !d || !hf || !enc || !enc_len || !tokens || !in || !out || !scratch || !logits || !logits_lm) return ASR_EINVAL;
    if (sel && (w->simple_w || lm->simple_w)) return ASR_EUNSUPPORTED;
    const int k = d->B, Te = d->Te, D = d->D, A = d->A, H = d->H, lmH = d->lmH, E = d->E, V = d->V;
    if (k <= 0) return ASR_EINVAL;
    int rc;
    float* sp = scratch;                              // [k, H]   SimpleProjection output
    float* x = sp + (size_t)k * H;                    // [k, E]
    float* p = x + (size_t)k * E;                     // [k, H]
    float* alpha = p + (size_t)k * H;                 // [k, Te]
    float* lsp = alpha + (size_t)k * Te;              // [k, lm->P] external LM SimpleProjection output
    // Without SimpleProjections the two LM cells (decoder's inner one, :183-186, and the external one, :200-203) and then the
    // two projections behind them (InputProjection, :188-189, and the external LM's output projection, :204-207) are pairwise
    // independent: two launches instead of four.
    const bool paired = !w->simple_w && !lm->simple_w;
    if (paired) {
        asr::SkinnyArgs c0{}, c1{};
        c0.x1 = w->embedding; c0.ld1 = E; c0.K1 = E; c0.gather1 = tokens; c0.x2 = in->dlh; c0.ld2 = lmH; c0.K2 = lmH; c0.gather2 = sel;
        c0.W = lm_kernel_t ? lm_kernel_t : w->lm_kernel; c0.wperm = lm_kernel_t != nullptr;
        c0.ldw = 4 * lmH; c0.bias = w->lm_bias; c0.M = k; c0.N = 4 * lmH; c0.H = lmH;
        c0.c_prev = in->dlc; c0.c_out = out->dlc; c0.h_out = out->dlh; c0.keep = 1.0f;
        c1.x1 = lm->embedding; c1.ld1 = lm->E; c1.K1 = lm->E; c1.gather1 = tokens; c1.x2 = in->lh; c1.ld2 = lm->H; c1.K2 = lm->H; c1.gather2 = sel;
        c1.W = ext_kernel_t ? ext_kernel_t : lm->lstm_kernel; c1.wperm = ext_kernel_t != nullptr;
        c1.ldw = 4 * lm->H; c1.bias = lm->lstm_bias; c1.M = k; c1.N = 4 * lm->H; c1.H = lm->H;
        c1.c_prev = in->lc; c1.c_out = out->lc; c1.h_out = out->lh; c1.keep = 1.0f;
        if ((rc = asr::skinny_launch_pair(static_cast<hipStream_t>(stream), true, c0, c1))) return rc;
        asr::SkinnyArgs p0{}, p1{};
        p0.x1 = out->dlh; p0.ld1 = lmH; p0.K1 = lmH; p0.x2 = in->ctx; p0.ld2 = D; p0.K2 = D; p0.gather2 = sel;
        p0.W = w->inp_w; p0.ldw = E; p0.bias = w->inp_b; p0.M = k; p0.N = E; p0.out = x; p0.ldo = E;
        p1.x1 = out->lh; p1.ld1 = lm->H; p1.K1 = lm->H; p1.W = lm->out_w; p1.ldw = lm->V; p1.bias = lm->out_b;
        p1.M = k; p1.N = lm->V; p1.out = logits_lm; p1.ldo = lm->V;
        if ((rc = asr::skinny_launch_pair(static_cast<hipStream_t>(stream), false, p0, p1))) return rc;
    }
    // decoder's inner LM cell on emb[token]  (:183-186)
    if (!paired && (rc = asr_lstm_cell_fwd(stream, w->embedding, E, E, tokens, in->dlh, in->dlc, w->lm_kernel, w->lm_bias, lmH, k,
                                           out->dlc, out->dlh, nullptr, nullptr, 1.0f, 0, 0))) return rc;
    const float* o = out->dlh; int P = lmH;
    if (w->simple_w) {
        if ((rc = asr_linear_fwd(stream, o, lmH, lmH, nullptr, nullptr, 0, 0, w->simple_w, H, w->simple_b, sp, H, k, H, nullptr, 0))) return rc;
        o = sp; P = H;
    }
    // x = [lm_out, ctx_prev] . W_inp + b  (:188-189), outer cell (:190-191)
    if (!paired && (rc = asr_linear_fwd(stream, o, P, P, nullptr, in->ctx, D, D, w->inp_w, E, w->inp_b, x, E, k, E, nullptr, 0))) return rc;
    {
        asr::SkinnyArgs oc{};
        oc.x1 = x; oc.ld1 = E; oc.K1 = E; oc.x2 = in->dh; oc.ld2 = H; oc.K2 = H; oc.gather2 = sel;
        oc.W = dec_kernel_t ? dec_kernel_t : w->dec_kernel; oc.wperm = dec_kernel_t != nullptr;
        oc.ldw = 4 * H; oc.bias = w->dec_bias; oc.M = k; oc.N = 4 * H; oc.H = H;
        oc.c_prev = in->dc; oc.c_out = out->dc; oc.h_out = out->dh; oc.keep = 1.0f;
        if ((rc = asr::skinny_launch(static_cast<hipStream_t>(stream), true, oc))) return rc;
    }
    // attention with query = c (:193), AttnProjection, OutputProjection (:194-198)
    if ((rc = asr_attention_shared_fwd(stream, out->dc, H, w->attn_w, w->attn_b, w->attn_v, hf, enc, enc_len, alpha, out->ctx,
                                       k, Te, H, A, D, 1))) return rc;
    if ((rc = asr_linear_fwd(stream, out->dc, H, H, nullptr, out->ctx, D, D, w->ap_w, H, w->ap_b, p, H, k, H, nullptr, 0))) return rc;
    if ((rc = asr_linear_fwd(stream, p, H, H, nullptr, nullptr, 0, 0, w->out_w, V, w->out_b, logits, V, k, V, nullptr, 0))) return rc;
    // external LM (:200-207)
    if (paired) return ASR_OK;
    if ((rc = asr_lstm_cell_fwd(stream, lm->embedding, lm->E, lm->E, tokens, in->lh, in->lc, lm->lstm_kernel, lm->lstm_bias,
                                lm->H, k, out->lc, out->lh, nullptr, nullptr, 1.0f, 0, 0))) return rc;
    const float* lo = out->lh; int LP = lm->H;
    if (lm->simple_w) {
        if ((rc = asr_linear_fwd(stream, lo, lm->H, lm->H, nullptr, nullptr, 0, 0, lm->simple_w, lm->P, lm->simple_b, lsp, lm->P, k, lm->P,
                                 nullptr, 0))) return rc;
        lo = lsp; LP = lm->P;
    }
    return asr_linear_fwd(stream, lo, LP, LP, nullptr, nullptr, 0, 0, lm->out_w, lm->V, lm->out_b, logits_lm, lm->V, k, lm->V, nullptr, 0);
}

namespace asr {
// out[k][16*tile + 4*unit + gate] = W[k][gate*H + 4*tile + unit]: the column order of the skinny LSTM tiles (csrc/skinny_body.h)
__global__ __launch_bounds__(256) void lstm_kernel_tile_order_kernel(const float* W, float* out, int K, int H) {
    const int n = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    if (n >= 4 * H || k >= K) return;
    const int tile = n >> 4, nl = n & 15;
    out[(size_t)k * 4 * H + n] = W[(size_t)k * 4 * H + (nl & 3) * H + 4 * tile + (nl >> 2)];
}
}  // namespace asr
extern "C" int asr_lstm_kernel_tile_order(void* stream, const float* W, int K, int H, float* out) {
    if (!W || !out || K <= 0 || H <= 0 || (H & 3)) return ASR_EINVAL;
    hipLaunchKernelGGL(asr::lstm_kernel_tile_order_kernel, dim3((4 * H + 255) / 256, K), dim3(256), 0, static_cast<hipStream_t>(stream), W, out, K, H);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

extern "C" size_t asr_beam_scratch_floats(int k, int Te, int H, int E, int lmP) {
    return (size_t)k * ((size_t)2 * H + E + Te + lmP);
}

namespace asr {
// row r of every state field of `out` = row sel[r] of `in` (the surviving hypotheses' parents, beam_search.py:306-318)
struct BeamGatherArgs { const float* src[7]; float* dst[7]; int width[7]; const int* sel; };
__global__ __launch_bounds__(256) void beam_gather_kernel(BeamGatherArgs a) {
    const int r = blockIdx.x, f = blockIdx.y, w = a.width[f];
    const float* s = a.src[f] + (size_t)a.sel[r] * w;
    float* d = a.dst[f] + (size_t)r * w;
    for (int i = threadIdx.x; i < w; i += 256) d[i] = s[i];
}
}  // namespace asr

extern "C" int asr_beam_gather(void* stream, const int* sel, int k, const asr_beam_state* in, const asr_beam_state* out,
                               int H, int lmH, int extH, int D) {
    if (!sel || !in || !out || k <= 0) return ASR_EINVAL;
    asr::BeamGatherArgs a;
    const float* src[7] = {in->dc, in->dh, in->dlc, in->dlh, in->lc, in->lh, in->ctx};
    float* dst[7] = {out->dc, out->dh, out->dlc, out->dlh, out->lc, out->lh, out->ctx};
    const int width[7] = {H, H, lmH, lmH, extH, extH, D};
    for (int i = 0; i < 7; ++i) { a.src[i] = src[i]; a.dst[i] = dst[i]; a.width[i] = width[i]; }
    a.sel = sel;
    hipLaunchKernelGGL(asr::beam_gather_kernel, dim3(k, 7), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Device-resident scoring / selection / bookkeeping of one beam step (beam_search.py:196-214 and :290-327), so that a
// whole utterance is decoded without a host round trip per step: float64 log-softmax of both logit vectors
// (`np.log(softmax(x))`, :196-198, :205-207), score = log p_dec + lm_weight * log p_lm + carried score (:208, :290),
// top-k per hypothesis (:214) then over the k*k continuations (:300), parent = candidate / k (:306), carried score =
// score + word_ins_penalty * len (:320-322), EOS -> finished list and k -= 1 (:323-327).  Winners are taken in
// descending score order (ties: lower candidate index), where NumPy's argpartition leaves an implementation-defined order;
// the order only matters between exactly equal float64 scores.
namespace asr {

struct BeamSelArgs {
    const float* logits; const float* logits_lm;     // [kmax][V]
    double lm_weight, wip;
    int V, kmax, eos, max_steps;
    int* ints;            // [2*kmax]  tokens | parent rows of the NEXT step's rows
    double* cum;          // [kmax]    carried scores, row order
    int* state;           // [4]       rows fed to this step, beam width left (k), finished count, step index
    int* bp;              // [max_steps][kmax][2]  (parent row, token) of the rows that leave step s
    int* fin;             // [cap][2]  (step, parent row) of finished hypotheses, in finishing order
    double* fin_score;    // [cap]
    double* cand;         // [kmax][16] scratch: top-k scores of every hypothesis
    int* cand_idx;        // [kmax][16] ... and their tokens
};

__device__ __forceinline__ void dmax_take(double& bv, int& bi, double ov, int oi) {
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)u, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)(u >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);
}
// Cross-row steps of a 64-lane butterfly on 64-bit values WITHOUT the LDS crossbar: v_permlane16_swap pairs rows 0|1 and
// 2|3, v_permlane32_swap the two half-waves.  After the four DPP row steps every lane of a row holds the row's value, so
// `other` below is what `__shfl_xor(v, 16)` / `(v, 32)` returned (the same two operands meet: bit-identical sums and
// maxima), at ~10 VALU instructions instead of four dependent ds_bpermute round trips per reduction -- the top-k loops
// below run 32 reductions per hypothesis and step and were 11 of the scoring kernel's 17 us.
__device__ __forceinline__ double other_row16_f64(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
    const auto sl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto sh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    // even rows: sl[1] holds the odd neighbour's value; odd rows: sl[0] holds the even neighbour's
    const bool odd = (__lane_id() >> 4) & 1;
    const unsigned ol = odd ? sl[0] : sl[1], oh = odd ? sh[0] : sh[1];
    return __longlong_as_double(((unsigned long long)oh << 32) | ol);
}
__device__ __forceinline__ double other_half32_f64(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
    const auto sl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto sh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const bool upper = __lane_id() >= 32;
    const unsigned ol = upper ? sl[0] : sl[1], oh = upper ? sh[0] : sh[1];
    return __longlong_as_double(((unsigned long long)oh << 32) | ol);
}
__device__ __forceinline__ double wave_max_f64(double v) {      // the butterflies of wave_allreduce_max, on 64-bit values
    v = fmax(v, dpp_mov_f64<0xB1>(v));
    v = fmax(v, dpp_mov_f64<0x4E>(v));
    v = fmax(v, dpp_mov_f64<0x141>(v));
    v = fmax(v, dpp_mov_f64<0x140>(v));
    v = fmax(v, other_row16_f64(v));
    v = fmax(v, other_half32_f64(v));
    return v;
}
// wave-wide (max value, lowest index among the lanes holding it): one 64-bit max reduction and a ballot; the index
// reduction runs only when several lanes tie (a (value, index) butterfly on every call was ~2000 cycles, 32 times per step)
__device__ __forceinline__ void wave_argmax(double& bv, int& bi) {
    const double m = wave_max_f64(bv);
    const bool mine = bv == m;
    const unsigned long long mask = __ballot(mine);
    int idx;
    if (__popcll(mask) == 1) idx = __builtin_amdgcn_readlane(bi, __ffsll((long long)mask) - 1);
    else {
        idx = mine ? bi : 0x7fffffff;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) idx = min(idx, __shfl_xor(idx, o));
    }
    bv = m; bi = idx;
}
// the xor butterfly 1, 2, 4, ..., 32 (same tree as before: at every stage the lanes of a group hold one partial sum, and
// row_half_mirror / row_mirror / the two swaps hand each lane exactly the partner group's partial)
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_mov_f64<0xB1>(v);
    v += dpp_mov_f64<0x4E>(v);
    v += dpp_mov_f64<0x141>(v);
    v += dpp_mov_f64<0x140>(v);
    v += other_row16_f64(v);
    v += other_half32_f64(v);
    return v;
}

// One workgroup per hypothesis (256 threads, 4 values per lane for V <= 1024): float64 scores of its V continuations and
// their top-k -> cand[row][k].  (All rows in ONE workgroup put ~10^5 float64 exp / div / log on a single CU: 71 us.)
// Body form: `tid` = thread within the 256 of this row, `sh` = this row's share of LDS; every thread reaches the three
// barriers whether or not its row is live (the persistent kernel runs two rows per 512-thread workgroup).  COH: common.h.
struct BeamScoreShared { float smax[2][4]; double ssum[2][4]; double cs[4][16]; int ci[4][16]; };
template <int COH>
__device__ __forceinline__ void beam_score_body(const BeamSelArgs& a, const int row, const int tid, BeamScoreShared& sh) {
    constexpr int VPL = 4, KM = 16;
    const int lane = tid & 63, wave = tid >> 6;
    // every load of the kernel is requested before the first one is looked at: the row's logits and carried score do not wait
    // for the live-row count (a row beyond it reads valid memory and is masked afterwards) -- one memory round trip, not three
    const int V = a.V;
    float x[VPL], xl[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int v = tid + 256 * j;
        x[j] = v < V ? ld_data<COH>(a.logits + (size_t)row * V + v) : -INFINITY;
        xl[j] = v < V ? ld_data<COH>(a.logits_lm + (size_t)row * V + v) : -INFINITY;
    }
    const double c0 = ld_d<COH>(a.cum + row);
    const int n_rows = ld_i<COH>(a.state), k = ld_i<COH>(a.state + 1);
    const bool act = row < n_rows && k > 0;
    float m = -INFINITY, ml = -INFINITY;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        if (!act) { x[j] = -INFINITY; xl[j] = -INFINITY; }
        m = fmaxf(m, x[j]); ml = fmaxf(ml, xl[j]);
    }
    m = wave_allreduce_max(m); ml = wave_allreduce_max(ml);
    if (lane == 0) { sh.smax[0][wave] = m; sh.smax[1][wave] = ml; }
    __syncthreads();
    m = fmaxf(fmaxf(sh.smax[0][0], sh.smax[0][1]), fmaxf(sh.smax[0][2], sh.smax[0][3]));
    ml = fmaxf(fmaxf(sh.smax[1][0], sh.smax[1][1]), fmaxf(sh.smax[1][2], sh.smax[1][3]));
    double e[VPL], el[VPL], sum = 0.0, suml = 0.0;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const bool ok = act && tid + 256 * j < V;
        e[j] = ok ? exp((double)x[j] - (double)m) : 0.0;
        el[j] = ok ? exp((double)xl[j] - (double)ml) : 0.0;
        sum += e[j]; suml += el[j];
    }
    sum = wave_sum_f64(sum); suml = wave_sum_f64(suml);
    if (lane == 0) { sh.ssum[0][wave] = sum; sh.ssum[1][wave] = suml; }
    __syncthreads();
    sum = (sh.ssum[0][0] + sh.ssum[0][1]) + (sh.ssum[0][2] + sh.ssum[0][3]);
    suml = (sh.ssum[1][0] + sh.ssum[1][1]) + (sh.ssum[1][2] + sh.ssum[1][3]);
    if (act) {
        double sc[VPL];
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const bool ok = tid + 256 * j < V;
            sc[j] = ok ? (log(e[j] / sum) + a.lm_weight * log(el[j] / suml)) + c0 : -INFINITY;
        }
        // top-k of this wave's quarter.  Each lane first sorts its four scores (descending, ties by token), so that an
        // iteration is ONE wave-wide maximum over the lanes' heads and a pop in the winning lane (before: four compare-selects to
        // find the lane's best and four compares to knock the winner out, every iteration)
        int id[VPL];
#pragma unroll
        for (int j = 0; j < VPL; ++j) id[j] = (tid + 256 * j < V) ? tid + 256 * j : 0x7fffffff;
        auto cswap = [&](int x, int y) {       // after: (sc[x], id[x]) is the better of the two
            const bool sw = sc[y] > sc[x] || (sc[y] == sc[x] && id[y] < id[x]);
            const double tv = sw ? sc[y] : sc[x], uv = sw ? sc[x] : sc[y];
            const int ti = sw ? id[y] : id[x], ui = sw ? id[x] : id[y];
            sc[x] = tv; sc[y] = uv; id[x] = ti; id[y] = ui;
        };
        cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2);
        for (int it = 0; it < k; ++it) {
            double bv = sc[0]; int bi = id[0];
            wave_argmax(bv, bi);
            if (lane == 0) { sh.cs[wave][it] = bv; sh.ci[wave][it] = bi; }
            if (id[0] == bi && bi != 0x7fffffff) {
                sc[0] = sc[1]; id[0] = id[1]; sc[1] = sc[2]; id[1] = id[2]; sc[2] = sc[3]; id[2] = id[3];
                sc[3] = -INFINITY; id[3] = 0x7fffffff;
            }
        }
    }
    __syncthreads();
    if (act && wave == 0) {
        // top-k of the hypothesis (:214): a four-way MERGE of the quarters' sorted lists on lanes 0..3 (score descending, ties by
        // token -- tokens interleave between the quarters, so the tie is broken on the token, not on the quarter)
        const int w = lane & 3;
        const bool mine = lane < 4;
        int hp = 0;
        double hv = mine ? sh.cs[w][0] : -INFINITY;
        int ht = mine ? sh.ci[w][0] : 0x7fffffff;
        for (int it = 0; it < k; ++it) {
            double m = hv;
            m = fmax(m, dpp_mov_f64<0xB1>(m)); m = fmax(m, dpp_mov_f64<0x4E>(m));          // over the quad: lanes 0..3 hold the heads
            int tm = (mine && hv == m) ? ht : 0x7fffffff;
            tm = min(tm, __builtin_amdgcn_update_dpp(0x7fffffff, tm, 0xB1, 0xF, 0xF, false));
            tm = min(tm, __builtin_amdgcn_update_dpp(0x7fffffff, tm, 0x4E, 0xF, 0xF, false));
            if (mine && hv == m && ht == tm) {
                st_d<COH>(a.cand + row * KM + it, hv); st_i<COH>(a.cand_idx + row * KM + it, ht);
                ++hp;
                hv = hp < k ? sh.cs[w][hp] : -INFINITY;
                ht = hp < k ? sh.ci[w][hp] : 0x7fffffff;
            }
        }
    }
}
__global__ __launch_bounds__(256) void beam_score_kernel(BeamSelArgs a) {
    __shared__ BeamScoreShared sh;
    beam_score_body<0>(a, blockIdx.x, threadIdx.x, sh);
}

// top-k over the n_rows * k continuations, candidate c = row * k + it (:294-306), and the bookkeeping (:306-327).  One wave.
struct BeamMergeShared { double wsc[16]; int wr[16], wv[16]; double cs[16][16]; int ct[16][16]; };
// max over the 16 lanes of a DPP row, 64-bit values (the four row steps of wave_max_f64)
__device__ __forceinline__ double row16_max_f64(double v) {
    v = fmax(v, dpp_mov_f64<0xB1>(v));
    v = fmax(v, dpp_mov_f64<0x4E>(v));
    v = fmax(v, dpp_mov_f64<0x141>(v));
    v = fmax(v, dpp_mov_f64<0x140>(v));
    return v;
}
// Every hypothesis' candidates arrive sorted (score descending, ties by token: beam_score_body emits them in that order), so the
// top-k of the n_rows * k continuations in the order (score descending, candidate c = row * k + it ascending) is a k-way MERGE of
// sorted lists: lane r of the first DPP row holds the head of list r, an iteration is one 16-lane maximum (ties: lowest row) and
// one LDS read for the winner's next head -- no division, no 64-lane reduction, no knock-out pass (round 3: the 16 serial
// wave-wide argmax iterations over 256 candidates were ~5 of the kernel's 12 us).
template <int COH>
__device__ __forceinline__ void beam_merge_body(const BeamSelArgs& a, const int lane, BeamMergeShared& sh) {
    constexpr int KM = 16;
    // all 16 x 16 slots of the candidate arrays, requested together with the counters (one memory round trip); slots beyond
    // row n_rows / rank k hold stale values and are masked
    double cvq[4]; int ctq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = lane + 64 * q, r = c >> 4, it = c & 15;
        cvq[q] = r < a.kmax ? ld_d<COH>(a.cand + r * KM + it) : -INFINITY;
        ctq[q] = r < a.kmax ? ld_i<COH>(a.cand_idx + r * KM + it) : -1;
    }
    const int n_rows = ld_i<COH>(a.state), k = ld_i<COH>(a.state + 1), s = ld_i<COH>(a.state + 3);
    if (n_rows <= 0 || k <= 0) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = lane + 64 * q, r = c >> 4, it = c & 15;
        const bool ok = r < n_rows && it < k;
        sh.cs[r][it] = ok ? cvq[q] : -INFINITY;
        sh.ct[r][it] = ok ? ctq[q] : -1;
    }
    __builtin_amdgcn_wave_barrier();
    int hp = 0;                                               // head of my list (lanes 0..15: list = lane)
    double hv = (lane < n_rows) ? sh.cs[lane & 15][0] : -INFINITY;
    for (int it = 0; it < k; ++it) {
        const double m = row16_max_f64(lane < 16 ? hv : -INFINITY);       // (lanes 16..63 idle along)
        const unsigned long long mask = __ballot(lane < 16 && hv == m && hv > -INFINITY);
        const int win = mask ? __ffsll((long long)mask) - 1 : -1;           // ties: the lowest row = the lowest candidate index
        if (lane == win) {
            sh.wsc[it] = hv; sh.wr[it] = lane; sh.wv[it] = sh.ct[lane][hp];
            ++hp;
            hv = hp < k ? sh.cs[lane][hp] : -INFINITY;
        }
        if (win < 0 && lane == 0) { sh.wsc[it] = -INFINITY; sh.wr[it] = 0; sh.wv[it] = -1; }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        int live = 0, nfin = ld_i<COH>(a.state + 2);
        for (int j = 0; j < k; ++j) {
            if (sh.wv[j] < 0 || sh.wv[j] == 0x7fffffff) continue;          // fewer candidates than k (cannot happen for V >= k)
            const double ns = s == 0 ? sh.wsc[j] : sh.wsc[j] + a.wip * (double)(s + 1);
            if (sh.wv[j] == a.eos) {
                st_i<COH>(a.fin + 2 * nfin, s); st_i<COH>(a.fin + 2 * nfin + 1, sh.wr[j]); st_d<COH>(a.fin_score + nfin, ns); ++nfin;
            } else {
                st_i<COH>(a.ints + live, sh.wv[j]); st_i<COH>(a.ints + a.kmax + live, sh.wr[j]); st_d<COH>(a.cum + live, ns);
                st_i<COH>(a.bp + ((size_t)s * a.kmax + live) * 2, sh.wr[j]); st_i<COH>(a.bp + ((size_t)s * a.kmax + live) * 2 + 1, sh.wv[j]);
                ++live;
            }
        }
        for (int j = live; j < a.kmax; ++j) { st_i<COH>(a.ints + j, 0); st_i<COH>(a.ints + a.kmax + j, 0); }
        st_i<COH>(a.state, live); st_i<COH>(a.state + 1, live); st_i<COH>(a.state + 2, nfin); st_i<COH>(a.state + 3, s + 1);
    }
}
__global__ __launch_bounds__(64) void beam_merge_kernel(BeamSelArgs a) {
    __shared__ BeamMergeShared sh;
    beam_merge_body<0>(a, threadIdx.x, sh);
}

}  // namespace asr

extern "C" int asr_beam_select(void* stream, const float* logits, const float* logits_lm, int V, int kmax, int max_steps,
                               int eos_id, double lm_weight, double word_ins_penalty, const asr_beam_book* book) {
    if (!logits || !logits_lm || !book || !book->ints || !book->cum || !book->state || !book->bp || !book->fin || !book->fin_score ||
        !book->cand || !book->cand_idx)
        return ASR_EINVAL;
    if (V <= 0 || V > 1024 || kmax <= 0 || kmax > 16 || max_steps <= 0) return ASR_EUNSUPPORTED;
    asr::BeamSelArgs a;
    a.logits = logits; a.logits_lm = logits_lm; a.lm_weight = lm_weight; a.wip = word_ins_penalty;
    a.V = V; a.kmax = kmax; a.eos = eos_id; a.max_steps = max_steps;
    a.ints = book->ints; a.cum = book->cum; a.state = book->state; a.bp = book->bp; a.fin = book->fin; a.fin_score = book->fin_score;
    a.cand = book->cand; a.cand_idx = book->cand_idx;
    hipLaunchKernelGGL(asr::beam_score_kernel, dim3(kmax), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    hipLaunchKernelGGL(asr::beam_merge_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), a);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The whole beam search of one utterance in ONE launch (beam_search.py:255-337; round 3).
//
// Per emitted token the launch path above is a chain of ~11 dependent launches (two paired skinny launches, outer cell,
// attention, two projections, scoring, merge): 80 us per token at beam 16, almost all of it launch-to-launch latency -- the
// arithmetic is 16 rows against ~10 MB of weights.  Here the SAME tile bodies (skinny_body, attention_body, the scoring and
// merge bodies: bit-identical arithmetic, tested) run as the phases of one persistent kernel of G workgroups x 512 threads;
// a phase hands its results to the next through WRITE-ONCE buffers -- a ring with one slot per token, stored agent-scope
// (write-through) and loaded with plain cached loads: no cache can hold a stale copy of a line nobody has read (COH = 2 of
// the bodies, common.h; with in-place buffers and agent-scope loads a phase took 15 us, 119 us per token, slower than the
// launch chain) -- plus a handful of agent-scope control words at fixed addresses, and a grid barrier: every thread waits for its own stores (vmcnt(0)), the workgroup meets, thread 0 adds
// one to a monotonic counter and polls it with agent-scope loads until all G have arrived.  No cache-wide write-back or
// invalidate is ever issued.  A virtual tile always runs in the same workgroup, so its slice of the weights stays in that
// XCD's L2 from token to token.  Eight phases per token:
//   1 both LM cells (lmH/4 + extH/4 tiles)   2 InputProjection | external LM's output projection   3 outer cell
//   4 attention (one workgroup per hypothesis)   5 AttnProjection   6 OutputProjection   7 float64 scoring, top-k per
//   hypothesis (two hypotheses per workgroup)   8 merge + bookkeeping (one wave) -> tokens, parents, live count
// and the loop ends when no hypothesis is live (beam_search.py:269).  All G workgroups must be co-resident (G <= 64 here,
// one per CU); a barrier that does not complete within 2 s reports err 61 and every workgroup leaves.
namespace asr {
extern unsigned long long* g_lstm_dbg;

struct BeamPersistArgs {
    asr_dec_weights w; asr_lm_weights lm; asr_dec_dims d;
    const float* hf; const float* enc; const int* enc_len;
    float* ws;                    // ring of max_steps + 1 token slots (write-once data, see beam_slot) + one alpha scratch
    long long slot_floats;
    BeamSelArgs sel;
    unsigned* bar; int* err;
    int G, max_steps;
    unsigned long long* dbg;      // diagnostic (ASR_BEAM_STAMP=1 + asr_debug_set_buffer): s_memtime per phase / barrier -> dbg[0..15], tokens -> dbg[16]
};

// Token slot t of the ring: the states that ENTER token t (written by token t-1; slot 0: zeros), then what token t itself
// produces for its later phases.  Every word of a slot is written once (agent-scope store) before anybody loads it.
struct BeamSlot { asr_beam_state st; float* x; float* p; float* logits; float* logits_lm; };
__host__ __device__ inline long long beam_slot_floats(int k, int H, int lmH, int extH, int D, int E, int V) {
    const long long n = (long long)k * (2 * H + 2 * lmH + 2 * extH + D + E + H + 2 * V);
    return (n + 31) & ~31LL;
}
__device__ __forceinline__ BeamSlot beam_slot(float* ws, long long slot_floats, int t, int k, int H, int lmH, int extH, int D, int E, int V) {
    BeamSlot o;
    float* q = ws + (size_t)t * slot_floats;
    o.st.dc = q; q += (size_t)k * H;  o.st.dh = q; q += (size_t)k * H;
    o.st.dlc = q; q += (size_t)k * lmH;  o.st.dlh = q; q += (size_t)k * lmH;
    o.st.lc = q; q += (size_t)k * extH;  o.st.lh = q; q += (size_t)k * extH;
    o.st.ctx = q; q += (size_t)k * D;
    o.x = q; q += (size_t)k * E;  o.p = q; q += (size_t)k * H;
    o.logits = q; q += (size_t)k * V;  o.logits_lm = q;
    return o;
}

__device__ __forceinline__ bool beam_grid_sync(unsigned* bar, unsigned& target, int G, int* err, int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's agent-scope stores are performed
    __syncthreads();
    target += (unsigned)G;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        long long t0 = 0;
        for (unsigned spins = 0;; ++spins) {
            if (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
            if ((spins & 255) == 255) {
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
                const long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                else if (now - t0 > 200000000LL) { __hip_atomic_store(err, 61, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
            }
        }
        *flag = ok;
    }
    __syncthreads();
    return *flag != 0;
}

__global__ __launch_bounds__(512) void beam_persist_kernel(BeamPersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // attention_body's dynamic LDS
    __shared__ float red[2][3][4][64];
    __shared__ float wred[16];
    __shared__ BeamScoreShared ssh[2];
    __shared__ BeamMergeShared msh;
    __shared__ int sflag;
    const int tid = threadIdx.x, half = tid >> 8, t256 = tid & 255;
    const int G = a.G, wg = blockIdx.x;
    const int k = a.d.B, Te = a.d.Te, D = a.d.D, A = a.d.A, H = a.d.H, lmH = a.d.lmH, E = a.d.E, V = a.d.V;
    const int extH = a.lm.H;
    float* alpha = a.ws + (size_t)(a.max_steps + 1) * a.slot_floats;      // never read back: one fixed scratch
    const int* tokens = a.sel.ints;
    unsigned target = 0;
    unsigned long long tl = a.dbg ? __builtin_amdgcn_s_memtime() : 0;
    unsigned int stamp[16] = {0};
    int sidx = 0;
#define BEAM_STAMP() if (a.dbg) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); stamp[sidx & 15] += (unsigned int)(t__ - tl); tl = t__; ++sidx; }
#define BEAM_SYNC() { BEAM_STAMP() const bool ok__ = beam_grid_sync(a.bar, target, G, a.err, &sflag); BEAM_STAMP() if (!ok__) return; }

    // two tiles of one or two skinny problems side by side in the two halves of the workgroup
    auto run = [&](auto lstm_tag, const SkinnyArgs& a0, const int nb0, const SkinnyArgs* a1, const int nb1) {
        constexpr bool LSTM = decltype(lstm_tag)::value;
        const int nvb = nb0 + nb1;
        for (int base = 2 * wg; base < nvb; base += 2 * G) {
            const int vb = base + half;
            if (vb < nb0) skinny_body<LSTM, false, 2>(a0, vb, 0, red[half], t256);
            else if (vb < nvb) skinny_body<LSTM, false, 2>(*a1, vb - nb0, 0, red[half], t256);
            else __syncthreads();                      // the body's one barrier
            __syncthreads();                           // red[] is reused by the next tile
        }
    };
    using T = std::true_type;
    using F = std::false_type;

    for (int s = 0; s < a.max_steps; ++s) {
        const BeamSlot cur = beam_slot(a.ws, a.slot_floats, s, k, H, lmH, extH, D, E, V);
        const BeamSlot nxt = beam_slot(a.ws, a.slot_floats, s + 1, k, H, lmH, extH, D, E, V);
        const asr_beam_state& in = cur.st;
        const asr_beam_state& out = nxt.st;
        float* x = cur.x; float* p = cur.p;
        const int* sel = s ? a.sel.ints + a.sel.kmax : nullptr;
        // ---- 1: decoder's inner LM cell (:183-186) | external LM cell (:200-203)
        {
            SkinnyArgs c0{}, c1{};
            c0.x1 = a.w.embedding; c0.ld1 = E; c0.K1 = E; c0.gather1 = tokens; c0.x2 = in.dlh; c0.ld2 = lmH; c0.K2 = lmH; c0.gather2 = sel;
            c0.W = a.w.lm_kernel; c0.ldw = 4 * lmH; c0.bias = a.w.lm_bias; c0.M = k; c0.N = 4 * lmH; c0.H = lmH;
            c0.c_prev = in.dlc; c0.c_out = out.dlc; c0.h_out = out.dlh; c0.keep = 1.0f;
            c1.x1 = a.lm.embedding; c1.ld1 = a.lm.E; c1.K1 = a.lm.E; c1.gather1 = tokens; c1.x2 = in.lh; c1.ld2 = a.lm.H; c1.K2 = a.lm.H; c1.gather2 = sel;
            c1.W = a.lm.lstm_kernel; c1.ldw = 4 * a.lm.H; c1.bias = a.lm.lstm_bias; c1.M = k; c1.N = 4 * a.lm.H; c1.H = a.lm.H;
            c1.c_prev = in.lc; c1.c_out = out.lc; c1.h_out = out.lh; c1.keep = 1.0f;
            run(T{}, c0, (lmH + 3) / 4, &c1, (a.lm.H + 3) / 4);
        }
        BEAM_SYNC()
        // ---- 2: x = [lm_out, ctx_prev] . W_inp + b (:188-189) | external LM's logits (:204-207)
        {
            SkinnyArgs p0{}, p1{};
            p0.x1 = out.dlh; p0.ld1 = lmH; p0.K1 = lmH; p0.x2 = in.ctx; p0.ld2 = D; p0.K2 = D; p0.gather2 = sel;
            p0.W = a.w.inp_w; p0.ldw = E; p0.bias = a.w.inp_b; p0.M = k; p0.N = E; p0.out = x; p0.ldo = E;
            p1.x1 = out.lh; p1.ld1 = a.lm.H; p1.K1 = a.lm.H; p1.W = a.lm.out_w; p1.ldw = a.lm.V; p1.bias = a.lm.out_b;
            p1.M = k; p1.N = a.lm.V; p1.out = cur.logits_lm; p1.ldo = a.lm.V;
            run(F{}, p0, (E + 15) / 16, &p1, (a.lm.V + 15) / 16);
        }
        BEAM_SYNC()
        // ---- 3: outer cell (:190-191)
        {
            SkinnyArgs oc{};
            oc.x1 = x; oc.ld1 = E; oc.K1 = E; oc.x2 = in.dh; oc.ld2 = H; oc.K2 = H; oc.gather2 = sel;
            oc.W = a.w.dec_kernel; oc.ldw = 4 * H; oc.bias = a.w.dec_bias; oc.M = k; oc.N = 4 * H; oc.H = H;
            oc.c_prev = in.dc; oc.c_out = out.dc; oc.h_out = out.dh; oc.keep = 1.0f;
            run(T{}, oc, (H + 3) / 4, nullptr, 0);
        }
        BEAM_SYNC()
        // ---- 4: attention with query = c (:193), one workgroup per hypothesis
        {
            AttnArgs at{};
            at.q = out.dc; at.ldq = H; at.w_att = a.w.attn_w; at.b_att = a.w.attn_b; at.v = a.w.attn_v; at.hf = a.hf; at.enc = a.enc;
            at.enc_len = a.enc_len; at.alpha = alpha; at.ctx = out.ctx; at.B = k; at.Te = Te; at.H = H; at.A = A; at.D = D;
            at.len_shared = 1; at.y_out = nullptr; at.hf_bs = 0; at.enc_bs = 0;
            for (int b = wg; b < k; b += G) { attention_body<2>(at, b, smem, wred); __syncthreads(); }
        }
        BEAM_SYNC()
        // ---- 5: AttnProjection (:194-196)
        {
            SkinnyArgs ap{};
            ap.x1 = out.dc; ap.ld1 = H; ap.K1 = H; ap.x2 = out.ctx; ap.ld2 = D; ap.K2 = D;
            ap.W = a.w.ap_w; ap.ldw = H; ap.bias = a.w.ap_b; ap.M = k; ap.N = H; ap.out = p; ap.ldo = H;
            run(F{}, ap, (H + 15) / 16, nullptr, 0);
        }
        BEAM_SYNC()
        // ---- 6: OutputProjection (:197-198)
        {
            SkinnyArgs op{};
            op.x1 = p; op.ld1 = H; op.K1 = H; op.W = a.w.out_w; op.ldw = V; op.bias = a.w.out_b; op.M = k; op.N = V;
            op.out = cur.logits; op.ldo = V;
            run(F{}, op, (V + 15) / 16, nullptr, 0);
        }
        BEAM_SYNC()
        // ---- 7: float64 scores and the top-k of every hypothesis (:196-214)
        {
            BeamSelArgs sa = a.sel;
            sa.logits = cur.logits; sa.logits_lm = cur.logits_lm;
            for (int base = 2 * wg; base < k; base += 2 * G) beam_score_body<2>(sa, base + half, t256, ssh[half]);
        }
        BEAM_SYNC()
        // ---- 8: top-k over the continuations, bookkeeping (:290-327)
        if (wg == 0 && tid < 64) beam_merge_body<2>(a.sel, tid, msh);
        BEAM_SYNC()
        sidx = 0;
        if (a.dbg && wg == 0 && tid == 0) a.dbg[16] += 1;
        if (ld_i<2>(a.sel.state + 1) <= 0) break;        // every hypothesis finished (:269)
    }
    if (a.dbg && wg == 0 && tid == 0) for (int i = 0; i < 16; ++i) a.dbg[i] += stamp[i];
#undef BEAM_STAMP
#undef BEAM_SYNC
}

}  // namespace asr

extern "C" size_t asr_attention_lds_bytes(int Te, int H, int A);
int asr_lstm_max_wgs();

extern "C" size_t asr_beam_decode_ws_floats(const asr_dec_dims* d, int extH, int max_steps) {
    if (!d || max_steps <= 0) return 0;
    return (size_t)(max_steps + 1) * (size_t)asr::beam_slot_floats(d->B, d->H, d->lmH, extH, d->D, d->E, d->V) + (size_t)d->B * d->Te;
}

// Whole-utterance beam search in one launch.  Initial conditions as for the step-by-step calls: d->B = beam, book->state =
// {1, beam, 0, 0}, book->cum = 0, book->ints[0] = GO; `ws`: asr_beam_decode_ws_floats floats (its first slot is zeroed
// here: the initial states); `barrier`: one zeroed 32-bit word.  On return the book holds what the loop of
// asr_beam_step_sel + asr_beam_select would have left in it (bit-identical: the same tile bodies).
// ASR_EUNSUPPORTED (use the step calls): SimpleProjections present, V > 1024, beam > 16, vocabularies differ.
extern "C" int asr_beam_decode(void* stream, const asr_dec_weights* w, const asr_lm_weights* lm, const asr_dec_dims* d,
                               const float* hf, const float* enc, const int* enc_len, float* ws, size_t ws_floats,
                               int max_steps, int eos_id, double lm_weight, double word_ins_penalty,
                               const asr_beam_book* book, unsigned* barrier, int* err_flag) {
    if (!w || !lm || !d || !hf || !enc || !enc_len || !ws || !book || !barrier || !err_flag) return ASR_EINVAL;
    if (!book->ints || !book->cum || !book->state || !book->bp || !book->fin || !book->fin_score || !book->cand || !book->cand_idx)
        return ASR_EINVAL;
    if (w->simple_w || lm->simple_w) return ASR_EUNSUPPORTED;
    const int kmax = d->B;
    if (kmax <= 0 || kmax > 16 || d->V <= 0 || d->V > 1024 || lm->V != d->V || max_steps <= 0) return ASR_EUNSUPPORTED;
    if ((d->H & 3) || (d->lmH & 3) || (lm->H & 3) || (d->E & 3) || (lm->E & 3) || (d->D & 3) || (d->A & 3) || d->A > 1024) return ASR_EUNSUPPORTED;
    if (ws_floats < asr_beam_decode_ws_floats(d, lm->H, max_steps) || (reinterpret_cast<uintptr_t>(ws) & 127)) return ASR_EINVAL;
    const size_t lds = asr_attention_lds_bytes(d->Te, d->H, d->A);
    if (lds > 120 * 1024) return ASR_EUNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    asr::BeamPersistArgs a{};
    a.w = *w; a.lm = *lm; a.d = *d;
    a.hf = hf; a.enc = enc; a.enc_len = enc_len; a.ws = ws;
    a.slot_floats = asr::beam_slot_floats(kmax, d->H, d->lmH, lm->H, d->D, d->E, d->V);
    a.sel.logits = nullptr; a.sel.logits_lm = nullptr; a.sel.lm_weight = lm_weight; a.sel.wip = word_ins_penalty;
    a.sel.V = d->V; a.sel.kmax = kmax; a.sel.eos = eos_id; a.sel.max_steps = max_steps;
    a.sel.ints = book->ints; a.sel.cum = book->cum; a.sel.state = book->state; a.sel.bp = book->bp; a.sel.fin = book->fin;
    a.sel.fin_score = book->fin_score; a.sel.cand = book->cand; a.sel.cand_idx = book->cand_idx;
    a.bar = barrier; a.err = err_flag; a.max_steps = max_steps;
    a.dbg = getenv("ASR_BEAM_STAMP") ? asr::g_lstm_dbg : nullptr;
    // widest phase: the two LM cells, two tiles per workgroup
    int G = ((d->lmH + 3) / 4 + (lm->H + 3) / 4 + 1) / 2;
    if (G > 64) G = 64;
    if (G < kmax) G = kmax;
    if (G > asr_lstm_max_wgs()) return ASR_EUNSUPPORTED;      // every workgroup must be resident: the co-residency budget of csrc/lstm.hip
    a.G = G;
    // initial states: the state part of slot 0
    const size_t st0 = (size_t)kmax * (2 * d->H + 2 * d->lmH + 2 * lm->H + d->D) * sizeof(float);
    if (hipMemsetAsync(ws, 0, st0, s) != hipSuccess) return ASR_ELAUNCH;
    hipLaunchKernelGGL(asr::beam_persist_kernel, dim3(G), dim3(512), lds, s, a);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
