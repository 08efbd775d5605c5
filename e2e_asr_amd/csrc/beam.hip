// One beam-search step for all live hypotheses of ONE utterance -- the arithmetic of BeamSearch.get_top_k
// (beam_search.py:163-221) up to the two logit vectors: decoder LM cell on the fed token, [SimpleProjection],
// InputProjection with the previous context, outer cell, attention over the shared encoder states (query = cell
// state c), AttnProjection, OutputProjection; and the external LM cell + its output projection for shallow fusion
// (:200-207).  Scoring (float64 log-softmax, lm_weight, argpartition) stays on the host as the reference defines it.
// Ten stream-ordered launches of the step kernels behind ONE call (skinny.hip, attention.hip).
#include "common.h"
#include "skinny.h"
#include "../../include/e2e_asr_hip.h"

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
        c0.W = w->lm_kernel; c0.ldw = 4 * lmH; c0.bias = w->lm_bias; c0.M = k; c0.N = 4 * lmH; c0.H = lmH;
        c0.c_prev = in->dlc; c0.c_out = out->dlc; c0.h_out = out->dlh; c0.keep = 1.0f;
        c1.x1 = lm->embedding; c1.ld1 = lm->E; c1.K1 = lm->E; c1.gather1 = tokens; c1.x2 = in->lh; c1.ld2 = lm->H; c1.K2 = lm->H; c1.gather2 = sel;
        c1.W = lm->lstm_kernel; c1.ldw = 4 * lm->H; c1.bias = lm->lstm_bias; c1.M = k; c1.N = 4 * lm->H; c1.H = lm->H;
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
        oc.W = w->dec_kernel; oc.ldw = 4 * H; oc.bias = w->dec_bias; oc.M = k; oc.N = 4 * H; oc.H = H;
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
__device__ __forceinline__ double wave_max_f64(double v) {      // the butterflies of wave_allreduce_max, on 64-bit values
    v = fmax(v, dpp_mov_f64<0xB1>(v));
    v = fmax(v, dpp_mov_f64<0x4E>(v));
    v = fmax(v, dpp_mov_f64<0x141>(v));
    v = fmax(v, dpp_mov_f64<0x140>(v));
    v = fmax(v, __shfl_xor(v, 16));
    v = fmax(v, __shfl_xor(v, 32));
    return v;
}
// wave-wide (max value, lowest index among the lanes holding it): one 64-bit max reduction and a ballot; the index
// reduction runs only when several lanes tie (a (value, index) butterfly on every call was ~2000 cycles, 32 times per step)
__device__ __forceinline__ void wave_argmax(double& bv, int& bi) {
    const double m = wave_max_f64(bv);
    const bool mine = bv == m;
    const unsigned long long mask = __ballot(mine);
    int idx;
    if (__popcll(mask) == 1) idx = __shfl(bi, __ffsll((long long)mask) - 1);
    else {
        idx = mine ? bi : 0x7fffffff;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) idx = min(idx, __shfl_xor(idx, o));
    }
    bv = m; bi = idx;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

// One workgroup per hypothesis (256 threads, 4 values per lane for V <= 1024): float64 scores of its V continuations and
// their top-k -> cand[row][k].  (All rows in ONE workgroup put ~10^5 float64 exp / div / log on a single CU: 71 us.)
__global__ __launch_bounds__(256) void beam_score_kernel(BeamSelArgs a) {
    constexpr int VPL = 4, KM = 16;
    __shared__ float smax[2][4];
    __shared__ double ssum[2][4];
    __shared__ double cs[4][KM];
    __shared__ int ci[4][KM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = blockIdx.x;
    const int n_rows = a.state[0], k = a.state[1];
    if (row >= n_rows || k <= 0) return;
    const int V = a.V;
    float x[VPL], xl[VPL];
    float m = -INFINITY, ml = -INFINITY;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int v = tid + 256 * j;
        x[j] = v < V ? a.logits[(size_t)row * V + v] : -INFINITY;
        xl[j] = v < V ? a.logits_lm[(size_t)row * V + v] : -INFINITY;
        m = fmaxf(m, x[j]); ml = fmaxf(ml, xl[j]);
    }
    m = wave_allreduce_max(m); ml = wave_allreduce_max(ml);
    if (lane == 0) { smax[0][wave] = m; smax[1][wave] = ml; }
    __syncthreads();
    m = fmaxf(fmaxf(smax[0][0], smax[0][1]), fmaxf(smax[0][2], smax[0][3]));
    ml = fmaxf(fmaxf(smax[1][0], smax[1][1]), fmaxf(smax[1][2], smax[1][3]));
    double e[VPL], el[VPL], sum = 0.0, suml = 0.0;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const bool ok = tid + 256 * j < V;
        e[j] = ok ? exp((double)x[j] - (double)m) : 0.0;
        el[j] = ok ? exp((double)xl[j] - (double)ml) : 0.0;
        sum += e[j]; suml += el[j];
    }
    sum = wave_sum_f64(sum); suml = wave_sum_f64(suml);
    if (lane == 0) { ssum[0][wave] = sum; ssum[1][wave] = suml; }
    __syncthreads();
    sum = (ssum[0][0] + ssum[0][1]) + (ssum[0][2] + ssum[0][3]);
    suml = (ssum[1][0] + ssum[1][1]) + (ssum[1][2] + ssum[1][3]);
    const double c0 = a.cum[row];
    double sc[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const bool ok = tid + 256 * j < V;
        sc[j] = ok ? (log(e[j] / sum) + a.lm_weight * log(el[j] / suml)) + c0 : -INFINITY;
    }
    for (int it = 0; it < k; ++it) {           // top-k of this wave's quarter
        double bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < VPL; ++j) dmax_take(bv, bi, sc[j], tid + 256 * j);
        wave_argmax(bv, bi);
        if (lane == 0) { cs[wave][it] = bv; ci[wave][it] = bi; }
#pragma unroll
        for (int j = 0; j < VPL; ++j) if (tid + 256 * j == bi) sc[j] = -INFINITY;
    }
    __syncthreads();
    if (wave == 0) {                           // top-k of the hypothesis (:214) from the 4 * k quarter winners
        const bool ok = lane < 4 * k;
        double cv = ok ? cs[lane / k][lane % k] : -INFINITY;
        const int cx = ok ? ci[lane / k][lane % k] : 0x7fffffff;
        for (int it = 0; it < k; ++it) {
            double bv = cv; int bi = cx;
            wave_argmax(bv, bi);
            if (lane == 0) { a.cand[row * KM + it] = bv; a.cand_idx[row * KM + it] = bi; }
            if (cx == bi) cv = -INFINITY;
        }
    }
}

// top-k over the n_rows * k continuations, candidate c = row * k + it (:294-306), and the bookkeeping (:306-327)
__global__ __launch_bounds__(64) void beam_merge_kernel(BeamSelArgs a) {
    constexpr int KM = 16;
    __shared__ double wsc[KM];
    __shared__ int wr[KM], wv[KM];
    const int lane = threadIdx.x;
    const int n_rows = a.state[0], k = a.state[1], s = a.state[3];
    if (n_rows <= 0 || k <= 0) return;
    double cv[4]; int cc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = lane + 64 * q;
        const bool ok = c < n_rows * k;
        cv[q] = ok ? a.cand[(c / k) * KM + c % k] : -INFINITY;
        cc[q] = ok ? c : 0x7fffffff;
    }
    for (int it = 0; it < k; ++it) {
        double bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < 4; ++q) dmax_take(bv, bi, cv[q], cc[q]);
        wave_argmax(bv, bi);
        if (lane == 0) {
            const bool ok = bi != 0x7fffffff;
            wsc[it] = bv; wr[it] = ok ? bi / k : 0; wv[it] = ok ? a.cand_idx[(bi / k) * KM + bi % k] : -1;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) if (cc[q] == bi) cv[q] = -INFINITY;
    }
    if (lane == 0) {
        int live = 0, nfin = a.state[2];
        for (int j = 0; j < k; ++j) {
            if (wv[j] < 0 || wv[j] == 0x7fffffff) continue;                // fewer candidates than k (cannot happen for V >= k)
            const double ns = s == 0 ? wsc[j] : wsc[j] + a.wip * (double)(s + 1);
            if (wv[j] == a.eos) {
                a.fin[2 * nfin] = s; a.fin[2 * nfin + 1] = wr[j]; a.fin_score[nfin] = ns; ++nfin;
            } else {
                a.ints[live] = wv[j]; a.ints[a.kmax + live] = wr[j]; a.cum[live] = ns;
                a.bp[((size_t)s * a.kmax + live) * 2] = wr[j]; a.bp[((size_t)s * a.kmax + live) * 2 + 1] = wv[j];
                ++live;
            }
        }
        for (int j = live; j < a.kmax; ++j) { a.ints[j] = 0; a.ints[a.kmax + j] = 0; }
        a.state[0] = live; a.state[1] = live; a.state[2] = nfin; a.state[3] = s + 1;
    }
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
