// One beam-search step for all live hypotheses of ONE utterance -- the arithmetic of BeamSearch.get_top_k
// (beam_search.py:163-221) up to the two logit vectors: decoder LM cell on the fed token, [SimpleProjection],
// InputProjection with the previous context, outer cell, attention over the shared encoder states (query = cell
// state c), AttnProjection, OutputProjection; and the external LM cell + its output projection for shallow fusion
// (:200-207).  Scoring (float64 log-softmax, lm_weight, argpartition) stays on the host as the reference defines it.
// Ten stream-ordered launches of the step kernels behind ONE call (skinny.hip, attention.hip).
#include "common.h"
#include "../../include/e2e_asr_hip.h"

extern "C" int asr_beam_step(void* stream, const asr_dec_weights* w, const asr_lm_weights* lm, const asr_dec_dims* d,
                             const float* hf, const float* enc, const int* enc_len, const int* tokens,
                             const asr_beam_state* in, const asr_beam_state* out, float* scratch,
                             float* logits, float* logits_lm) {
    if (!w || !lm || !d || !hf || !enc || !enc_len || !tokens || !in || !out || !scratch || !logits || !logits_lm) return ASR_EINVAL;
    const int k = d->B, Te = d->Te, D = d->D, A = d->A, H = d->H, lmH = d->lmH, E = d->E, V = d->V;
    if (k <= 0) return ASR_EINVAL;
    int rc;
    float* sp = scratch;                              // [k, H]   SimpleProjection output
    float* x = sp + (size_t)k * H;                    // [k, E]
    float* p = x + (size_t)k * E;                     // [k, H]
    float* alpha = p + (size_t)k * H;                 // [k, Te]
    float* lsp = alpha + (size_t)k * Te;              // [k, lm->P] external LM SimpleProjection output
    // decoder's inner LM cell on emb[token]  (:183-186)
    if ((rc = asr_lstm_cell_fwd(stream, w->embedding, E, E, tokens, in->dlh, in->dlc, w->lm_kernel, w->lm_bias, lmH, k,
                                out->dlc, out->dlh, nullptr, nullptr, 1.0f, 0, 0))) return rc;
    const float* o = out->dlh; int P = lmH;
    if (w->simple_w) {
        if ((rc = asr_linear_fwd(stream, o, lmH, lmH, nullptr, nullptr, 0, 0, w->simple_w, H, w->simple_b, sp, H, k, H, nullptr, 0))) return rc;
        o = sp; P = H;
    }
    // x = [lm_out, ctx_prev] . W_inp + b  (:188-189), outer cell (:190-191)
    if ((rc = asr_linear_fwd(stream, o, P, P, nullptr, in->ctx, D, D, w->inp_w, E, w->inp_b, x, E, k, E, nullptr, 0))) return rc;
    if ((rc = asr_lstm_cell_fwd(stream, x, E, E, nullptr, in->dh, in->dc, w->dec_kernel, w->dec_bias, H, k,
                                out->dc, out->dh, nullptr, nullptr, 1.0f, 0, 0))) return rc;
    // attention with query = c (:193), AttnProjection, OutputProjection (:194-198)
    if ((rc = asr_attention_shared_fwd(stream, out->dc, H, w->attn_w, w->attn_b, w->attn_v, hf, enc, enc_len, alpha, out->ctx,
                                       k, Te, H, A, D, 1))) return rc;
    if ((rc = asr_linear_fwd(stream, out->dc, H, H, nullptr, out->ctx, D, D, w->ap_w, H, w->ap_b, p, H, k, H, nullptr, 0))) return rc;
    if ((rc = asr_linear_fwd(stream, p, H, H, nullptr, nullptr, 0, 0, w->out_w, V, w->out_b, logits, V, k, V, nullptr, 0))) return rc;
    // external LM (:200-207)
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
