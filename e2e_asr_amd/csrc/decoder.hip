// Attention decoder, whole-sequence forward: the raw_rnn loop of attn_decoder.py:37-172
// unrolled on the host into stream-ordered launches of the step kernels (skinny.hip,
// attention.hip, loss.hip).  No allocation, no synchronisation: the caller can capture the
// whole call in a hipGraph.
//
// Step structure (see oracle/asr_oracle.py::attn_decoder for the restated semantics):
//   i = 0..T_out-1:
//     lm cell on emb[tok[i]]                 (:148)   -> lm_c[i], lm_h[i]
//     [SimpleProjection]                     (:149-151)
//     x[i] = [lm_out, ctx[i-1]].W_inp + b    (:157-158)
//     outer cell on x[i]                     (:166)   -> dec_c[i], dec_h[i]
//     attention(q = dec_c[i])                (:114, decoder.py:79-80) -> alpha[i], ctx[i]
//     p[i] = [q, ctx[i]].W_ap + b            (:116-118)
//     logits[i] = p[i].W_out + b             (:124-125), zero rows once i >= seq_len[b]
//     tok[i+1] = teacher / argmax / sample   (:128-145)
// raw_rnn's state copy-through for finished rows is not materialised: a finished row's
// emit is zero and its loss weight is zero, so its state is unobservable.
#include "common.h"
#include <cstdlib>
#include "../../include/e2e_asr_hip.h"

extern "C" int asr_zero_finished_rows(void* stream, float* logits, const int* len, int T, int B, int V);
extern "C" int asr_decoder_chain_supported(int B, int Te, int D, int A, int H);
extern "C" int asr_decoder_greedy_supported(int B, int Te, int D, int A, int H, int lmH, int E, int V);
extern "C" int asr_decoder_greedy_fwd(void* stream, const float* embedding, const float* lm_kernel, const float* lm_bias,
                                      const float* wk, const float* bprime, const float* dec_kh, const float* w_att,
                                      const float* b_att, const float* v, const float* ap_w, const float* ap_b,
                                      const float* out_w, const float* out_b, const float* hf, const float* enc,
                                      const int* enc_len, const int* seq_len, int* tok, float* logits, void* ws, int* err,
                                      int B, int Te, int D, int A, int H, int lmH, int E, int V, int T);
int asr_decoder_chain_fwd(void* stream, float* gates, const float* wh, const float* wc, const float* w_att,
                          const float* b_att, const float* v, const float* hf, const float* enc, const int* enc_len,
                          float* dec_c, float* dec_h, float* alpha, float* ctx, float* y, void* ws, int* err,
                          int B, int Te, int D, int A, int H, int t0, int t1);
int asr_decoder_train_fwd(void* stream, const float* embedding, const float* lm_kernel, const float* lm_bias,
                          const float* wk, const float* bprime, const float* dec_kh, const float* w_att,
                          const float* b_att, const float* v, const float* ap_w, const float* ap_b,
                          const float* out_w, const float* out_b, const float* hf, const float* enc,
                          const int* enc_len, const int* seq_len, int* tok, const unsigned* fbmask8, float keep, unsigned seed,
                          float* lm_out, float* lm_hprev, float* lm_act, float* dec_gates, float* dec_c, float* dec_h,
                          float* y, float* alpha, float* ctx, void* ws, int* err,
                          int B, int Te, int D, int A, int H, int lmH, int E, int V, int T);
bool asr_lstm_tm_supported(int B, int H);
int asr_lstm_rec_fwd_tm(hipStream_t s, const float* gates, const float* kh, const int* full_len, float* out, int ldo,
                        float* act, float* hprev, const float* h0, const float* c0, float* h_last, float* c_last,
                        void* hx_ws, int* err, int B, int T, int H, int toff, float keep, unsigned seed);
extern "C" int asr_gather_rows(void*, const float*, const int*, float*, int, int);
extern "C" int asr_decoder_lm_chain_supported(int B, int lmH) {
    if (getenv("ASR_LM_CHAIN") && atoi(getenv("ASR_LM_CHAIN")) == 0) return 0;
    return asr_lstm_tm_supported(B, lmH) ? 1 : 0;
}
int asr_attention_launch(void* stream, const float* q, int ldq, const float* w_att, const float* b_att, const float* v,
                         const float* hf, const float* enc, const int* enc_len, float* alpha, float* ctx, float* y_out,
                         int B, int Te, int H, int A, int D, int shared);

// Two streams.  The LM cell chain (attn_decoder.py:148-151) depends only on the fed tokens, so it
// runs on the library's side stream AHEAD of the attention chain and re-synchronises only after a
// step whose token is produced on the device (argmax / sampled feedback).  AttnProjection and
// OutputProjection (:116-125) are not on the per-step dependency chain either: outside feedback
// steps they are computed for all steps at once by MFMA GEMMs after the loop.  Per step the main
// stream runs three kernels: InputProjection, outer cell, fused attention.
extern "C" int asr_attn_decoder_fwd(void* stream, const asr_dec_weights* w, const asr_dec_dims* d,
                                    const asr_dec_ws* ws, const float* enc, const int* enc_len,
                                    const int* seq_len, int mode, const float* coin_host, float samp_prob,
                                    float keep_lm, unsigned seed, float* logits) {
    if (!w || !d || !ws || !enc || !enc_len || !seq_len || !logits) return ASR_EINVAL;
    if (mode < 0 || mode > 2 || (mode == 2 && !coin_host)) return ASR_EINVAL;
    const int B = d->B, Te = d->Te, D = d->D, A = d->A, H = d->H, lmH = d->lmH, E = d->E, V = d->V, T = d->T_out;
    if (B <= 0 || T <= 0) return ASR_EINVAL;
    if (keep_lm < 1.0f && !ws->lm_hd) return ASR_EINVAL;
    if (w->simple_w && !ws->sp) return ASR_EINVAL;
    hipStream_t ms = static_cast<hipStream_t>(stream);
    hipStream_t ss = asr::side_stream();
    void* side = static_cast<void*>(ss);
    int rc;
    asr::prof_begin(ASR_PROF_DECODER_FWD, ms);
    hipEvent_t e_fork = asr::next_event();
    if (hipEventRecord(e_fork, ms) != hipSuccess || hipStreamWaitEvent(ss, e_fork, 0) != hipSuccess) return ASR_ELAUNCH;
    // hf = enc . AttnW   (attn_decoder.py:70-73)
    if ((rc = asr_gemm_f32(stream, 0, 0, B * Te, A, D, enc, D, w->attn_enc_w, A, ws->hf, A, nullptr, 0))) return rc;
    const int P = w->simple_w ? H : lmH;
    // WK = W_inp . K_x and b' = b_inp . K_x + b_dec of the one-launch decoders: weights only -- on the side stream, next to the hf
    // product above (three small launches in a row are three launch latencies; side by side they are one)
    // (ASR_DEC_SIDE_SMALL=0: everything on the caller's stream, as before round 3's last day)
    static const bool side_small = [] { const char* e = getenv("ASR_DEC_SIDE_SMALL"); return !(e && e[0] == '0'); }();
    void* small_s = side_small ? side : stream;
    auto fold_input_projection = [&](float* wk, float* bprime) -> int {
        int r;
        if ((r = asr_gemm_f32(small_s, 0, 0, lmH + D, 4 * H, E, w->inp_w, E, w->dec_kernel, 4 * H, wk, 4 * H, nullptr, 0))) return r;
        if ((r = asr_gemm_f32(small_s, 0, 0, 1, 4 * H, E, w->inp_b, E, w->dec_kernel, 4 * H, bprime, 4 * H, w->dec_bias, 0))) return r;
        if (side_small) {
            hipEvent_t e_wk = asr::next_event();
            if (hipEventRecord(e_wk, ss) != hipSuccess || hipStreamWaitEvent(ms, e_wk, 0) != hipSuccess) return ASR_ELAUNCH;
        }
        return ASR_OK;
    };
    hipEvent_t e_x = nullptr;          // the saved InputProjection output of the training graph (side stream, joined at the end)
    auto feedback = [&](int i) {       // is tok[i+1] produced on the device from step i's logits?
        if (i < 0 || i + 1 >= T) return false;
        if (mode == 1) return true;
        return mode == 2 && samp_prob > 0.f && !(coin_host[i] < 1.0f - samp_prob);
    };
    hipEvent_t e_tok = nullptr;
    // ---- inference graph (greedy feedback at every step): the whole loop in one launch of csrc/decoder_greedy.hip
    if (mode == 1 && ws->greedy_ws && ws->w2k && ws->err && !w->simple_w && keep_lm >= 1.0f &&
        asr_decoder_greedy_supported(B, Te, D, A, H, lmH, E, V)) {
        float* wk = ws->w2k;                       // [(lmH+D), 4H] followed by b' [4H] (InputProjection folded, as below)
        float* bprime = wk + (size_t)(lmH + D) * 4 * H;
        if ((rc = fold_input_projection(wk, bprime))) return rc;
        if ((rc = asr_decoder_greedy_fwd(stream, w->embedding, w->lm_kernel, w->lm_bias, wk, bprime, w->dec_kernel + (size_t)E * 4 * H,
                                         w->attn_w, w->attn_b, w->attn_v, w->ap_w, w->ap_b, w->out_w, w->out_b, ws->hf, enc, enc_len,
                                         seq_len, ws->tok, logits, ws->greedy_ws, ws->err, B, Te, D, A, H, lmH, E, V, T))) return rc;
        asr::prof_end(ASR_PROF_DECODER_FWD, ms);
        return ASR_OK;
    }
    // ---- training graph in ONE persistent launch (csrc/decoder_greedy.hip, TRAIN instantiation): LM cell, outer cell and
    // attention of every step, AttnProjection / OutputProjection / Gumbel-max draw at the scheduled-sampling feedback steps
    // (host coins -> a 256-bit mask passed by value), LM dropout, activations saved in the layouts the backward reads.  No
    // per-segment launches; ASR_DEC_TRAINK=0 selects the segment-wise chain path below (the path it is tested against).
    static const bool traink_env = [] { const char* e = getenv("ASR_DEC_TRAINK"); return !(e && e[0] == '0'); }();
    bool done_train_kernel = false;
    if (traink_env && mode != 1 && T <= 256 && ws->greedy_ws && ws->w2k && ws->err && ws->y && ws->dec_gates && !w->simple_w &&
        ws->lm_act && ws->lm_hprev && (keep_lm >= 1.0f || ws->lm_hd) && asr_decoder_greedy_supported(B, Te, D, A, H, lmH, E, V)) {
        float* wk = ws->w2k;                       // [(lmH+D), 4H] followed by b' [4H] (InputProjection folded, as below)
        float* bprime = wk + (size_t)(lmH + D) * 4 * H;
        if ((rc = fold_input_projection(wk, bprime))) return rc;
        unsigned fbmask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < T; ++i)
            if (feedback(i)) fbmask[i >> 5] |= 1u << (i & 31);
        float* lm_out_buf = keep_lm < 1.0f ? ws->lm_hd : ws->lm_h;
        if ((rc = asr_decoder_train_fwd(stream, w->embedding, w->lm_kernel, w->lm_bias, wk, bprime, w->dec_kernel + (size_t)E * 4 * H,
                                        w->attn_w, w->attn_b, w->attn_v, w->ap_w, w->ap_b, w->out_w, w->out_b, ws->hf, enc, enc_len,
                                        seq_len, ws->tok, fbmask, keep_lm, seed, lm_out_buf, ws->lm_hprev, ws->lm_act, ws->dec_gates,
                                        ws->dec_c, ws->dec_h, ws->y, ws->alpha, ws->ctx, ws->greedy_ws, ws->err,
                                        B, Te, D, A, H, lmH, E, V, T))) return rc;
        // x (saved for the backward: only the outer cell's weight gradient reads it, on the side stream) = lm_out . W_inp[:P] +
        // b_inp + ctx_prev . W_inp[P:]  -- two GEMMs over all steps, on the side stream next to the projections of p and the logits
        if (side_small) {
            hipEvent_t e_k = asr::next_event();
            if (hipEventRecord(e_k, ms) != hipSuccess || hipStreamWaitEvent(ss, e_k, 0) != hipSuccess) return ASR_ELAUNCH;
        }
        if ((rc = asr_gemm_f32(small_s, 0, 0, T * B, E, lmH, lm_out_buf, lmH, w->inp_w, E, ws->x, E, w->inp_b, 0))) return rc;
        if (T > 1 && (rc = asr_gemm_f32(small_s, 0, 0, (T - 1) * B, E, D, ws->ctx, D, w->inp_w + (size_t)lmH * E, E,
                                        ws->x + (size_t)B * E, E, nullptr, 1))) return rc;
        if (side_small) {
            e_x = asr::next_event();
            if (hipEventRecord(e_x, ss) != hipSuccess) return ASR_ELAUNCH;
        }
        done_train_kernel = true;
    }
    // ---- persistent chain path: the per-step attention chain of a whole SEGMENT (steps up to and
    // including the next feedback step) runs in one launch of csrc/decoder_chain.hip
    const bool use_chain = !done_train_kernel && mode != 1 && ws->chain_ws && ws->w2k && ws->err && ws->y && ws->dec_gates &&
                           asr_decoder_chain_supported(B, Te, D, A, H);
    if (use_chain) {
        // WK = W_inp . K_x  ([P+D,E].[E,4H]) and b' = b_inp . K_x + b_dec: InputProjection folded into the
        // outer cell, so that gates_i = lm_out_i . WK[:P] + ctx_{i-1} . WK[P:] + h_{i-1} . K_h + b'
        float* wk = ws->w2k;                       // [(P+D), 4H] followed by b' [4H]
        float* bprime = wk + (size_t)(P + D) * 4 * H;
        if ((rc = asr_gemm_f32(stream, 0, 0, P + D, 4 * H, E, w->inp_w, E, w->dec_kernel, 4 * H, wk, 4 * H, nullptr, 0))) return rc;
        if ((rc = asr_gemm_f32(stream, 0, 0, 1, 4 * H, E, w->inp_b, E, w->dec_kernel, 4 * H, bprime, 4 * H, w->dec_bias, 0))) return rc;
        const float* lm_base = w->simple_w ? ws->sp : (keep_lm < 1.0f ? ws->lm_hd : ws->lm_h);
        // persistent LM cell chain (csrc/lstm.hip, time-major): x.K_x + b of ALL steps from the teacher tokens by one
        // gather + GEMM up front (the rows of a step fed with a sampled token are redone when it exists), then one
        // recurrent launch per segment with the previous segment's final (h,c) as initial state.  ws->x is scratch here:
        // it is (re)built after the loop on the main stream, which by then has waited for this side-stream work.
        const bool lm_chain = ws->lm_act && ws->lm_hprev && ws->lm_state && ws->lm_len && ws->lm_hx && ws->lm_gates &&
                              asr_decoder_lm_chain_supported(B, lmH);
        if (lm_chain) {
            if ((rc = asr_gather_rows(side, w->embedding, ws->tok, ws->x, T * B, E))) return rc;
            if ((rc = asr_gemm_f32(side, 0, 0, T * B, 4 * lmH, E, ws->x, E, w->lm_kernel, 4 * lmH, ws->lm_gates, 4 * lmH,
                                   w->lm_bias, 0))) return rc;
        }
        float* lm_out_buf = keep_lm < 1.0f ? ws->lm_hd : ws->lm_h;
        int seg = 0;
        int t0 = 0;
        while (t0 < T) {
            int t1 = t0;
            while (t1 < T && !feedback(t1)) ++t1;
            t1 = t1 < T ? t1 + 1 : T;                       // the feedback step closes the segment
            // side stream: LM cells of the segment (they need the token produced by the previous segment)
            if (lm_chain) {
                // one recurrent launch per segment, on the MAIN stream: the segment's LM needs the token the previous
                // segment produced and its attention chain needs this LM output, so a second stream would only add two
                // cross-queue hand-overs (~14 us each) per segment; only the hoisted x.K_x product ran on the side stream
                const size_t o0 = (size_t)t0 * B;
                if (t0 == 0) {
                    hipEvent_t e_pre = asr::next_event();
                    if (hipEventRecord(e_pre, ss) != hipSuccess || hipStreamWaitEvent(ms, e_pre, 0) != hipSuccess) return ASR_ELAUNCH;
                }
                if (feedback(t0 - 1) &&       // the sampled token of step t0: redo its rows of x.K_x + b
                    (rc = asr_linear_fwd(stream, w->embedding, E, E, ws->tok + o0, nullptr, 0, 0, w->lm_kernel, 4 * lmH, w->lm_bias,
                                         ws->lm_gates + o0 * 4 * lmH, 4 * lmH, B, 4 * lmH, nullptr, 0))) return rc;
                float* st_in = ws->lm_state + (size_t)((seg + 1) & 1) * 2 * B * lmH;
                float* st_out = ws->lm_state + (size_t)(seg & 1) * 2 * B * lmH;
                if ((rc = asr_lstm_rec_fwd_tm(ms, ws->lm_gates + o0 * 4 * lmH, w->lm_kernel + (size_t)E * 4 * lmH, ws->lm_len,
                                              lm_out_buf + o0 * lmH, lmH, ws->lm_act + o0 * lmH * 8, ws->lm_hprev + o0 * lmH,
                                              t0 ? st_in : nullptr, t0 ? st_in + (size_t)B * lmH : nullptr, st_out,
                                              st_out + (size_t)B * lmH, ws->lm_hx, ws->err, B, t1 - t0, lmH, t0, keep_lm, seed)))
                    return rc;
                if (w->simple_w) {
                    const int rows = (t1 - t0) * B;
                    if (rows <= 512) {
                        if ((rc = asr_linear_fwd(stream, lm_out_buf + o0 * lmH, lmH, lmH, nullptr, nullptr, 0, 0, w->simple_w, H,
                                                 w->simple_b, ws->sp + o0 * H, H, rows, H, nullptr, 0))) return rc;
                    } else if ((rc = asr_gemm_f32(stream, 0, 0, rows, H, lmH, lm_out_buf + o0 * lmH, lmH, w->simple_w, H,
                                                  ws->sp + o0 * H, H, w->simple_b, 0))) return rc;
                }
                ++seg;
            } else if (feedback(t0 - 1) && hipStreamWaitEvent(ss, e_tok, 0) != hipSuccess) return ASR_ELAUNCH;
            for (int i = t0; i < t1 && !lm_chain; ++i) {
                const size_t o = (size_t)i * B;
                const float* lm_hp = i ? ws->lm_h + (o - B) * lmH : ws->zeros;
                const float* lm_cp = i ? ws->lm_c + (o - B) * lmH : nullptr;
                if ((rc = asr_lstm_cell_fwd(side, w->embedding, E, E, ws->tok + o, lm_hp, lm_cp, w->lm_kernel,
                                            w->lm_bias, lmH, B, ws->lm_c + o * lmH, ws->lm_h + o * lmH,
                                            keep_lm < 1.0f ? ws->lm_hd + o * lmH : nullptr,
                                            ws->lm_gates ? ws->lm_gates + o * 4 * lmH : nullptr, keep_lm, seed, (unsigned)i)))
                    return rc;
                if (w->simple_w) {
                    const float* lo = keep_lm < 1.0f ? ws->lm_hd + o * lmH : ws->lm_h + o * lmH;
                    if ((rc = asr_linear_fwd(side, lo, lmH, lmH, nullptr, nullptr, 0, 0, w->simple_w, H,
                                             w->simple_b, ws->sp + o * H, H, B, H, nullptr, 0)))
                        return rc;
                }
            }
            if (!lm_chain) {
                hipEvent_t e_lm = asr::next_event();
                if (hipEventRecord(e_lm, ss) != hipSuccess || hipStreamWaitEvent(ms, e_lm, 0) != hipSuccess) return ASR_ELAUNCH;
            }
            const int rows = (t1 - t0) * B;
            const size_t o0 = (size_t)t0 * B;
            // preG = lm_out . WK[:P] + b'  -> the gates buffer (the chain kernel overwrites it with the
            // activations); short segments through the skinny MFMA kernel, long ones through the GEMM
            if (rows <= 512) {
                if ((rc = asr_linear_fwd(stream, lm_base + o0 * P, P, P, nullptr, nullptr, 0, 0, wk, 4 * H, bprime,
                                         ws->dec_gates + o0 * 4 * H, 4 * H, rows, 4 * H, nullptr, 0))) return rc;
            } else if ((rc = asr_gemm_f32(stream, 0, 0, rows, 4 * H, P, lm_base + o0 * P, P, wk, 4 * H,
                                          ws->dec_gates + o0 * 4 * H, 4 * H, bprime, 0))) return rc;
            if ((rc = asr_decoder_chain_fwd(stream, ws->dec_gates, w->dec_kernel + (size_t)E * 4 * H, wk + (size_t)P * 4 * H, w->attn_w, w->attn_b,
                                            w->attn_v, ws->hf, enc, enc_len, ws->dec_c, ws->dec_h, ws->alpha, ws->ctx, ws->y,
                                            ws->chain_ws, ws->err, B, Te, D, A, H, t0, t1))) return rc;
            const int i = t1 - 1;
            if (feedback(i)) {
                const size_t o = (size_t)i * B;
                if ((rc = asr_linear_fwd(stream, ws->dec_c + o * H, H, H, nullptr, ws->ctx + o * D, D, D, w->ap_w, H,
                                         w->ap_b, ws->p + o * H, H, B, H, nullptr, 0))) return rc;
                if ((rc = asr_linear_fwd(stream, ws->p + o * H, H, H, nullptr, nullptr, 0, 0, w->out_w, V, w->out_b,
                                         logits + o * V, V, B, V, seq_len, i))) return rc;
                if ((rc = asr_next_token(stream, logits + o * V, B, V, V, ws->tok + o + B, mode == 2 ? 1 : 0, seed, (unsigned)i)))
                    return rc;
                e_tok = asr::next_event();
                if (hipEventRecord(e_tok, ms) != hipSuccess) return ASR_ELAUNCH;
            }
            t0 = t1;
        }
        // x (saved for the backward) = lm_out . W_inp[:P] + b_inp + ctx_prev . W_inp[P:]  -- two GEMMs over all steps
        if ((rc = asr_gemm_f32(stream, 0, 0, T * B, E, P, lm_base, P, w->inp_w, E, ws->x, E, w->inp_b, 0))) return rc;
        if (T > 1 && (rc = asr_gemm_f32(stream, 0, 0, (T - 1) * B, E, D, ws->ctx, D, w->inp_w + (size_t)P * E, E,
                                        ws->x + (size_t)B * E, E, nullptr, 1))) return rc;
    }
    for (int i = 0; i < T && !use_chain && !done_train_kernel; ++i) {
        const size_t o = (size_t)i * B;
        // ---- side stream: LM cell of step i
        if (feedback(i - 1) && hipStreamWaitEvent(ss, e_tok, 0) != hipSuccess) return ASR_ELAUNCH;
        const float* lm_hp = i ? ws->lm_h + (o - B) * lmH : ws->zeros;
        const float* lm_cp = i ? ws->lm_c + (o - B) * lmH : nullptr;
        if ((rc = asr_lstm_cell_fwd(side, w->embedding, E, E, ws->tok + o, lm_hp, lm_cp, w->lm_kernel,
                                    w->lm_bias, lmH, B, ws->lm_c + o * lmH, ws->lm_h + o * lmH,
                                    keep_lm < 1.0f ? ws->lm_hd + o * lmH : nullptr,
                                    ws->lm_gates ? ws->lm_gates + o * 4 * lmH : nullptr, keep_lm, seed, (unsigned)i)))
            return rc;
        const float* lm_out = keep_lm < 1.0f ? ws->lm_hd + o * lmH : ws->lm_h + o * lmH;
        if (w->simple_w) {
            if ((rc = asr_linear_fwd(side, lm_out, lmH, lmH, nullptr, nullptr, 0, 0, w->simple_w, H,
                                     w->simple_b, ws->sp + o * H, H, B, H, nullptr, 0)))
                return rc;
            lm_out = ws->sp + o * H;
        }
        hipEvent_t e_lm = asr::next_event();
        if (hipEventRecord(e_lm, ss) != hipSuccess || hipStreamWaitEvent(ms, e_lm, 0) != hipSuccess) return ASR_ELAUNCH;
        // ---- main stream: the attention chain of step i
        const float* ctx_prev = i ? ws->ctx + (o - B) * D : ws->zeros;
        if ((rc = asr_linear_fwd(stream, lm_out, P, P, nullptr, ctx_prev, D, D, w->inp_w, E, w->inp_b,
                                 ws->x + o * E, E, B, E, nullptr, 0)))
            return rc;
        const float* dh = i ? ws->dec_h + (o - B) * H : ws->zeros;
        const float* dc = i ? ws->dec_c + (o - B) * H : nullptr;
        if ((rc = asr_lstm_cell_fwd(stream, ws->x + o * E, E, E, nullptr, dh, dc, w->dec_kernel, w->dec_bias, H, B,
                                    ws->dec_c + o * H, ws->dec_h + o * H, nullptr,
                                    ws->dec_gates ? ws->dec_gates + o * 4 * H : nullptr, 1.0f, 0, 0)))
            return rc;
        if ((rc = asr_attention_launch(stream, ws->dec_c + o * H, H, w->attn_w, w->attn_b, w->attn_v, ws->hf, enc,
                                       enc_len, ws->alpha + o * Te, ws->ctx + o * D, ws->y ? ws->y + o * A : nullptr,
                                       B, Te, H, A, D, 0)))
            return rc;
        if (feedback(i) || mode == 1) {      // this step's logits feed the next token (or eval mode): project now
            if ((rc = asr_linear_fwd(stream, ws->dec_c + o * H, H, H, nullptr, ws->ctx + o * D, D, D, w->ap_w, H,
                                     w->ap_b, ws->p + o * H, H, B, H, nullptr, 0)))
                return rc;
            if ((rc = asr_linear_fwd(stream, ws->p + o * H, H, H, nullptr, nullptr, 0, 0, w->out_w, V, w->out_b,
                                     logits + o * V, V, B, V, seq_len, i)))
                return rc;
            if (feedback(i)) {
                if ((rc = asr_next_token(stream, logits + o * V, B, V, V, ws->tok + o + B, mode == 2 ? 1 : 0, seed, (unsigned)i)))
                    return rc;
                e_tok = asr::next_event();
                if (hipEventRecord(e_tok, ms) != hipSuccess) return ASR_ELAUNCH;
            }
        }
    }
    if (mode != 1) {
        // p = [q | ctx] . W_ap + b and logits = p . W_out + b for ALL steps at once
        const int TB = T * B;
        if ((rc = asr_gemm_f32(stream, 0, 0, TB, H, H, ws->dec_c, H, w->ap_w, H, ws->p, H, w->ap_b, 0))) return rc;
        if ((rc = asr_gemm_f32(stream, 0, 0, TB, H, D, ws->ctx, D, w->ap_w + (size_t)H * H, H, ws->p, H, nullptr, 1))) return rc;
        if ((rc = asr_gemm_f32(stream, 0, 0, TB, V, H, ws->p, H, w->out_w, V, logits, V, w->out_b, 0))) return rc;
        if ((rc = asr_zero_finished_rows(stream, logits, seq_len, T, B, V))) return rc;
    }
    if (e_x && hipStreamWaitEvent(ms, e_x, 0) != hipSuccess) return ASR_ELAUNCH;      // whatever follows the call is ordered after x
    asr::prof_end(ASR_PROF_DECODER_FWD, ms);
    return ASR_OK;
}
