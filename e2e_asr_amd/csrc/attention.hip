// Fused Bahdanau content attention: query projection + score + masked softmax + context.
//
// Reference: attn_decoder.py:77-93 --
//   y = q.W_att + b_att                       (:80, `_linear(query, A, True)`)
//   e[tau] = sum_a v[a] * tanh(hf[b,tau,a] + y[a])   (:82-83; hf = enc.AttnW, :70-73)
//   alpha = softmax(e) * mask; alpha /= sum(alpha)   (:85-88)
//   ctx = sum_tau alpha[tau] * enc[b,tau,:]          (:92)
// and its NumPy twin beam_search.py:150-159 (unmasked: states pre-sliced to the length).
// softmax-then-mask-then-renormalise equals a softmax restricted to tau < len (the
// common exp(-max) and the full-length denominator cancel), so only the live positions
// are ever read: padded hf/enc rows cost no bandwidth.
//
// One workgroup (512 threads) per utterance; HBM/L2-bound streaming of hf (Te x A) and
// enc (Te x D) with 16-byte loads; 16-lane DPP-row reductions for the per-position
// score, wavefront reductions for the softmax; nothing but alpha/ctx is written.
#include "common.h"
#include "attention_body.h"

namespace asr {

__global__ __launch_bounds__(ATT_NT) void attention_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float wred[16];
    attention_body<0>(a, blockIdx.x, smem, wred);
}

}  // namespace asr

extern "C" size_t asr_attention_lds_bytes(int Te, int H, int A) {
    return sizeof(float) * (size_t)(((H + 3) & ~3) + ((A + 3) & ~3) + ((Te + 3) & ~3) + 512 * 4);
}

extern "C" int asr_attention_shared_fwd(void* stream, const float* q, int ldq, const float* w_att,
                                        const float* b_att, const float* v, const float* hf,
                                        const float* enc, const int* enc_len, float* alpha, float* ctx,
                                        int B, int Te, int H, int A, int D, int shared);

extern "C" int asr_attention_fwd(void* stream, const float* q, int ldq, const float* w_att,
                                 const float* b_att, const float* v, const float* hf,
                                 const float* enc, const int* enc_len, float* alpha, float* ctx,
                                 int B, int Te, int H, int A, int D) {
    return asr_attention_shared_fwd(stream, q, ldq, w_att, b_att, v, hf, enc, enc_len, alpha, ctx, B, Te, H, A, D, 0);
}

int asr_attention_launch(void* stream, const float* q, int ldq, const float* w_att, const float* b_att, const float* v,
                         const float* hf, const float* enc, const int* enc_len, float* alpha, float* ctx, float* y_out,
                         int B, int Te, int H, int A, int D, int shared);

// shared != 0: hf/enc/enc_len describe ONE utterance attended by all B query rows (beam search).
extern "C" int asr_attention_shared_fwd(void* stream, const float* q, int ldq, const float* w_att,
                                        const float* b_att, const float* v, const float* hf,
                                        const float* enc, const int* enc_len, float* alpha, float* ctx,
                                        int B, int Te, int H, int A, int D, int shared) {
    return asr_attention_launch(stream, q, ldq, w_att, b_att, v, hf, enc, enc_len, alpha, ctx, nullptr, B, Te, H, A, D, shared);
}

// internal (C++ linkage): also saves the query projection y [B,A] for the backward pass
int asr_attention_launch(void* stream, const float* q, int ldq, const float* w_att, const float* b_att, const float* v,
                         const float* hf, const float* enc, const int* enc_len, float* alpha, float* ctx, float* y_out,
                         int B, int Te, int H, int A, int D, int shared) {
    if (!q || !w_att || !b_att || !v || !hf || !enc || !enc_len || !alpha || !ctx) return ASR_EINVAL;
    if (B <= 0 || Te <= 0 || H <= 0 || (A & 3) || (D & 3) || A <= 0 || D <= 0 || A > 1024) return ASR_EINVAL;
    const size_t lds = asr_attention_lds_bytes(Te, H, A);
    if (lds > 150 * 1024) return ASR_EUNSUPPORTED;
    asr::AttnArgs a{};
    a.q = q; a.ldq = ldq; a.w_att = w_att; a.b_att = b_att; a.v = v; a.hf = hf; a.enc = enc; a.enc_len = enc_len;
    a.alpha = alpha; a.ctx = ctx; a.B = B; a.Te = Te; a.H = H; a.A = A; a.D = D;
    a.len_shared = shared;
    a.y_out = y_out;
    a.hf_bs = shared ? 0 : (long long)Te * A; a.enc_bs = shared ? 0 : (long long)Te * D;
    hipLaunchKernelGGL(asr::attention_fwd_kernel, dim3(B), dim3(asr::ATT_NT), lds, static_cast<hipStream_t>(stream), a);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
