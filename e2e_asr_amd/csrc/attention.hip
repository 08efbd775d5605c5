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

namespace asr {

struct AttnArgs {
    const float* q; int ldq;          // [B][H] (decoder cell state c, decoder.py:79-80)
    const float* w_att; const float* b_att; const float* v;   // [H][A], [A], [A]
    const float* hf;                  // [B][Te][A]
    const float* enc;                 // [B][Te][D]
    const int* enc_len;               // [B]
    float* alpha;                     // [B][Te]
    float* ctx;                       // [B][D]
    float* y_out;                     // [B][A] query projection, saved for the backward (or nullptr)
    int B, Te, H, A, D;
    int len_shared;
    long long hf_bs, enc_bs;          // batch strides (elements); 0 = one utterance shared by all rows (beam search)
};

// dynamic LDS: qs[H] | y[A] | e[Te] | part[NT*4]
// 512 threads; every phase issues ALL of its global loads before consuming any (one memory round
// trip per phase instead of one per pass): up to PB float4 per thread per batch.
constexpr int ATT_NT = 512;
__global__ __launch_bounds__(ATT_NT) void attention_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = ATT_NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int A = a.A, H = a.H, D = a.D;
    float* qs = smem;
    float* ys = qs + ((H + 3) & ~3);
    float* es = ys + ((A + 3) & ~3);
    float* part = es + ((a.Te + 3) & ~3);
    __shared__ float wred[16];
    const int L = min(max(a.enc_len[a.len_shared ? 0 : b], 0), a.Te);

    for (int k = tid; k < H; k += NT) qs[k] = a.q[(size_t)b * a.ldq + k];
    __syncthreads();
    // ---- y = q.W_att + b_att : thread -> (a4 = 4 columns, kp = K part); PB loads in flight
    {
        const int na4 = A >> 2;
        const int kparts = max(1, NT / na4);
        const int a4 = tid % na4, kp = tid / na4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kp < kparts) {
            const int kc = (H + kparts - 1) / kparts;
            const int k0 = kp * kc, k1 = min(H, k0 + kc);
            constexpr int PB = 16;
            for (int kb = k0; kb < k1; kb += PB) {
                float4 wv[PB];
#pragma unroll
                for (int i = 0; i < PB; ++i)
                    wv[i] = (kb + i < k1) ? *reinterpret_cast<const float4*>(a.w_att + (size_t)(kb + i) * A + 4 * a4)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < PB; ++i) {
                    const float qk = (kb + i < k1) ? qs[kb + i] : 0.f;
                    s.x = fmaf(qk, wv[i].x, s.x); s.y = fmaf(qk, wv[i].y, s.y);
                    s.z = fmaf(qk, wv[i].z, s.z); s.w = fmaf(qk, wv[i].w, s.w);
                }
            }
        }
        *reinterpret_cast<float4*>(part + 4 * tid) = s;
        __syncthreads();
        for (int aa = tid; aa < A; aa += NT) {
            float acc = a.b_att[aa];
            for (int p = 0; p < kparts; ++p) acc += part[4 * (p * na4 + (aa >> 2)) + (aa & 3)];
            ys[aa] = acc;
            if (a.y_out) a.y_out[(size_t)b * A + aa] = acc;
        }
        __syncthreads();
    }
    // ---- scores: one DPP row (16 lanes) per position; NT/16 positions per pass, PP passes in flight
    {
        const int kq = lane & 15, rr = tid >> 4;
        constexpr int RW = NT / 16, PP = 4, CH = 2;     // CH float4 chunks per lane per row (A <= 128 fast path)
        const int nch = ((A >> 2) + 15) / 16;
        for (int t0 = 0; t0 < L; t0 += RW * PP) {
            if (nch <= CH) {
                float4 hv[PP][CH];
#pragma unroll
                for (int p = 0; p < PP; ++p) {
                    const int tau = t0 + p * RW + rr;
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const int a4 = kq + 16 * c;
                        hv[p][c] = (tau < L && a4 < (A >> 2))
                            ? *reinterpret_cast<const float4*>(a.hf + (size_t)b * a.hf_bs + (size_t)tau * A + 4 * a4)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int p = 0; p < PP; ++p) {
                    const int tau = t0 + p * RW + rr;
                    float sc = 0.f;
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const int a4 = kq + 16 * c;
                        if (a4 < (A >> 2)) {
                            const float4 yv = *reinterpret_cast<const float4*>(ys + 4 * a4);
                            const float4 vv = *reinterpret_cast<const float4*>(a.v + 4 * a4);
                            sc = fmaf(vv.x, fast_tanh(hv[p][c].x + yv.x), sc);
                            sc = fmaf(vv.y, fast_tanh(hv[p][c].y + yv.y), sc);
                            sc = fmaf(vv.z, fast_tanh(hv[p][c].z + yv.z), sc);
                            sc = fmaf(vv.w, fast_tanh(hv[p][c].w + yv.w), sc);
                        }
                    }
                    sc = row16_allreduce_sum(sc);
                    if (kq == 0 && tau < L) es[tau] = sc;
                }
            } else {      // wide attention vectors: plain loop
                for (int p = 0; p < PP; ++p) {
                    const int tau = t0 + p * RW + rr;
                    float sc = 0.f;
                    if (tau < L)
                        for (int a4 = kq; a4 < (A >> 2); a4 += 16) {
                            const float4 h4 = *reinterpret_cast<const float4*>(a.hf + (size_t)b * a.hf_bs + (size_t)tau * A + 4 * a4);
                            const float4 yv = *reinterpret_cast<const float4*>(ys + 4 * a4);
                            const float4 vv = *reinterpret_cast<const float4*>(a.v + 4 * a4);
                            sc = fmaf(vv.x, fast_tanh(h4.x + yv.x), sc); sc = fmaf(vv.y, fast_tanh(h4.y + yv.y), sc);
                            sc = fmaf(vv.z, fast_tanh(h4.z + yv.z), sc); sc = fmaf(vv.w, fast_tanh(h4.w + yv.w), sc);
                        }
                    sc = row16_allreduce_sum(sc);
                    if (kq == 0 && tau < L) es[tau] = sc;
                }
            }
        }
        __syncthreads();
    }
    // ---- softmax over tau < L
    float m = -INFINITY;
    for (int tau = tid; tau < L; tau += NT) m = fmaxf(m, es[tau]);
    m = wave_allreduce_max(m);
    if (lane == 0) wred[wave] = m;
    __syncthreads();
    m = wred[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) m = fmaxf(m, wred[i]);
    float sum = 0.f;
    for (int tau = tid; tau < L; tau += NT) { const float p = __expf(es[tau] - m); es[tau] = p; sum += p; }
    sum = wave_allreduce_sum(sum);
    if (lane == 0) wred[8 + wave] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) tot += wred[8 + i];
    const float inv = 1.0f / tot;
    for (int tau = tid; tau < a.Te; tau += NT) {
        const float p = tau < L ? es[tau] * inv : 0.f;
        if (tau < L) es[tau] = p;
        a.alpha[(size_t)b * a.Te + tau] = p;
    }
    __syncthreads();
    // ---- ctx = alpha . enc : thread -> (d4 = 4 columns, tp = tau part); PB rows in flight per batch
    {
        const int nd4 = D >> 2;
        for (int base = 0; base < nd4; base += NT) {
            const int cols = min(nd4 - base, NT);
            const int tparts = max(1, NT / cols);
            const int d4 = base + tid % cols, tp = tid / cols;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tp < tparts) {
                const float* ep = a.enc + (size_t)b * a.enc_bs + 4 * d4;
                constexpr int PB = 13;
                for (int tb = tp; tb < L; tb += tparts * PB) {
                    float4 ev[PB];
#pragma unroll
                    for (int i = 0; i < PB; ++i) {
                        const int tau = tb + i * tparts;
                        ev[i] = tau < L ? *reinterpret_cast<const float4*>(ep + (size_t)tau * D) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int i = 0; i < PB; ++i) {
                        const int tau = tb + i * tparts;
                        const float al = tau < L ? es[tau] : 0.f;
                        s.x = fmaf(al, ev[i].x, s.x); s.y = fmaf(al, ev[i].y, s.y);
                        s.z = fmaf(al, ev[i].z, s.z); s.w = fmaf(al, ev[i].w, s.w);
                    }
                }
            }
            *reinterpret_cast<float4*>(part + 4 * tid) = s;
            __syncthreads();
            if (tid < cols) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int p = 0; p < tparts; ++p) {
                    const float4 v4 = *reinterpret_cast<const float4*>(part + 4 * (p * cols + tid));
                    t.x += v4.x; t.y += v4.y; t.z += v4.z; t.w += v4.w;
                }
                *reinterpret_cast<float4*>(a.ctx + (size_t)b * D + 4 * (base + tid)) = t;
            }
            __syncthreads();
        }
    }
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
