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
// One workgroup (256 threads) per utterance; HBM/L2-bound streaming of hf (Te x A) and
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
    int B, Te, H, A, D;
};

// dynamic LDS: qs[H] | y[A] | e[Te] | part[NT*4]
__global__ __launch_bounds__(256) void attention_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NT = 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int A = a.A, H = a.H, D = a.D;
    float* qs = smem;
    float* ys = qs + ((H + 3) & ~3);
    float* es = ys + ((A + 3) & ~3);
    float* part = es + ((a.Te + 3) & ~3);
    __shared__ float wred[8];
    const int L = min(max(a.enc_len[b], 0), a.Te);

    for (int k = tid; k < H; k += NT) qs[k] = a.q[(size_t)b * a.ldq + k];
    __syncthreads();
    // ---- y = q.W_att + b_att : thread -> (a4 = 4 columns, kp = K part), all loads in flight
    {
        const int na4 = A >> 2;
        const int kparts = max(1, NT / na4);
        const int a4 = tid % na4, kp = tid / na4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kp < kparts) {
            const int kc = (H + kparts - 1) / kparts;
            const int k0 = kp * kc, k1 = min(H, k0 + kc);
#pragma unroll 8
            for (int k = k0; k < k1; ++k) {
                const float4 wv = *reinterpret_cast<const float4*>(a.w_att + (size_t)k * A + 4 * a4);
                const float qk = qs[k];
                s.x = fmaf(qk, wv.x, s.x); s.y = fmaf(qk, wv.y, s.y);
                s.z = fmaf(qk, wv.z, s.z); s.w = fmaf(qk, wv.w, s.w);
            }
        }
        *reinterpret_cast<float4*>(part + 4 * tid) = s;
        __syncthreads();
        for (int aa = tid; aa < A; aa += NT) {
            float acc = a.b_att[aa];
            for (int p = 0; p < kparts; ++p) acc += part[4 * (p * na4 + (aa >> 2)) + (aa & 3)];
            ys[aa] = acc;
        }
        __syncthreads();
    }
    // ---- scores: one DPP row (16 lanes) per position, 16 positions in flight per pass
    {
        const int kq = lane & 15, rr = tid >> 4;
        for (int tau = rr; tau < L; tau += NT / 16) {
            const float* hp = a.hf + ((size_t)b * a.Te + tau) * A;
            float s = 0.f;
            for (int a4 = kq; a4 < (A >> 2); a4 += 16) {
                const float4 hv = *reinterpret_cast<const float4*>(hp + 4 * a4);
                const float4 yv = *reinterpret_cast<const float4*>(ys + 4 * a4);
                const float4 vv = *reinterpret_cast<const float4*>(a.v + 4 * a4);
                s = fmaf(vv.x, fast_tanh(hv.x + yv.x), s);
                s = fmaf(vv.y, fast_tanh(hv.y + yv.y), s);
                s = fmaf(vv.z, fast_tanh(hv.z + yv.z), s);
                s = fmaf(vv.w, fast_tanh(hv.w + yv.w), s);
            }
            s = row16_allreduce_sum(s);
            if (kq == 0) es[tau] = s;
        }
        __syncthreads();
    }
    // ---- softmax over tau < L
    float m = -INFINITY;
    for (int tau = tid; tau < L; tau += NT) m = fmaxf(m, es[tau]);
    m = wave_allreduce_max(m);
    if (lane == 0) wred[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
    float sum = 0.f;
    for (int tau = tid; tau < L; tau += NT) { const float p = __expf(es[tau] - m); es[tau] = p; sum += p; }
    sum = wave_allreduce_sum(sum);
    if (lane == 0) wred[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (wred[4] + wred[5] + wred[6] + wred[7]);
    for (int tau = tid; tau < a.Te; tau += NT) {
        const float p = tau < L ? es[tau] * inv : 0.f;
        if (tau < L) es[tau] = p;
        a.alpha[(size_t)b * a.Te + tau] = p;
    }
    __syncthreads();
    // ---- ctx = alpha . enc : thread -> (d4 = 4 columns, tp = tau part)
    {
        const int nd4 = D >> 2;
        for (int base = 0; base < nd4; base += NT) {
            const int cols = min(nd4 - base, NT);
            const int tparts = max(1, NT / cols);
            const int d4 = base + tid % cols, tp = tid / cols;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tp < tparts) {
                const float* ep = a.enc + (size_t)b * a.Te * D + 4 * d4;
#pragma unroll 8
                for (int tau = tp; tau < L; tau += tparts) {
                    const float4 ev = *reinterpret_cast<const float4*>(ep + (size_t)tau * D);
                    const float al = es[tau];
                    s.x = fmaf(al, ev.x, s.x); s.y = fmaf(al, ev.y, s.y);
                    s.z = fmaf(al, ev.z, s.z); s.w = fmaf(al, ev.w, s.w);
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
    return sizeof(float) * (size_t)(((H + 3) & ~3) + ((A + 3) & ~3) + ((Te + 3) & ~3) + 256 * 4);
}

extern "C" int asr_attention_fwd(void* stream, const float* q, int ldq, const float* w_att,
                                 const float* b_att, const float* v, const float* hf,
                                 const float* enc, const int* enc_len, float* alpha, float* ctx,
                                 int B, int Te, int H, int A, int D) {
    if (!q || !w_att || !b_att || !v || !hf || !enc || !enc_len || !alpha || !ctx) return ASR_EINVAL;
    if (B <= 0 || Te <= 0 || H <= 0 || (A & 3) || (D & 3) || A <= 0 || D <= 0 || A > 1024) return ASR_EINVAL;
    const size_t lds = asr_attention_lds_bytes(Te, H, A);
    if (lds > 150 * 1024) return ASR_EUNSUPPORTED;
    asr::AttnArgs a{q, ldq, w_att, b_att, v, hf, enc, enc_len, alpha, ctx, B, Te, H, A, D};
    hipLaunchKernelGGL(asr::attention_fwd_kernel, dim3(B), dim3(256), lds, static_cast<hipStream_t>(stream), a);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
