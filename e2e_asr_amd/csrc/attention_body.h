// The per-utterance body of the fused attention kernel (csrc/attention.hip); also called by the persistent beam kernel
// (csrc/beam.hip).  See csrc/attention.hip for the reference lines and the phase structure.
#pragma once
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
// COH (common.h): how the query is read and the context written when the caller is the persistent beam kernel.
constexpr int ATT_NT = 512;
template <int COH>
__device__ __forceinline__ void attention_body(const AttnArgs& a, const int b, float* smem, float* wred) {
    constexpr int NT = ATT_NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int A = a.A, H = a.H, D = a.D;
    float* qs = smem;
    float* ys = qs + ((H + 3) & ~3);
    float* es = ys + ((A + 3) & ~3);
    float* part = es + ((a.Te + 3) & ~3);
    const int L = min(max(a.enc_len[a.len_shared ? 0 : b], 0), a.Te);

    for (int k = tid; k < H; k += NT) qs[k] = ld_data<COH>(a.q + (size_t)b * a.ldq + k);
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
                if constexpr (COH != 0) {
                    float* cp = a.ctx + (size_t)b * D + 4 * (base + tid);
                    st_f<1>(cp, t.x); st_f<1>(cp + 1, t.y); st_f<1>(cp + 2, t.z); st_f<1>(cp + 3, t.w);
                } else *reinterpret_cast<float4*>(a.ctx + (size_t)b * D + 4 * (base + tid)) = t;
            }
            __syncthreads();
        }
    }
}

}  // namespace asr
