// The tile body of the skinny step kernels (csrc/skinny.hip); also called, phase by phase, by the persistent beam kernel
// (csrc/beam.hip).
#pragma once
#include "common.h"
#include "skinny.h"

namespace asr {

// WT: the weight is given transposed, W[n][k] (row stride ldw along n) -- the data-gradient
// products dY.W^T of the backward pass; the lane then reads 4 consecutive k as one float4.
// COH (common.h): how activations, cell states and gather indices are read and results written when the caller is the
// persistent beam kernel (csrc/beam.hip), whose consecutive phases run in workgroups of DIFFERENT XCDs without a kernel
// boundary between them.  `tid` = the
// thread's index within its 256-thread problem (a 512-thread workgroup of that kernel runs two tiles side by side).
template <bool LSTM, bool WT, int COH = 0>
__device__ __forceinline__ void skinny_body(const SkinnyArgs& a, const int bx, const int by, float (*red)[4][64], const int tid) {
    const int lane = tid & 63, w = tid >> 6;
    const int nl = lane & 15, g4 = lane >> 4;
    const int K = a.K1 + a.K2;
    // column of W feeding output row nl of the MFMA tile
    int colW; bool colok;
    if (LSTM) { const int j = bx * 4 + (nl >> 2); colW = a.wperm ? bx * 16 + nl : (nl & 3) * a.H + j; colok = j < a.H; }
    else      { colW = bx * 16 + nl; colok = colW < a.N; }
    const int brow = by * 16 + nl;      // batch row this lane supplies as MFMA B operand
    const bool rowok = brow < a.M;
    const float* xr1 = nullptr; const float* xr2 = nullptr;
    if (rowok) {
        xr1 = a.x1 + (size_t)(a.gather1 ? ld_i<COH>(a.gather1 + brow) : brow) * a.ld1;
        if (a.K2 > 0) xr2 = a.x2 + (size_t)(a.gather2 ? ld_i<COH>(a.gather2 + brow) : brow) * a.ld2;
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
                if (a.c_prev) ecp = ld_data<COH>(a.c_prev + (size_t)(a.gather2 ? ld_i<COH>(a.gather2 + brow) : brow) * a.H + j);
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
                // COH: x1 gathered through `gather1` is a constant table (the embedding): a plain load
                if (rowok) xv[it] = (k < a.K1) ? (a.gather1 ? ld_data4<0>(xr1 + k) : ld_data4<COH>(xr1 + k)) : ld_data4<COH>(xr2 + (k - a.K1));
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
        const float c = fmaf(cp, gf, gi * gj);          // explicit: the two legal contractions of cp*gf + gi*gj round differently
        const float h = go * fast_tanh(c);
        st_f<COH>(a.c_out + (size_t)b * H + j, c);
        st_f<COH>(a.h_out + (size_t)b * H + j, h);
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
                st_f<COH>(op + r, v);
            }
    }
}

}  // namespace asr
