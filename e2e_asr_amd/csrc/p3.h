// "P3" operands: matrices stored as bf16 planes for the GEMMs of csrc/gemm_p3.hip (layout: include/e2e_asr_hip.h, asr_p3_bytes).
// Shared by the producers: the recurrent kernels write their outputs (h, h_prev, dG) as planes, so that no GEMM has to split
// fp32 operands inside its k-loop.
#pragma once
#include "common.h"

namespace asr {

typedef __bf16 p3h_bf16x2 __attribute__((ext_vector_type(2)));
// v_cvt_pk_bf16_f32 (round to nearest even), emitted by the compiler so that it tracks the instruction's hazards
__device__ __forceinline__ uint32_t p3_pack2(float lo, float hi) {
    const f32x2 t = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(t, p3h_bf16x2));
}

// x0, x1 -> NP packed pairs, plane by plane: x = h1 + h2 + h3 exactly at NP = 3 (every residual of a bf16 rounding is
// representable in fp32, 3 x 8 = 24 significand bits)
template <int NP>
__device__ __forceinline__ void p3_split_pair(float x0, float x1, uint32_t* p) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        p[q] = p3_pack2(x0, x1);
        if (q + 1 < NP) { x0 -= __uint_as_float(p[q] << 16); x1 -= __uint_as_float(p[q] & 0xffff0000u); }
    }
}

// byte offset of element (row, col) of plane 0 in a P3 image with row pitch ld8 chunks and np planes
__device__ __forceinline__ size_t p3_elem_off(size_t row, int col, int ld8, int np) {
    return ((row * (size_t)ld8 + (size_t)(col >> 3)) * np) * 16 + (size_t)(col & 7) * 2;
}

// one value -> its np planes (2-byte stores 16 bytes apart)
__device__ __forceinline__ void p3_store1(char* base, size_t off, float x, int np, bool nontemporal) {
    uint32_t p[3];
    p3_split_pair<3>(x, 0.f, p);
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (q < np) {
            unsigned short* d = reinterpret_cast<unsigned short*>(base + off + q * 16);
            if (nontemporal) __builtin_nontemporal_store((unsigned short)(p[q] & 0xffffu), d);
            else *d = (unsigned short)(p[q] & 0xffffu);
        }
}

// four consecutive values (8 bytes per plane: half a piece; `off` = p3_elem_off of the first, col % 4 == 0)
__device__ __forceinline__ void p3_store4(char* base, size_t off, float x0, float x1, float x2, float x3, int np, bool nontemporal) {
    uint32_t a[3], b[3];
    p3_split_pair<3>(x0, x1, a);
    p3_split_pair<3>(x2, x3, b);
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (q < np) {
            typedef unsigned int p3h_u32x2 __attribute__((ext_vector_type(2)));
            p3h_u32x2* d = reinterpret_cast<p3h_u32x2*>(base + off + q * 16);
            const p3h_u32x2 v = {a[q], b[q]};
            if (nontemporal) __builtin_nontemporal_store(v, d);
            else *d = v;
        }
}

// csrc/gemm_p3.hip: dK_d += [X | Hprev_d]^T . dG_d for every direction of one layer, one launch
int p3_lstm_wgrad(hipStream_t s, int rows, int in_pad, int in_valid, int H, int ndir, const void* x_p3, int x_ld8,
                  const void* hprev_p3, const void* dg_p3, int np, float* dk, long long dk_stride, const int* colmap);

}  // namespace asr
