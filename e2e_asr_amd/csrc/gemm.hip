// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 fmaf chain).
//
//   C[M,N] (+)= op(A) . op(B) + bias[N]
//
// Used for every non-recurrent dense contraction on the hot path: the all-timestep
// LSTM input projection X.K_x (encoder.py:78-81 hoisted out of the while_loop), the
// attention precompute enc.AttnW (attn_decoder.py:70-73), and in training the data
// and weight gradient products.
//
// Tiling (wave64, not warp32): 128x128 block tile, BK=16, 4 waves as 2x2, each wave a
// 64x64 sub-tile = 2x2 MFMA 32x32 tiles (4 x 16 accumulator registers).  LDS holds both
// operands k-major ([k][m] / [k][n]) so that an MFMA operand read is one conflict-free
// ds_read_b32 per lane (lanes 0-31: consecutive m at k, lanes 32-63: at k+1).  Global ->
// register prefetch of tile t+1 is issued before the MFMAs of tile t.
// A second instantiation with a 64x64 block tile (one 32x32 MFMA tile per wave) serves products
// with few output tiles (the decoder's [3840,256]-shaped ones): 4x the workgroups, so the chip is
// filled without split-K and the result stays bit-reproducible.
#include "common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>

namespace asr {

constexpr int BK = 16;

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc;
    int accumulate;   // C += result
    int splits;       // K split over blockIdx.y; >1 => atomicAdd epilogue into a pre-zeroed / accumulated C
    int vecA, vecB;   // 16-B vector loads legal for this operand
    int xcd_split;    // split-K with one 1-D grid: K slice s runs entirely on XCD s % 8 (splits is a multiple of 8)
    long long sA, sB, sC;   // batch strides (elements); batch index = blockIdx.z
    float* slab; int slab_ld;   // splits > 1: partial tiles go to slab[(blockIdx.z * splits + slice) * M * slab_ld + m * slab_ld + n] (csrc/splitk.hip); NULL: float atomics into C
    float* sk_ws; int* sk_flag; int sk_epoch;   // gemm_planes_kernel<..., SK = true>: one partial tile ([256 threads][64 accumulators]) and one flag per workgroup
};

// Load 4 consecutive floats p[0..3] where element i is valid iff i < nvalid.
__device__ __forceinline__ float4 ld4(const float* p, int nvalid, bool vec) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid >= 4 && vec) return *reinterpret_cast<const float4*>(p);
    if (nvalid > 0) r.x = p[0];
    if (nvalid > 1) r.y = p[1];
    if (nvalid > 2) r.z = p[2];
    if (nvalid > 3) r.w = p[3];
    return r;
}

// Operand staging for a BT-wide tile (BT = 128 or 64), 256 threads.
// "KM" = operand stored [k][m] in memory (contiguous along m/n): thread -> (k = tid>>4, mq = tid&15),
// float4 at m = mq*4 (and mq*4+64 for BT = 128).
// "MK" = operand stored [m][k] (contiguous along k): BT = 128: thread -> (m = tid>>1, kh = tid&1), two
// float4 at k = kh*8, kh*8+4; BT = 64: thread -> (m = tid>>2, kq = tid&3), one float4 at k = kq*4;
// transposed on the LDS store.
template <bool KMAJOR, int BT>
struct Stager {
    static constexpr int LDT = BT + 4;   // padded k-row (16-B aligned)
    float4 r0, r1;
    // FULL: the whole tile is in range and 16-B aligned -> unconditional vector loads
    __device__ __forceinline__ void load_full(const float* P, int ld, int m0, int k0, int tid) {
        if (KMAJOR) {
            const float* p = P + (size_t)(k0 + (tid >> 4)) * ld + m0 + (tid & 15) * 4;
            r0 = *reinterpret_cast<const float4*>(p);
            if (BT == 128) r1 = *reinterpret_cast<const float4*>(p + 64);
        } else if (BT == 128) {
            const float* p = P + (size_t)(m0 + (tid >> 1)) * ld + k0 + (tid & 1) * 8;
            r0 = *reinterpret_cast<const float4*>(p);
            r1 = *reinterpret_cast<const float4*>(p + 4);
        } else {
            r0 = *reinterpret_cast<const float4*>(P + (size_t)(m0 + (tid >> 2)) * ld + k0 + (tid & 3) * 4);
        }
    }
    __device__ __forceinline__ void load(const float* P, int ld, int m0, int k0, int Mdim, int Kdim,
                                         bool vec, int tid) {
        r0 = r1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KMAJOR) {
            int k = k0 + (tid >> 4), m = m0 + (tid & 15) * 4;
            if (k < Kdim) {
                const float* p = P + (size_t)k * ld + m;
                r0 = ld4(p, Mdim - m, vec);
                if (BT == 128) r1 = ld4(p + 64, Mdim - m - 64, vec);
            }
        } else if (BT == 128) {
            int m = m0 + (tid >> 1), k = k0 + (tid & 1) * 8;
            if (m < Mdim) {
                const float* p = P + (size_t)m * ld + k;
                r0 = ld4(p, Kdim - k, vec);
                r1 = ld4(p + 4, Kdim - k - 4, vec);
            }
        } else {
            int m = m0 + (tid >> 2), k = k0 + (tid & 3) * 4;
            if (m < Mdim) r0 = ld4(P + (size_t)m * ld + k, Kdim - k, vec);
        }
    }
    __device__ __forceinline__ void store(float* S, int tid) const {
        if (KMAJOR) {
            float* s = S + (tid >> 4) * LDT + (tid & 15) * 4;
            *reinterpret_cast<float4*>(s) = r0;
            if (BT == 128) *reinterpret_cast<float4*>(s + 64) = r1;
        } else if (BT == 128) {
            float* s = S + ((tid & 1) * 8) * LDT + (tid >> 1);
            s[0] = r0.x; s[LDT] = r0.y; s[2 * LDT] = r0.z; s[3 * LDT] = r0.w;
            s[4 * LDT] = r1.x; s[5 * LDT] = r1.y; s[6 * LDT] = r1.z; s[7 * LDT] = r1.w;
        } else {
            float* s = S + ((tid & 3) * 4) * LDT + (tid >> 2);
            s[0] = r0.x; s[LDT] = r0.y; s[2 * LDT] = r0.z; s[3 * LDT] = r0.w;
        }
    }
};

// TA: A is given transposed ([K,M] row-major).  TB: B is given transposed ([N,K] row-major).
template <bool TA, bool TB, int BT = 128, bool FULL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void gemm_f32_kernel(GemmArgs a) {
    constexpr int BM = BT, BN = BT, LDT = BT + 4, WT = BT / 2, MI = BT / 64;   // wave tile WT x WT = MI x MI MFMA tiles
    __shared__ __attribute__((aligned(16))) float As[2 * BK * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BK * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    a.A += (size_t)blockIdx.z * a.sA; a.B += (size_t)blockIdx.z * a.sB; a.C += (size_t)blockIdx.z * a.sC;

    // XCD-aware tile order: blocks b and b+8 share an L2, so give each XCD label a
    // contiguous run of row-tiles that re-use the same B panel / neighbouring A panels.
    const int ntn = (a.N + BN - 1) / BN, ntm = (a.M + BM - 1) / BM;
    const int nwg = ntn * ntm;
    int bid = blockIdx.x, ksl = blockIdx.y;
    if (a.xcd_split) {
        // weight-gradient form (few tiles, long K): workgroup L runs on XCD L & 7; give each XCD whole K slices (slice =
        // xcd + 8 * round), all tiles of a slice on ONE L2: every element of X and dG is then fetched by one XCD only
        // (with the slices spread over the XCDs each XCD streamed all of dG: ~4x the fills)
        const int xcd = bid & 7, idx = bid >> 3;
        ksl = xcd + 8 * (idx / nwg);
        bid = idx % nwg;
    } else {
        const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;

    f32x16 acc[MI][MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    Stager<TA, BT> sa;    // A natural [M,K] is m-major => KMAJOR = TA
    Stager<!TB, BT> sb;   // B natural [K,N] is k-major => KMAJOR = !TB
    // split-K: this block owns k-tiles [kt0, kt1)
    const int nk_all = (a.K + BK - 1) / BK;
    const int per = (nk_all + a.splits - 1) / a.splits;
    const int kt0 = ksl * per, kt1 = min(nk_all, kt0 + per);
    if (kt0 >= kt1) return;
    if (FULL) { sa.load_full(a.A, a.lda, m0, kt0 * BK, tid); sb.load_full(a.B, a.ldb, n0, kt0 * BK, tid); }
    else {
        sa.load(a.A, a.lda, m0, kt0 * BK, a.M, a.K, a.vecA, tid);
        sb.load(a.B, a.ldb, n0, kt0 * BK, a.N, a.K, a.vecB, tid);
    }
    const int nk = kt1;
    // double-buffered LDS: tile kt is read from buffer (kt-kt0)&1 while tile kt+1 goes global -> registers
    // (issued before the MFMAs) -> the other buffer (after them): ONE barrier per k-tile
    sa.store(As, tid);
    sb.store(Bs, tid);
    __syncthreads();
    auto compute = [&](int cur) {
        const float* ap = As + cur * (BK * LDT) + (lane >> 5) * LDT + wr * WT + (lane & 31);
        const float* bp = Bs + cur * (BK * LDT) + (lane >> 5) * LDT + wc * WT + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float av[MI], bv[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) { av[i] = ap[kk * LDT + 32 * i]; bv[i] = bp[kk * LDT + 32 * i]; }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };
    // steady state (no conditionals around the prefetch: its registers go straight from the load to the LDS store)
    for (int kt = kt0; kt + 1 < nk; ++kt) {
        const int cur = (kt - kt0) & 1;
        if (FULL) { sa.load_full(a.A, a.lda, m0, (kt + 1) * BK, tid); sb.load_full(a.B, a.ldb, n0, (kt + 1) * BK, tid); }
        else {
            sa.load(a.A, a.lda, m0, (kt + 1) * BK, a.M, a.K, a.vecA, tid);
            sb.load(a.B, a.ldb, n0, (kt + 1) * BK, a.N, a.K, a.vecB, tid);
        }
        compute(cur);
        sa.store(As + (cur ^ 1) * (BK * LDT), tid);
        sb.store(Bs + (cur ^ 1) * (BK * LDT), tid);
        __syncthreads();
    }
    compute((nk - 1 - kt0) & 1);
    // Epilogue.  C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < MI; ++ni) {
            const int n = n0 + wc * WT + ni * 32 + (lane & 31);
            if (!FULL && n >= a.N) continue;
            const float bv = (a.bias && ksl == 0) ? a.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * WT + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (FULL || m < a.M) {
                    float* cp = a.C + (size_t)m * a.ldc + n;
                    float v = acc[mi][ni][r] + bv;
                    if (a.splits > 1) {                          // 32 lanes x 4 B = one 128-B segment per row
                        if (a.slab) a.slab[((size_t)(blockIdx.z * a.splits + ksl) * a.M + m) * a.slab_ld + n] = v;
                        else atomicAdd(cp, v);
                    }
                    else { if (a.accumulate) v += *cp; *cp = v; }
                }
            }
        }
}


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// v_cvt_pk_bf16_f32 (round to nearest even), emitted by the compiler so that it tracks the instruction's hazards
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    const f32x2_t t = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(t, bf16x2_t));
}

// ---------------------------------------------------------------------------------------------
// fp32 product on the bf16 matrix pipe by EXACT operand splitting ("split3"): SAME interface and fp32-class accuracy.
// Every fp32 operand value is x = h1 + h2 + h3 with three bf16 terms (h1 = bf16(x), h2 = bf16(x - h1), h3 = bf16(x - h1 -
// h2): each residual is exactly representable, so the sum is exact, 3 x 8 = 24 significand bits), formed with three
// v_cvt_pk_bf16_f32 per pair while the tile is staged into LDS as three bf16 planes.  The product keeps the six terms
// a_i.b_j with i + j <= 4 (a1b1, a1b2, a2b1, a1b3, a2b2, a3b1); the three dropped ones are below 2^-24 |a||b|, the size of
// the rounding an fp32 FMA makes itself.  Each bf16 x bf16 product is exact in fp32 and v_mfma_f32_32x32x16_bf16
// accumulates in fp32, so the result carries the error of an fp32 GEMM with another summation order -- at 6 MFMAs of the
// bf16 pipe (16x the fp32 matrix rate) per k-step: peak 2.5 PF / 6 = 417 TFLOP/s fp32-equivalent vs 157.3 for
// v_mfma_f32_32x32x2_f32.  tests/test_gpu_gemm_split.py holds it to the exact-fp32 kernel's error against float64.
// 128x128 block tile, BK = 16 (one MFMA k-step), 4 waves x 64x64, double-buffered planes (72 KB: two workgroups per CU),
// 48-byte LDS rows (conflict-free ds_read_b128 fragments).  Whole tiles only; other shapes use the exact fp32 kernel.
// The kernel is generic in the number of bf16 planes per operand value:
//   NP = 3 (BK 16): x = h1 + h2 + h3 exactly, six products  -> fp32-accurate            ("split3", the fp32 default)
//   NP = 2 (BK 16): x ~ h1 + h2 (16 significand bits), products a1b1 + a1b2 + a2b1       ("bf16x2": ~2^-16 relative)
//   NP = 1 (BK 32): x ~ bf16(x), one product                                             ("bf16", BASELINE config 3)
constexpr int BKS = 16;                     // k-tile of the NP = 2, 3 instantiations (and the granularity K must divide by)

// Splitting fp32 into bf16 planes costs VALU issue slots next to the MFMAs (PMC: the matrix pipe is idle most of the time
// because the waves are issuing this).  The residual x - h of a bf16 rounding is ONE instruction with v_dot2_f32_bf16
// (h.lo * c.lo + h.hi * c.hi + x with c = (-1, 0) or (0, -1)) instead of shift / mask + subtract, and exact because the
// residual is representable in fp32 (scratch probe on the GPU: bit-equal to the subtract, random and tiny values).  A DOT
// result read by another VALU instruction needs 3 wait states that the hardware does NOT interlock, and the compiler cannot
// see into inline asm (a naive asm version computed garbage; the builtin version was hazard-safe but paid a v_mov per value
// for the two-address v_dot2c form).  So four pairs (the 8 consecutive k one thread stages per operand) are split together in
// three asm phases whose instruction order satisfies the wait states by construction: every cvt reads residuals written at
// least 3 instructions earlier, whatever the compiler schedules between the phases.  7 VALU per pair instead of 11.
struct Split4 {
    float r[8];
    // p0 = bf16(x), r = x - p0
    __device__ __forceinline__ void phase1(const float* x, uint32_t* p0) {
        asm("v_cvt_pk_bf16_f32 %0, %12, %13\n\tv_cvt_pk_bf16_f32 %1, %14, %15\n\t"
            "v_cvt_pk_bf16_f32 %2, %16, %17\n\tv_cvt_pk_bf16_f32 %3, %18, %19\n\t"
            "v_dot2_f32_bf16 %4, %0, %20, %12\n\tv_dot2_f32_bf16 %5, %0, %21, %13\n\t"
            "v_dot2_f32_bf16 %6, %1, %20, %14\n\tv_dot2_f32_bf16 %7, %1, %21, %15\n\t"
            "v_dot2_f32_bf16 %8, %2, %20, %16\n\tv_dot2_f32_bf16 %9, %2, %21, %17\n\t"
            "v_dot2_f32_bf16 %10, %3, %20, %18\n\tv_dot2_f32_bf16 %11, %3, %21, %19"
            : "=&v"(p0[0]), "=&v"(p0[1]), "=&v"(p0[2]), "=&v"(p0[3]),
              "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
            : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]),
              "s"(0x0000bf80u), "s"(0xbf800000u));
    }
    // p1 = bf16(r), r -= p1
    __device__ __forceinline__ void phase2(uint32_t* p1) {
        asm("v_cvt_pk_bf16_f32 %0, %4, %5\n\tv_cvt_pk_bf16_f32 %1, %6, %7\n\t"
            "v_cvt_pk_bf16_f32 %2, %8, %9\n\tv_cvt_pk_bf16_f32 %3, %10, %11\n\t"
            "v_dot2_f32_bf16 %4, %0, %12, %4\n\tv_dot2_f32_bf16 %5, %0, %13, %5\n\t"
            "v_dot2_f32_bf16 %6, %1, %12, %6\n\tv_dot2_f32_bf16 %7, %1, %13, %7\n\t"
            "v_dot2_f32_bf16 %8, %2, %12, %8\n\tv_dot2_f32_bf16 %9, %2, %13, %9\n\t"
            "v_dot2_f32_bf16 %10, %3, %12, %10\n\tv_dot2_f32_bf16 %11, %3, %13, %11"
            : "=&v"(p1[0]), "=&v"(p1[1]), "=&v"(p1[2]), "=&v"(p1[3]),
              "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
            : "s"(0x0000bf80u), "s"(0xbf800000u));
    }
    // p = bf16(r): the last plane
    __device__ __forceinline__ void phase3(uint32_t* p2) {
        asm("v_cvt_pk_bf16_f32 %0, %4, %5\n\tv_cvt_pk_bf16_f32 %1, %6, %7\n\t"
            "v_cvt_pk_bf16_f32 %2, %8, %9\n\tv_cvt_pk_bf16_f32 %3, %10, %11"
            : "=&v"(p2[0]), "=&v"(p2[1]), "=&v"(p2[2]), "=&v"(p2[3])
            : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]));
    }
};

template <int NP>
__device__ __forceinline__ void split_pk(float x0, float x1, uint32_t* p) {      // p[0..NP-1]: packed bf16 pairs, plane by plane
    p[0] = cvt_pk_bf16(x0, x1);                                                  // (prologue tile and NP = 1 only)
    if (NP > 1) {
        float r0 = x0 - __uint_as_float(p[0] << 16);
        float r1 = x1 - __uint_as_float(p[0] & 0xffff0000u);
        p[1] = cvt_pk_bf16(r0, r1);
        if (NP > 2) {
            r0 -= __uint_as_float(p[1] << 16);
            r1 -= __uint_as_float(p[1] & 0xffff0000u);
            p[2] = cvt_pk_bf16(r0, r1);
        }
    }
}

// One thread stages BKT/2 consecutive k of one tile row (256 threads: 128 rows x two k-halves).
// rows: valid rows of this operand from m0 on (>= 128 for a whole tile; the A operand of a product whose M is not a
// multiple of 128 stages zeros for the rows past M -- e.g. the layer-1 weight gradient, M = 80 features)
template <bool KMAJOR, int BKT>
struct StagerP {
    static constexpr int NV = BKT / 2;
    float v[NV];
    static constexpr int PIECES = KMAJOR ? NV : NV / 4;          // memory instructions per k-tile
    // one memory instruction of load(): issued between the MFMAs of the tile before (see the kernel's pipeline comment)
    __device__ __forceinline__ void load_piece(int j, const float* P, int ld, int m0, int k0, int tid, int rows = 128) {
        if (KMAJOR) {
            const bool ok = rows >= 128 || (tid & 127) < rows;
            const int kh = __builtin_amdgcn_readfirstlane(tid >> 7);
            const float* base = P + (size_t)(k0 + kh * NV) * ld + m0;
            const unsigned col = ok ? (unsigned)(tid & 127) : 0u;
            v[j] = (base + (size_t)j * ld)[col];         // rows past `rows`: a valid address, zeroed by the consumer (row_ok):
        } else {                                         // a select here would wait for the load in the middle of the MFMAs
            const bool ok = rows >= 128 || (tid >> 1) < rows;
            const float4* p = reinterpret_cast<const float4*>(P + (size_t)(m0 + (ok ? (tid >> 1) : 0)) * ld + k0 + (tid & 1) * NV);
            const float4 x = p[j];
            v[4 * j] = x.x; v[4 * j + 1] = x.y; v[4 * j + 2] = x.z; v[4 * j + 3] = x.w;
        }
    }
    static __device__ __forceinline__ bool row_ok(int tid, int rows) { return rows >= 128 || (KMAJOR ? (tid & 127) : (tid >> 1)) < rows; }
    __device__ __forceinline__ void load(const float* P, int ld, int m0, int k0, int tid, int rows = 128) {
        if (KMAJOR) {
            // row base on the scalar unit (k-half is uniform per wave), the lane's column as a 32-bit offset: one
            // global_load_dword with an SGPR base per k instead of a 64-bit VALU address each
            const bool ok = rows >= 128 || (tid & 127) < rows;
            const int kh = __builtin_amdgcn_readfirstlane(tid >> 7);
            const float* base = P + (size_t)(k0 + kh * NV) * ld + m0;
            const unsigned col = ok ? (unsigned)(tid & 127) : 0u;
#pragma unroll
            for (int j = 0; j < NV; ++j) { const float x = (base + (size_t)j * ld)[col]; v[j] = ok ? x : 0.f; }
        } else {
            const bool ok = rows >= 128 || (tid >> 1) < rows;
            const float4* p = reinterpret_cast<const float4*>(P + (size_t)(m0 + (ok ? (tid >> 1) : 0)) * ld + k0 + (tid & 1) * NV);
#pragma unroll
            for (int j = 0; j < NV / 4; ++j) {
                const float4 x = p[j];
                v[4 * j] = ok ? x.x : 0.f; v[4 * j + 1] = ok ? x.y : 0.f; v[4 * j + 2] = ok ? x.z : 0.f; v[4 * j + 3] = ok ? x.w : 0.f;
            }
        }
    }
};

// PARTM: M is not a multiple of 128 -- the A operand stages zeros for the rows past M (only that instantiation pays the
// selects) and the epilogue skips them.
// SK ("stream-K", Osama et al. 2023 restated for this kernel): the grid is what the chip holds at once (two workgroups per CU) and
// every workgroup multiplies the SAME number of k-tiles (+-1) -- a contiguous run of the (tile, k-tile) sequence, so a run covers
// a tail of one output tile, some whole tiles and a head of another.  With whole tiles per workgroup, T tiles on 512 slots cost
// ceil(T / 512) rounds (the encoder's products have 25 * 2^n tiles: 1 600 -> 4 rounds for 3.125 of work, 200 -> 1 for 0.39).  A
// workgroup whose run ends inside a tile leaves its accumulators in a workspace slot and raises a flag; the workgroup that
// reaches the tile's last k-tile adds the slots of the (lower-numbered, hence earlier dispatched: no circular wait) workgroups
// before it in ascending order and writes C -- a FIXED summation order: results are reproducible run to run, like the
// whole-tile kernel's.  Each XCD's 64 workgroups share one eighth of the tiles (blockIdx & 7 = XCD under round-robin
// dispatch), so no run crosses XCDs and operand tiles stay in one L2 as before.
template <bool TA, bool TB, bool PARTM, int NP, int BKT, bool SK = false>
__global__ __launch_bounds__(256, 2) void gemm_planes_kernel(GemmArgs a) {
    constexpr int BM = 128, BN = 128;
    constexpr int PITCH = BKT + 8;                  // bf16 elements: 48- / 80-byte rows, conflict-free ds_read_b128 fragments
    constexpr int PLANE = 128 * PITCH;
    constexpr int NV = BKT / 2, NPAIR = NV / 2;     // values / packed pairs one thread stages per operand and k-tile
    constexpr int KSTEPS = BKT / 16;
    __shared__ __attribute__((aligned(16))) unsigned short As[2 * NP * PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2 * NP * PLANE];
    __shared__ int sk_lost;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    a.A += (size_t)blockIdx.z * a.sA; a.B += (size_t)blockIdx.z * a.sB; a.C += (size_t)blockIdx.z * a.sC;
    const int ntn = a.N / BN, ntm = (a.M + BM - 1) / BM;
    const int nwg = ntn * ntm;
    int bid = blockIdx.x, ksl = blockIdx.y;
    const int nk_all = a.K / BKT;
    // SK: this workgroup's run of units [su, su1) of its XCD's share (unit = one k-tile of one tile; tiles in order)
    const int sk_g8 = gridDim.x >> 3, sk_li = blockIdx.x >> 3;
    const long long sk_u8 = (long long)(nwg >> 3) * nk_all;
    int su = 0, su1 = 0;
    if (SK) {
        su = (int)(sk_u8 * sk_li / sk_g8); su1 = (int)(sk_u8 * (sk_li + 1) / sk_g8);
        ksl = 0;
    } else if (a.xcd_split) {          // weight-gradient form: whole K slices per XCD (see gemm_f32_kernel)
        const int xcd = bid & 7, idx = bid >> 3;
        ksl = xcd + 8 * (idx / nwg);
        bid = idx % nwg;
    } else {
        const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    bool sk_first = true;
    do {
    int sk_kt0 = 0, sk_nk = 0, sk_tl = 0;
    if (SK) {       // next segment of the run, LAST FIRST: (tile, [kt0, nk)).  The head of a tile this run ends in is what another
        if (su >= su1) break;                   // workgroup waits for, so it is multiplied first; the tail of the tile the run
        sk_tl = (su1 - 1) / nk_all;             // begins in -- where this workgroup waits for others -- comes last
        sk_nk = su1 - sk_tl * nk_all; sk_kt0 = max(su - sk_tl * nk_all, 0);
        su1 -= sk_nk - sk_kt0;
        bid = (blockIdx.x & 7) * (nwg >> 3) + sk_tl;
        if (!sk_first) __syncthreads();          // the last tile's fragment reads are done before the planes are overwritten
        sk_first = false;
    }
    const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // Software pipeline, three k-tiles deep: while tile t is multiplied out of LDS buffer t&1, tile t+1 (global loads issued
    // one iteration ago, long since landed) is split and written to the other buffer in four slices BETWEEN the MFMA groups --
    // an MFMA holds the SIMD's issue port for 8 of its 32 cycles, the split's VALU work fits into the rest -- and the global
    // loads of tile t+2 are issued at the top.  Two register sets (sa0/sb0, sa1/sb1) alternate, hence the loop unrolled by 2.
    StagerP<TA, BKT> sa0, sa1;       // A natural [M,K]: k contiguous unless transposed
    StagerP<!TB, BKT> sb0, sb1;      // B natural [K,N]: n contiguous unless transposed
    const int per = (nk_all + a.splits - 1) / a.splits;
    const int kt0 = SK ? sk_kt0 : ksl * per, nk = SK ? sk_nk : min(nk_all, kt0 + per);
    if (!SK && kt0 >= nk) return;
    const int arows = PARTM ? a.M - m0 : 128;   // >= 128 except in the last row of tiles of a partial-M product
    const int rowA = TA ? (tid & 127) : (tid >> 1), khA = TA ? (tid >> 7) : (tid & 1);
    const int rowB = !TB ? (tid & 127) : (tid >> 1), khB = !TB ? (tid >> 7) : (tid & 1);
    unsigned short* const dA = As + rowA * PITCH + khA * NV;
    unsigned short* const dB = Bs + rowB * PITCH + khB * NV;
    const unsigned short* const apb = As + (wr * 64 + (lane & 31)) * PITCH + 8 * (lane >> 5);
    const unsigned short* const bpb = Bs + (wc * 64 + (lane & 31)) * PITCH + 8 * (lane >> 5);
    // split registers v[0..NV) into the planes of buffer `buf` (prologue: all at once)
    auto stage_all = [&](unsigned short* d, const float* v) {
        uint32_t pk[NP][NPAIR];
#pragma unroll
        for (int j = 0; j < NPAIR; ++j) { uint32_t t[NP]; split_pk<NP>(v[2 * j], v[2 * j + 1], t);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) pk[pl][j] = t[pl]; }
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int c = 0; c < NPAIR / 4; ++c)
                *reinterpret_cast<uint4*>(d + pl * PLANE + c * 8) = make_uint4(pk[pl][4 * c], pk[pl][4 * c + 1], pk[pl][4 * c + 2], pk[pl][4 * c + 3]);
    };
    sa0.load(a.A, a.lda, m0, kt0 * BKT, tid, arows);
    sb0.load(a.B, a.ldb, n0, kt0 * BKT, tid);
    if (kt0 + 1 < nk) { sa1.load(a.A, a.lda, m0, (kt0 + 1) * BKT, tid, arows); sb1.load(a.B, a.ldb, n0, (kt0 + 1) * BKT, tid); }
    stage_all(dA, sa0.v);
    stage_all(dB, sb0.v);
    __syncthreads();
    // one iteration: multiply buffer `cur`; if `stage`, split registers (va, vb) into buffer cur^1 along the way
    // The global loads of the tile after next are issued a few memory instructions per MFMA gap, from the second MFMA of this
    // tile on, instead of in one block between the barrier and the fragment reads (stamps: that block took 260-720 cycles
    // per k-tile, during which this wave had no MFMA in flight; dKx TN 165 -> 142 us, 4096^3 822 -> 796 us).
    auto step = [&](int cur, const float* va_, const float* vb, bool stage, auto& la, auto& lb, int kload, bool doload) {
        int slot = 0;
        auto ld = [&]() {                               // called after every MFMA; slot is a compile-time value after unrolling
            constexpr int PA = StagerP<TA, BKT>::PIECES, PB = StagerP<!TB, BKT>::PIECES;
            constexpr int NSLOT = KSTEPS * 4 * (NP == 3 ? 6 : NP == 2 ? 3 : 1);
            constexpr int PPS = (PA + PB + NSLOT - 2) / (NSLOT - 1);            // pieces per gap
            if (doload && slot >= 1) {
#pragma unroll
                for (int e = 0; e < PPS; ++e) {
                    const int pc = (slot - 1) * PPS + e;
                    if (pc < PA) la.load_piece(pc, a.A, a.lda, m0, kload * BKT, tid, arows);
                    else if (pc < PA + PB) lb.load_piece(pc - PA, a.B, a.ldb, n0, kload * BKT, tid);
                }
                if ((slot - 1) * PPS < PA + PB) __builtin_amdgcn_sched_barrier(0);
            }
            ++slot;
        };
        const unsigned short* ap = apb + cur * (NP * PLANE);
        const unsigned short* bp = bpb + cur * (NP * PLANE);
        unsigned short* wa = dA + (cur ^ 1) * (NP * PLANE);
        unsigned short* wb = dB + (cur ^ 1) * (NP * PLANE);
        uint32_t pa[NP][NPAIR], pb[NP][NPAIR];
        float xa[NV];                               // PARTM: rows past M stage zeros (the loads above fetched a valid row)
        if (PARTM) {
            const bool okA = StagerP<TA, BKT>::row_ok(tid, arows);
#pragma unroll
            for (int e = 0; e < NV; ++e) xa[e] = okA ? va_[e] : 0.f;
        }
        const float* va = PARTM ? xa : va_;
        Split4 qa, qb;                              // NP >= 2 (NPAIR = 4): the phased split, riding on the MFMA groups
        constexpr int PPG = NPAIR / 4;              // NP = 1: pairs per operand riding on each of the 4 MFMA groups
        auto put = [&](unsigned short* w, uint32_t (*pk)[NPAIR], int pl) {
#pragma unroll
            for (int c = 0; c < NPAIR / 4; ++c)
                *reinterpret_cast<uint4*>(w + pl * PLANE + c * 8) = make_uint4(pk[pl][4 * c], pk[pl][4 * c + 1], pk[pl][4 * c + 2], pk[pl][4 * c + 3]);
        };
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            bf16x8 af[2][NP], bf[2][NP];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    af[i][pl] = *reinterpret_cast<const bf16x8*>(ap + pl * PLANE + i * 32 * PITCH + ks * 16);
                    bf[i][pl] = *reinterpret_cast<const bf16x8*>(bp + pl * PLANE + i * 32 * PITCH + ks * 16);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {           // smallest terms first
                    const int q = 2 * i + j;
                    if (NP == 3) { acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0); ld(); }
                    if (stage && NP == 1 && ks == 0) {
#pragma unroll
                        for (int e = 0; e < PPG; ++e) { uint32_t t[NP]; split_pk<NP>(va[2 * (q * PPG + e)], va[2 * (q * PPG + e) + 1], t);
#pragma unroll
                            for (int pl = 0; pl < NP; ++pl) pa[pl][q * PPG + e] = t[pl]; }
                    }
                    if (stage && NP >= 2) {             // A: one phase per MFMA group; the planes go to LDS as soon as they exist
                        __builtin_amdgcn_sched_barrier(0);      // (keep the phases BETWEEN the MFMAs: the scheduler would
                        if (q == 0) qa.phase1(va, pa[0]);       //  otherwise hoist all MFMAs and leave the split for the end)
                        if (q == 1) { if (NP == 3) qa.phase2(pa[1]); else { qa.phase3(pa[NP - 1]); put(wa, pa, 0); put(wa, pa, 1); } }
                        if (q == 2 && NP == 3) { qa.phase3(pa[NP - 1]); put(wa, pa, 0); put(wa, pa, 1); }
                        if (q == 3 && NP == 3) put(wa, pa, 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (NP == 3) { acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0); ld(); }
                    if (NP == 3) { acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0); ld(); }
                    if (stage && NP == 1 && ks == KSTEPS - 1) {
#pragma unroll
                        for (int e = 0; e < PPG; ++e) { uint32_t t[NP]; split_pk<NP>(vb[2 * (q * PPG + e)], vb[2 * (q * PPG + e) + 1], t);
#pragma unroll
                            for (int pl = 0; pl < NP; ++pl) pb[pl][q * PPG + e] = t[pl]; }
                    }
                    if (stage && NP >= 2) {             // B likewise
                        __builtin_amdgcn_sched_barrier(0);
                        if (q == 0) qb.phase1(vb, pb[0]);
                        if (q == 1) { if (NP == 3) qb.phase2(pb[1]); else { qb.phase3(pb[NP - 1]); put(wb, pb, 0); put(wb, pb, 1); } }
                        if (q == 2 && NP == 3) { qb.phase3(pb[NP - 1]); put(wb, pb, 0); put(wb, pb, 1); }
                        if (q == 3 && NP == 3) put(wb, pb, 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (NP >= 2) { acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0); ld(); }
                    if (NP >= 2) { acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0); ld(); }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0); ld();
                }
        }
        if (stage && NP == 1) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) { put(wa, pa, pl); put(wb, pb, pl); }
        }
    };
    int kt = kt0;
    for (; kt + 2 < nk; kt += 2) {
        // even tile: multiply buffer 0, stage set 1 (tile kt+1) into buffer 1, load tile kt+2 into set 0
        step(0, sa1.v, sb1.v, true, sa0, sb0, kt + 2, true);
        __syncthreads();
        // odd tile: multiply buffer 1, stage set 0 (tile kt+2) into buffer 0, load tile kt+3 into set 1
        step(1, sa0.v, sb0.v, true, sa1, sb1, min(kt + 3, nk - 1), true);    // (past the end: a valid tile again, unused -- no branch per load)
        __syncthreads();
    }
    // tail: one or two tiles left; buffer 0 holds tile kt, set 1 (if any) tile kt+1
    if (kt + 1 < nk) {
        step(0, sa1.v, sb1.v, true, sa0, sb0, 0, false);
        __syncthreads();
        step(1, sa0.v, sb0.v, false, sa1, sb1, 0, false);
    } else {
        step(0, sa0.v, sb0.v, false, sa1, sb1, 0, false);
    }
    // Slots and flags cross workgroups (possibly XCDs) inside one launch: agent-scope accesses (sc1: stores write through, loads
    // are served from the coherent level) instead of fences -- a buffer_inv / buffer_wbl2 costs every workgroup of the XCD its
    // L2 contents (first version, with acquire polls: +150 us per launch).  A store is acknowledged (vmcnt) once it is there.
    typedef float f32x4s __attribute__((ext_vector_type(4)));
    if (SK && nk < nk_all) {            // the run ends inside this tile: accumulators -> this workgroup's slot, then the flag
        float* wp = a.sk_ws + (size_t)blockIdx.x * (BM * BN) + tid * 64;          // 256 B per thread: one base, immediate offsets
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x4s v = {acc[q >> 1][q & 1][4 * r], acc[q >> 1][q & 1][4 * r + 1], acc[q >> 1][q & 1][4 * r + 2], acc[q >> 1][q & 1][4 * r + 3]};
                asm volatile("global_store_dwordx4 %0, %1, off offset:%2 sc1" :: "v"(wp), "v"(v), "n"((q * 4 + r) * 16) : "memory");
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.sk_flag + blockIdx.x, a.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;
    }
    if (SK && kt0 > 0) {                // the tile began in earlier workgroups of this XCD's share: add their slots, lowest first
        const long long ub = (long long)sk_tl * nk_all;         // the tile's first unit
        int c = sk_li - 1;
        while (sk_u8 * c / sk_g8 > ub) --c;
        for (; c < sk_li; ++c) {
            const int cb = c * 8 + (blockIdx.x & 7);
            if (tid == 0) {
                int guard = 0;
                while (__hip_atomic_load(a.sk_flag + cb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.sk_epoch && ++guard < (1 << 24))
                    __builtin_amdgcn_s_sleep(8);
                sk_lost = guard >= (1 << 24);       // (seconds: the slot never came -- the tile is written as NaN, never silently short)
            }
            __syncthreads();
            const bool lost = sk_lost != 0;
            __syncthreads();                        // (thread 0 writes the word again for the next slot)
            if (lost) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[q >> 1][q & 1][e] = __builtin_nanf("");
                break;
            }
            const float* rp = a.sk_ws + (size_t)cb * (BM * BN) + tid * 64;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4s v0, v1, v2, v3;
                asm volatile("global_load_dwordx4 %0, %4, off offset:%5 sc1\n\tglobal_load_dwordx4 %1, %4, off offset:%6 sc1\n\t"
                             "global_load_dwordx4 %2, %4, off offset:%7 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:%8 sc1\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                             : "v"(rp), "n"(q * 64), "n"(q * 64 + 16), "n"(q * 64 + 32), "n"(q * 64 + 48) : "memory");
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[q >> 1][q & 1][e] += v0[e]; acc[q >> 1][q & 1][4 + e] += v1[e];
                    acc[q >> 1][q & 1][8 + e] += v2[e]; acc[q >> 1][q & 1][12 + e] += v3[e];
                }
            }
        }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int n = n0 + wc * 64 + ni * 32 + (lane & 31);
            const float bv = (a.bias && ksl == 0) ? a.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (PARTM && m >= a.M) continue;
                float* cp = a.C + (size_t)m * a.ldc + n;
                float v = acc[mi][ni][r] + bv;
                if (a.splits > 1) {
                    if (a.slab) a.slab[((size_t)(blockIdx.z * a.splits + ksl) * a.M + m) * a.slab_ld + n] = v;
                    else atomicAdd(cp, v);
                }
                else { if (a.accumulate) v += *cp; *cp = v; }
            }
        }
    } while (SK);
}

// ---------------------------------------------------------------------------------------------
// split3 on 64x64 block tiles: the decoder's products (M = T*B = 3840 rows, N = 128..768, K = 256..1024; 60 tiles of 128x128
// would leave three quarters of the chip idle, so they ran on the 64x64 instantiation of the fp32-input MFMA kernel: 8
// dependent v_mfma_f32_32x32x2_f32 = 512 cycles per k-tile and wave).  Same arithmetic as gemm_planes_kernel<..., 3, 16> (six
// bf16 products of exact three-way splits, fp32 accumulate): 6 x 32 cycles per k-tile and wave.  Four waves x 32x32; waves 0-1
// stage the A tile, waves 2-3 the B tile (one Split4 per thread and k-tile); K need not be a multiple of 16 (V = 1000 is the
// contraction length of dLogits . W_out^T): the last k-tile stages zeros past K.  Whole 64x64 tiles only.
template <bool TA, bool TB>
__global__ __launch_bounds__(256, 4) void gemm_split3s_kernel(GemmArgs a) {
    constexpr int BT = 64, BKT = 16, NV = 8;
    constexpr int PITCH = BKT + 8;                  // 48-byte rows (conflict-free ds_read_b128 fragments, as the 128x128 kernel)
    constexpr int PLANE = BT * PITCH;
    constexpr int BUF = 3 * PLANE;
    __shared__ __attribute__((aligned(16))) unsigned short As[2 * BUF];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    a.A += (size_t)blockIdx.z * a.sA; a.B += (size_t)blockIdx.z * a.sB; a.C += (size_t)blockIdx.z * a.sC;
    const int ntn = (a.N + BT - 1) / BT, ntm = (a.M + BT - 1) / BT;
    const int nwg = ntn * ntm;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    const int m0 = (bid / ntn) * BT, n0 = (bid % ntn) * BT;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int nk = (a.K + BKT - 1) / BKT;
    // this thread's share of the staging: waves 0-1 -> A, waves 2-3 -> B (uniform per wave); t = index among the 128 stagers
    const bool isB = __builtin_amdgcn_readfirstlane(wid >> 1) != 0;
    const int t = tid & 127;
    const bool kmaj = isB ? !TB : TA;               // operand stored with the tile's row index contiguous
    const int row = kmaj ? (t & 63) : (t >> 1), kh = kmaj ? (t >> 6) : (t & 1);
    const float* const P = isB ? a.B : a.A;
    const int ld = isB ? a.ldb : a.lda;
    const int rc = min((isB ? n0 : m0) + row, (isB ? a.N : a.M) - 1);     // rows / columns past the matrix: a valid address; the
                                                                          // epilogue never stores what they produce
    unsigned short* const dW = (isB ? Bs : As) + row * PITCH + kh * NV;
    const unsigned short* const apb = As + (wr * 32 + (lane & 31)) * PITCH + 8 * (lane >> 5);
    const unsigned short* const bpb = Bs + (wc * 32 + (lane & 31)) * PITCH + 8 * (lane >> 5);
    float x[NV];
    // the 8 values (row, k0 + kh*8 .. +7) of this thread; k >= K reads as zero (last tile only: `tail`)
    auto load_piece = [&](int j, int k0, bool tail) {          // j: 0..7 (row-contiguous operand) or 0..1 (k-contiguous)
        if (kmaj) {
            const int k = k0 + kh * NV + j;
            const bool ok = !tail || k < a.K;
            const float v = (P + (size_t)(ok ? k : 0) * ld)[rc];
            x[j] = ok ? v : 0.f;
        } else {
            const int k = k0 + kh * NV + 4 * j;
            const bool ok = !tail || k < a.K;                  // (K % 4 == 0 is a condition of this kernel)
            const float4 v = *reinterpret_cast<const float4*>(P + (size_t)rc * ld + (ok ? k : 0));
            x[4 * j] = ok ? v.x : 0.f; x[4 * j + 1] = ok ? v.y : 0.f; x[4 * j + 2] = ok ? v.z : 0.f; x[4 * j + 3] = ok ? v.w : 0.f;
        }
    };
    auto load_all = [&](int k0, bool tail) {
        if (kmaj) {
#pragma unroll
            for (int j = 0; j < 8; ++j) load_piece(j, k0, tail);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) load_piece(j, k0, tail);
        }
    };
    auto put3 = [&](unsigned short* w, uint32_t (*pk)[4]) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint4*>(w + pl * PLANE) = make_uint4(pk[pl][0], pk[pl][1], pk[pl][2], pk[pl][3]);
    };
    {   // prologue: tile 0 -> buffer 0; tile 1 in flight in x
        load_all(0, nk == 1);
        uint32_t pk[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { uint32_t tt[3]; split_pk<3>(x[2 * j], x[2 * j + 1], tt); pk[0][j] = tt[0]; pk[1][j] = tt[1]; pk[2][j] = tt[2]; }
        put3(dW, pk);
        if (nk > 1) load_all(BKT, nk == 2);
    }
    __syncthreads();
    // one k-tile: six MFMAs out of buffer cur; the split of tile kt+1 (registers x) rides in their gaps and lands in buffer
    // cur^1; x is then refilled with tile kt+2
    auto step = [&](int cur, bool stage, bool reload, int kload, bool tail) {
        const unsigned short* ap = apb + cur * BUF;
        const unsigned short* bp = bpb + cur * BUF;
        unsigned short* w = dW + (cur ^ 1) * BUF;
        bf16x8 af[3], bf[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { af[pl] = *reinterpret_cast<const bf16x8*>(ap + pl * PLANE); bf[pl] = *reinterpret_cast<const bf16x8*>(bp + pl * PLANE); }
        Split4 sp;
        uint32_t pk[3][4];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], acc, 0, 0, 0);
        if (stage) { __builtin_amdgcn_sched_barrier(0); sp.phase1(x, pk[0]); __builtin_amdgcn_sched_barrier(0); }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], acc, 0, 0, 0);
        if (stage) { __builtin_amdgcn_sched_barrier(0); sp.phase2(pk[1]); __builtin_amdgcn_sched_barrier(0); }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc, 0, 0, 0);
        if (stage) { __builtin_amdgcn_sched_barrier(0); sp.phase3(pk[2]); put3(w, pk); __builtin_amdgcn_sched_barrier(0); }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc, 0, 0, 0);
        if (reload) { __builtin_amdgcn_sched_barrier(0); load_all(kload * BKT, tail); __builtin_amdgcn_sched_barrier(0); }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc, 0, 0, 0);
    };
    int kt = 0;
    for (; kt + 3 < nk; ++kt) {                       // steady state: tile kt+2 is not the last one
        step(kt & 1, true, true, kt + 2, false);
        __syncthreads();
    }
    for (; kt + 1 < nk; ++kt) {                       // the last refill may be the partial tile (or there is none)
        step(kt & 1, true, kt + 2 < nk, kt + 2, true);
        __syncthreads();
    }
    step(kt & 1, false, false, 0, false);
    const int n = n0 + wc * 32 + (lane & 31);
    if (n >= a.N) return;
    const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= a.M) continue;
        float* cp = a.C + (size_t)m * a.ldc + n;
        float v = acc[r] + bv;
        if (a.accumulate) v += *cp;
        *cp = v;
    }
}

// workspace of the stream-K launches of one stream: a partial tile and a flag per workgroup (allocated at the stream's first use)
struct SkWorkspace { float* ws; int* flag; int* epoch; };      // epoch: launches so far (a flag is raised by writing the launch's number: no clearing)
constexpr int SK_MAX_WGS = 1024;
static SkWorkspace sk_workspace(hipStream_t s) {
    static std::mutex mu;
    static std::map<hipStream_t, SkWorkspace> all;
    std::lock_guard<std::mutex> lk(mu);
    auto it = all.find(s);
    if (it != all.end()) return it->second;
    SkWorkspace w{nullptr, nullptr, nullptr};
    if (hipMalloc(&w.ws, (size_t)SK_MAX_WGS * 128 * 128 * sizeof(float)) != hipSuccess ||
        hipMalloc(&w.flag, SK_MAX_WGS * sizeof(int)) != hipSuccess ||
        hipMemset(w.flag, 0, SK_MAX_WGS * sizeof(int)) != hipSuccess) { w.ws = nullptr; w.flag = nullptr; (void)hipGetLastError(); return w; }
    w.epoch = new int(0);
    all[s] = w;
    return w;
}

template <int NP, int BKT>
static void launch_planes_sk(dim3 grid, hipStream_t s, const GemmArgs& g, int transB) {
    if (transB) hipLaunchKernelGGL((gemm_planes_kernel<false, true, false, NP, BKT, true>), grid, dim3(256), 0, s, g);
    else        hipLaunchKernelGGL((gemm_planes_kernel<false, false, false, NP, BKT, true>), grid, dim3(256), 0, s, g);
}

template <int NP, int BKT>
static void launch_planes(dim3 grid, hipStream_t s, const GemmArgs& g, int transA, int transB, bool partm) {
    if (transA && partm) hipLaunchKernelGGL((gemm_planes_kernel<true, false, true, NP, BKT>), grid, dim3(256), 0, s, g);
    else if (transA)     hipLaunchKernelGGL((gemm_planes_kernel<true, false, false, NP, BKT>), grid, dim3(256), 0, s, g);
    else if (transB)     hipLaunchKernelGGL((gemm_planes_kernel<false, true, false, NP, BKT>), grid, dim3(256), 0, s, g);
    else                 hipLaunchKernelGGL((gemm_planes_kernel<false, false, false, NP, BKT>), grid, dim3(256), 0, s, g);
}

static int g_gemm_bf16 = 0;
// fp32 products of whole tiles on the bf16 pipe by exact 3-way splitting (default on; ASR_GEMM_SPLIT=0 or
// asr_set_gemm_split(0) selects v_mfma_f32_32x32x2_f32 everywhere)
static int g_gemm_split = [] { const char* e = getenv("ASR_GEMM_SPLIT"); return e ? (atoi(e) != 0) : 1; }();
}  // namespace asr

// ---------------------------------------------------------------------------------
// C ABI (declared in include/e2e_asr_hip.h)
// ---------------------------------------------------------------------------------
extern "C" int asr_gemm_f32_batched(void* stream, int transA, int transB, int M, int N, int K,
                                    const float* A, int lda, long long strideA, const float* B, int ldb, long long strideB,
                                    float* C, int ldc, long long strideC, const float* bias, int accumulate, int batch);

// 0: fp32 (default; see asr_set_gemm_split for how whole-tile products are evaluated).  1: products made of whole tiles round
// their operands to bf16 on the way into LDS (one plane, one product; fp32 accumulate/output).  2: two bf16 planes per
// operand, three products (~2^-16 relative).  Products with partial tiles or few output tiles stay on the exact fp32 kernel.
extern "C" int asr_set_gemm_precision(int mode) {
    if (mode < 0 || mode > 2) return ASR_EINVAL;
    asr::g_gemm_bf16 = mode;
    return ASR_OK;
}
extern "C" int asr_get_gemm_precision(void) { return asr::g_gemm_bf16; }
extern "C" int asr_set_gemm_split(int on) { asr::g_gemm_split = on != 0; return ASR_OK; }
extern "C" int asr_get_gemm_split(void) { return asr::g_gemm_split; }

extern "C" int asr_gemm_f32(void* stream, int transA, int transB, int M, int N, int K,
                            const float* A, int lda, const float* B, int ldb,
                            float* C, int ldc, const float* bias, int accumulate) {
    return asr_gemm_f32_batched(stream, transA, transB, M, N, K, A, lda, 0, B, ldb, 0, C, ldc, 0, bias, accumulate, 1);
}

extern "C" int asr_gemm_f32_batched(void* stream, int transA, int transB, int M, int N, int K,
                                    const float* A, int lda, long long strideA, const float* B, int ldb, long long strideB,
                                    float* C, int ldc, long long strideC, const float* bias, int accumulate, int batch) {
    using namespace asr;
    if (batch <= 0) return ASR_EINVAL;
    if (M < 0 || N < 0 || K < 0 || !C || (K > 0 && (!A || !B))) return ASR_EINVAL;
    if (M == 0 || N == 0) return ASR_OK;
    if (lda < (transA ? M : K) || ldb < (transB ? K : N) || ldc < N) return ASR_EINVAL;
    // Weight-gradient form with a long K, whole row tiles and a ragged N (the decoder's OutputProjection gradient h^T . dlogits,
    // 256 x 1000 x 3840): the whole-tile split3 kernel takes N % 128 == 0 only and the product fell to the exact-fp32 MFMA kernel
    // with bounds checks (40 TF/s, 50-59 us per step; rocBLAS: 76 TF/s).  The TRANSPOSED product dlogits^T . h has the ragged
    // dimension in M, which the weight-gradient instantiation of the split3 kernel takes (partial last row of tiles): it goes to a
    // scratch [N][M] of this stream and a tiled transpose adds it into C.  (Splitting the columns into whole tiles + a ragged
    // rest was measured first: the 104-column rest alone -- 2 tiles x 15 K slices -- took 41 us.)
    if (transA && !transB && !bias && batch == 1 && (g_gemm_split || g_gemm_bf16 != 0) && N % 128 != 0 && N >= 64 && M % 128 == 0 &&
        K >= 1024 && K % BKS == 0 && lda % 4 == 0 && ldb % 4 == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0) {
        hipStream_t st = static_cast<hipStream_t>(stream);
        float* T = slab_arena(st, (size_t)N * M * sizeof(float), 1);
        if (T) {
            const int rc = asr_gemm_f32_batched(stream, 1, 0, N, M, K, B, ldb, 0, A, lda, 0, T, M, 0, nullptr, 0, 1);
            if (rc != ASR_OK) return rc;
            return transpose_add(st, C, ldc, T, M, N, accumulate);
        }
    }
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.accumulate = accumulate;
    g.sA = strideA; g.sB = strideB; g.sC = strideC;
    g.sk_ws = nullptr; g.sk_flag = nullptr; g.sk_epoch = 0;
    g.vecA = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (lda % 4 == 0) && (strideA % 4 == 0);
    g.vecB = ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (ldb % 4 == 0) && (strideB % 4 == 0);
    int nwg = ((M + 127) / 128) * ((N + 127) / 128);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // Few output tiles but a long K (weight gradients X^T.dY with K = B*T): split K over
    // blockIdx.y so the chip is filled; partial tiles meet in C through float atomics
    // (summation order, hence the last bits, may differ from run to run).
    int splits = 1;
    const int nk = (K + BK - 1) / BK;
    // (batched only when accumulating: the split needs a pre-zeroed or live C, and only batch 1 is zeroed here)
    const int tiles_all = nwg * batch;
    if (transA && (batch == 1 || accumulate) && tiles_all < 192 && nk >= 64) {     // weight-gradient form only: forward products stay bit-reproducible
        // as many workgroups as the chip holds at once (two per CU = 512): measured stand-alone on the batched weight-gradient
        // shapes (scripts/exp_splitk2.py), 16 x 32 slices beat 16 x 48 by 15 % and 32 x 16 beat 32 x 24 by 6 %; never fewer
        // slices than the old 1.5-round rule gave up to 8 (128 tiles stay at 8 slices pinned to the 8 XCDs)
        static const int wg_target = [] { const char* e = getenv("ASR_GEMM_WGTARGET"); return e ? atoi(e) : 512; }();
        const int old_rule = (768 + tiles_all - 1) / tiles_all;
        splits = std::min(std::max((wg_target + tiles_all - 1) / tiles_all, std::min(8, old_rule)), nk / 16);
        if (const char* e = getenv("ASR_GEMM_SPLITK")) splits = std::max(1, atoi(e));
    }
    g.splits = splits;
    g.xcd_split = 0;
    static const bool shape_log = getenv("ASR_GEMM_LOG") != nullptr;          // diagnostic: one line per product on stderr
    if (shape_log) fprintf(stderr, "gemm tA=%d tB=%d M=%d N=%d K=%d batch=%d splits=%d acc=%d stream=%p\n", transA, transB, M, N, K, batch, splits, accumulate, stream);
    // K slices meet either through float atomics in a pre-zeroed / live C (rounds 1-4; ASR_WGRAD_SLABS=0) or -- default -- as
    // partial tiles in this stream's slab arena, added into C in slice order by slab_reduce (csrc/splitk.hip): bit-reproducible
    g.slab = nullptr; g.slab_ld = 0;
    if (splits > 1 && wgrad_slabs()) {
        const int ldp = (N + 3) & ~3, smax = std::max(splits, (splits + 4) / 8 * 8);      // (the XCD-pinned form rounds the slices up to 8 n)
        g.slab = slab_arena(s, (size_t)batch * smax * M * ldp * sizeof(float));
        if (g.slab) g.slab_ld = ldp;
    }
    auto slab_end = [&](int bkt) -> int {       // behind the launch: C (+)= the slices in ascending order; bkt = the kernel's k-tile
        if (!g.slab) return ASR_OK;
        const int nk_all = (K + bkt - 1) / bkt, per = (nk_all + g.splits - 1) / g.splits, nsl = (nk_all + per - 1) / per;
        SlabMap q;
        q.M = M; q.N = g.slab_ld; q.Nvalid = N; q.nsl = nsl; q.nsl_alloc = g.splits; q.batch = batch; q.mA = M; q.mA_valid = M;
        q.colmap = nullptr; q.ldc = ldc; q.zC = strideC; q.accumulate = accumulate;
        return slab_reduce(s, C, g.slab, q);
    };
    if (splits > 1 && !accumulate && !g.slab) {
        if (hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, s) != hipSuccess) return ASR_ELAUNCH;
    }
    // bf16 matrix pipe (gemm_planes_kernel): fp32 mode with the exact 3-way split (NP = 3, fp32-accurate; default), or the
    // reduced-precision modes of asr_set_gemm_precision: 1 = one bf16 plane (config 3), 2 = two planes / three products
    const int np = g_gemm_bf16 == 1 ? 1 : (g_gemm_bf16 == 2 ? 2 : (g_gemm_split ? 3 : 0));
    if (np && (M % 128 == 0 || (transA && !transB && M >= 64)) && N % 128 == 0 && K % BKS == 0 && g.vecA && g.vecB &&
        !(transA && transB) && (np < 3 || splits > 1 || (long long)nwg * batch >= 96)) {
        // whole tiles (or, for the weight-gradient form, a partial last row of tiles); same split-K policy as the fp32 kernel
        static const int xs = [] { const char* e = getenv("ASR_GEMM_XCD_SPLIT"); return e ? atoi(e) : 1; }();
        const bool partm = M % 128 != 0;
        g.splits = splits;
        dim3 grid(nwg, splits, batch);
        if (transA && xs && splits >= 6 && nk >= 128 && (batch == 1 || (nwg * ((splits + 4) / 8 * 8)) % 8 == 0)) {
            g.splits = (splits + 4) / 8 * 8;
            g.xcd_split = 1;
            grid = dim3(nwg * g.splits, 1, batch);
        }
        // Stream-K (see the kernel): forward / data-gradient forms whose tiles fill the last round of workgroup slots badly.
        // OPT-IN (ASR_GEMM_SK=1), measured and not adopted: stand-alone it is 0-6 % ahead on the encoder's products with >= 800
        // tiles and 14 % behind at 400 (a lone workgroup on a CU multiplies a k-tile in 0.77 us, two sharing it in 0.73 us
        // each: the "rounds" a partial last round wastes are mostly not there to win back), and inside the train step the runs
        // that wait for a slot of a workgroup still queued behind the other stream's GEMM cost +0.19 ms (7.99 vs 7.80 ms).
        static const int sk_env = [] { const char* e = getenv("ASR_GEMM_SK"); return e ? atoi(e) : 0; }();      // (read once, not per launch)
        // stream-K numbers its launches on the HOST (a flag is raised by writing the launch's number): a captured graph would replay
        // a stale number, so it is refused while the stream is capturing (advisor, round 3)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        const bool capturing = sk_env && hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
        const int sk_on = capturing ? 0 : sk_env;
        static const int sk_slots = [] {
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
            return std::min(2 * cus, SK_MAX_WGS) & ~7;
        }();
        const int bkt = (np == 1 && K % 32 == 0) ? 32 : 16;
        const int nkp = K / bkt;
        if (sk_on && !transA && !partm && splits == 1 && batch == 1 && nwg % 8 == 0 && nkp >= 8 && sk_slots >= 8) {
            const int rounds = (nwg + sk_slots - 1) / sk_slots;
            const double waste = 1.0 - (double)nwg / ((double)rounds * sk_slots);
            const long long units = (long long)nwg * nkp;
            const int G = (int)std::min<long long>(sk_slots, (units / 8) & ~7LL);          // at least 8 k-tiles per workgroup
            if (waste >= 0.1 && G >= 8) {
                const SkWorkspace w = sk_workspace(s);
                if (w.ws) {
                    g.sk_ws = w.ws; g.sk_flag = w.flag; g.sk_epoch = ++*w.epoch;
                    if (np == 3)        launch_planes_sk<3, 16>(dim3(G), s, g, transB);
                    else if (np == 2)   launch_planes_sk<2, 16>(dim3(G), s, g, transB);
                    else if (bkt == 32) launch_planes_sk<1, 32>(dim3(G), s, g, transB);
                    else                launch_planes_sk<1, 16>(dim3(G), s, g, transB);
                    ASR_CHECK_LAUNCH();
                    return ASR_OK;
                }
            }
        }
        if (np == 3)                      launch_planes<3, 16>(grid, s, g, transA, transB, partm);
        else if (np == 2)                 launch_planes<2, 16>(grid, s, g, transA, transB, partm);
        else if (K % 32 == 0)             launch_planes<1, 32>(grid, s, g, transA, transB, partm);
        else                              launch_planes<1, 16>(grid, s, g, transA, transB, partm);
        ASR_CHECK_LAUNCH();
        return slab_end((np == 1 && K % 32 == 0) ? 32 : 16);
    }
    // few output tiles and no split-K: 64x64 block tiles fill the chip 4x better (still reproducible)
    static const int small_thr = [] { const char* e = getenv("ASR_GEMM_SMALL"); return e ? atoi(e) : 160; }();
    static const int s3s = [] { const char* e = getenv("ASR_GEMM_S3S"); return e ? atoi(e) : 1; }();
    {
        // split3 on 64x64 tiles (fp32-accurate, bf16 pipe): few 128x128 tiles, or edges the 128x128 kernel does not take
        // (V = 1000 logit columns, (T-1)*B rows)
        const int nwg64 = ((M + 63) / 64) * ((N + 63) / 64);
        if (np == 3 && s3s && splits == 1 && ((long long)nwg * batch < small_thr || M % 128 || N % 128 || K % BK) && (K % 4 == 0 || (transA && !transB)) && K >= 16 &&
            g.vecA && g.vecB && !(transA && transB) && nwg64 >= 8) {
            if (transA)       hipLaunchKernelGGL((gemm_split3s_kernel<true, false>), dim3(nwg64, 1, batch), dim3(256), 0, s, g);
            else if (transB)  hipLaunchKernelGGL((gemm_split3s_kernel<false, true>), dim3(nwg64, 1, batch), dim3(256), 0, s, g);
            else              hipLaunchKernelGGL((gemm_split3s_kernel<false, false>), dim3(nwg64, 1, batch), dim3(256), 0, s, g);
            ASR_CHECK_LAUNCH();
            return ASR_OK;
        }
    }
    if (splits == 1 && (long long)nwg * batch < small_thr) {
        nwg = ((M + 63) / 64) * ((N + 63) / 64);
        if (transA && transB)       hipLaunchKernelGGL((gemm_f32_kernel<true, true, 64>), dim3(nwg, 1, batch), dim3(256), 0, s, g);
        else if (transA)            hipLaunchKernelGGL((gemm_f32_kernel<true, false, 64>), dim3(nwg, 1, batch), dim3(256), 0, s, g);
        else if (transB)            hipLaunchKernelGGL((gemm_f32_kernel<false, true, 64>), dim3(nwg, 1, batch), dim3(256), 0, s, g);
        else                        hipLaunchKernelGGL((gemm_f32_kernel<false, false, 64>), dim3(nwg, 1, batch), dim3(256), 0, s, g);
        ASR_CHECK_LAUNCH();
        return ASR_OK;
    }
    // whole tiles, aligned operands: the variant without bounds checks (all large encoder products)
    if (M % 128 == 0 && N % 128 == 0 && K % BK == 0 && g.vecA && g.vecB && (!transA || !transB)) {
        // ASR_GEMM_TA_PAD (bytes, experiment knob): extra dynamic LDS per workgroup of the weight-gradient form, i.e. fewer of
        // them co-resident with the persistent BPTT workgroup of the CU they share
        static const int ta_pad = [] { const char* e = getenv("ASR_GEMM_TA_PAD"); return e ? atoi(e) : 0; }();
        static const int xs = [] { const char* e = getenv("ASR_GEMM_XCD_SPLIT"); return e ? atoi(e) : 1; }();
        if (transA && xs && splits >= 6 && nk >= 128 && (batch == 1 || (nwg * ((splits + 4) / 8 * 8)) % 8 == 0)) {   // K slices pinned to XCDs: a multiple of 8 slices
            g.splits = (splits + 4) / 8 * 8;
            g.xcd_split = 1;
            hipLaunchKernelGGL((gemm_f32_kernel<true, false, 128, true>), dim3(nwg * g.splits, 1, batch), dim3(256), ta_pad, s, g);
            ASR_CHECK_LAUNCH();
            return slab_end(BK);
        }
        if (transA)       hipLaunchKernelGGL((gemm_f32_kernel<true, false, 128, true>), dim3(nwg, splits, batch), dim3(256), ta_pad, s, g);
        else if (transB)  hipLaunchKernelGGL((gemm_f32_kernel<false, true, 128, true>), dim3(nwg, splits, batch), dim3(256), 0, s, g);
        else              hipLaunchKernelGGL((gemm_f32_kernel<false, false, 128, true>), dim3(nwg, splits, batch), dim3(256), 0, s, g);
        ASR_CHECK_LAUNCH();
        return slab_end(BK);
    }
    if (transA && transB)       hipLaunchKernelGGL((gemm_f32_kernel<true, true>), dim3(nwg, splits, batch), dim3(256), 0, s, g);
    else if (transA)            hipLaunchKernelGGL((gemm_f32_kernel<true, false>), dim3(nwg, splits, batch), dim3(256), 0, s, g);
    else if (transB)            hipLaunchKernelGGL((gemm_f32_kernel<false, true>), dim3(nwg, splits, batch), dim3(256), 0, s, g);
    else                        hipLaunchKernelGGL((gemm_f32_kernel<false, false>), dim3(nwg, splits, batch), dim3(256), 0, s, g);
    ASR_CHECK_LAUNCH();
    return slab_end(BK);
}
