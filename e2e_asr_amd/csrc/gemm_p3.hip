// GEMMs on operands that are ALREADY split into bf16 planes in HBM ("P3" operands): no conversion work in the k-loop.
//
// Round 3's split3 kernel (gemm.hip) reads fp32 operands and splits every value into three bf16 terms while its tile is staged
// into LDS.  Its own knock-outs named what bounds it: the split's VALU work (v_cvt_pk / v_dot2) takes the matrix unit away from
// both waves of a SIMD, and one barrier + one LDS round trip stand against only 24 MFMAs per k-tile (0.39 of the 2.5 PF / 6
// ceiling).  Here the data flow is changed instead of the loop: whoever PRODUCES an operand writes it as bf16 planes (the
// recurrent kernels for activations and gate gradients, one small pass per step for the weights), and the GEMM
//   * stages tiles with LDS-DMA (global_load_lds, 16 B per lane, no VGPR destination, no VALU), two stages, one barrier per k-step;
//   * holds a 64 x 128 wave tile (8 accumulator tiles of 32x32 = 128 registers): 48 MFMAs per wave and barrier at three planes;
//   * reads fragments with ds_read_b128 from a piece-swizzled, conflict-free image (the swizzle sits in the per-lane SOURCE
//     address of the LDS-DMA, whose LDS side is linear: cdna_hip_programming.md rule 21).
//
// P3 layout of a logical matrix X[R][C] (C a multiple of 8) with NP planes (3: x = h1 + h2 + h3 exactly, the fp32 default; 2:
// 16 significand bits; 1: plain bf16, BASELINE config 3): row-major in 16-byte PIECES,
//     piece(r, c8, p) = the 8 bf16 values plane p holds for X[r][8 c8 .. 8 c8 + 7],   at byte  r * ld8 * 16 NP + (c8 * NP + p) * 16,
// i.e. the planes of one 8-element chunk sit next to each other (48 contiguous bytes at NP = 3).  A producer whose lane owns one
// element writes 2 bytes per plane; a k-step of 16 needs 96 contiguous bytes of a row, a k-step over ROWS (the weight-gradient
// form) whole contiguous row segments.  With NP = 1 the layout is ordinary row-major bf16.
//
// Two forms:
//   KK  C[M][N] (+)= sum_k A[m][k] * B[n][k]      both operands contiguous along the contraction   (forward projections with the
//       weights stored transposed, data gradients dX = dG . K_x^T with the weights as they are)
//   RR  C[M][N] (+)= sum_r A[r][m] * B[r][n]      both operands contiguous ACROSS the contraction  (weight gradients X^T . dG:
//       the contraction runs over the rows = the B*T frames); fragments by ds_read_b64_tr_b16 (the transposing LDS read)
// The six products kept per k-step are those of gemm.hip (a_i b_j, i + j <= 4, smallest first); the arithmetic and its error
// are the same (tests/test_gpu_gemm_p3.py holds both forms to the exact-fp32 kernel's error against float64).
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace asr {

typedef __bf16 p3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 p3_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 p3_bf16x2 __attribute__((ext_vector_type(2)));

struct P3Args {
    const char* A; const char* B;       // P3 operands (bytes)
    float* C; const float* bias;
    int M, N, K;                        // KK: A [M][K], B [N][K];  RR: A [K][M], B [K][N]
    long long rbA, rbB;                 // row pitch of A / B in bytes
    int ldc;
    int accumulate;                     // C += result
    int splits;                         // K split over blockIdx.y (atomicAdd epilogue into a live / pre-zeroed C)
    const int* colmap;                  // RR: output column of product column n (NULL: identity) -- gate-major <- unit-major
    // RR, weight gradients dK = [X | Hprev]^T . dG of one (Bi)LSTM layer in ONE launch: output rows [0, mA_valid) come from the
    // columns of A (mA of them staged: a multiple of 128, zero columns past mA_valid), rows mA_valid ... from the columns of A2;
    // blockIdx.z = direction: A2 += z * zA2, B += z * zB (bytes), C += z * zC (elements)
    const char* A2; long long rbA2; int mA, mA_valid; long long zA2, zB, zC;
    // splits > 1: slice y of batch z stores its partial tile at slab[((z * splits + y) * M + m) * N + n] in PRODUCT coordinates (row
    // m of [A | A2]^T, column n before colmap) and slab_reduce (csrc/splitk.hip) adds the slices into C in order; NULL: float atomics
    float* slab;
};

__device__ __forceinline__ constexpr int p3_ctz(int v) { int n = 0; while (!(v & 1)) { v >>= 1; ++n; } return n; }

// LDS image of an operand tile of one stage: row r holds its PPR pieces (k-chunk major, plane minor, as in memory) at piece index
// r * PPR + (q ^ sw(r)): PPR = 2^a * odd puts rows 16 / 2^a apart on the same 16-byte slots of the 256-byte bank row, so the low a
// bits of the piece number are XORed with the row bits that tell such rows apart -- every 16-lane group of a ds_read_b128 then
// covers 16 distinct slots (worked through for PPR = 2, 4, 6, 8, 12 in DESIGN.md).
template <int PPR>
__device__ __forceinline__ int p3_sw(int r) {
    constexpr int a = p3_ctz(PPR) > 4 ? 4 : p3_ctz(PPR);
    return (r >> (4 - a)) & ((1 << a) - 1);
}

// RR image: rotation of contraction row r's pieces, chosen so that for a fixed plane the pieces of four consecutive chunks of four
// consecutive rows (what one 32-lane half of a ds_read_b64_tr_b16 touches) cover 16 distinct 16-byte slots of the bank row.
// Pieces of a row are chunk-major, plane-minor (u = chunk * NP + plane), and a row holds a multiple of 16 pieces.
template <int NP>
__device__ __forceinline__ int p3_rot(int r) {
    return NP == 3 ? 12 * (r & 3) : NP == 2 ? 8 * (r & 1) + ((r >> 1) & 1) : 4 * (r & 3);
}
__device__ __forceinline__ p3_bf16x4 p3_tr_read(const char* l) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(l));
    return __builtin_bit_cast(p3_bf16x4, v);
}

// One LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to LDS bytes [lds, lds + 1024) (M0 = the wave-uniform LDS
// address; the lane's 16 bytes land at + 16 lane).  In inline asm ON PURPOSE: hipcc knows that __builtin_amdgcn_global_load_lds
// writes LDS and puts `s_waitcnt vmcnt(0)` in front of every later LDS read it cannot prove disjoint (it did, in front of each
// ds_read_b64_tr_b16 of the RR form: the whole DMA queue drained eight times per k-step, 91 instead of 165 TF/s).  The ordering
// between a DMA and the reads of ITS buffer is this kernel's own: counted vmcnt + barrier, S stages apart.
__device__ __forceinline__ void p3_glds16(const char* g, char* l) {
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)l;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(lds) : "memory", "m0");
}

// ---------------------------------------------------------------------------------------------------------------------------
// KK form.  Workgroup = WM x WN waves, wave tile 64 x 128, block tile (64 WM) x (128 WN); KS MFMA k-steps (16 each) per stage.
// ---------------------------------------------------------------------------------------------------------------------------
// DBG (timing experiments only, wrong results): 1 = no LDS-DMA inside the loop, 2 = no MFMAs, 3 = no fragment reads
// RR form (RR = true): the contraction runs over the ROWS of both operands; a stage holds 16 KS rows x (BM / 8) NP pieces of A and
// x (BN / 8) NP pieces of B exactly as they lie in memory (whole contiguous row segments), and a fragment is two
// ds_read_b64_tr_b16: the transposing read hands lane i of a 16-lane group column i of a 4 (rows) x 16 (columns) block.
template <bool RR, int NP, int KS, int WM, int WN, int DBG = 0, int S = 4>   // S >= 3
__global__ __launch_bounds__(64 * WM * WN, 1) void gemm_p3_kernel(P3Args a) {
    constexpr int NW = WM * WN, BM = 64 * WM, BN = 128 * WN;
    constexpr int PPR = 2 * KS * NP;                      // KK: pieces per row and stage
    constexpr int PRA = BM / 8 * NP, PRB = BN / 8 * NP;   // RR: pieces per contraction row of the A / B tile
    constexpr int PA = RR ? 16 * KS * PRA : BM * PPR, PB = RR ? 16 * KS * PRB : BN * PPR;   // pieces per stage
    constexpr int JA = PA / 64 / NW, JB = PB / 64 / NW;   // LDS-DMA instructions per wave and stage
    static_assert(PA % (64 * NW) == 0 && PB % (64 * NW) == 0, "tile does not divide over the waves");
    constexpr int STAGE = (PA + PB) * 16;
    constexpr int JW = JA + JB;
    constexpr int NPROD = NP == 3 ? 6 : NP == 2 ? 3 : 1;
    __shared__ __attribute__((aligned(1024))) char smem[S * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ntn = a.N / BN, ntm = a.M / BM, nwg = ntn * ntm;
    int bid = blockIdx.x;
    {   // XCD-aware order: blocks b and b + 8 share an L2 -- each XCD label gets a contiguous run of tiles (gemm.hip)
        const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
    int crow0 = m0, mvalid = BM;                          // first output row of the tile, rows of it that exist
    if (RR) {
        a.B += blockIdx.z * a.zB; a.C += blockIdx.z * a.zC;
        if (m0 >= a.mA) { a.A = a.A2 + blockIdx.z * a.zA2 + (long long)((m0 - a.mA) / 8) * (NP * 16); a.rbA = a.rbA2; crow0 = a.mA_valid + (m0 - a.mA); }
        else { a.A += (long long)(m0 / 8) * (NP * 16); mvalid = a.mA_valid - m0; }
    }
    const int nk_all = a.K / (16 * KS);                  // (RR: K = the number of operand rows)
    const int per = (nk_all + a.splits - 1) / a.splits;
    const int kt0 = blockIdx.y * per, kt1 = min(nk_all, kt0 + per);
    if (kt0 >= kt1) return;

    // ---- staging: instruction j of this wave covers LDS pieces [(wave * J + j) * 64, +64) of the A (then B) image; the lane's
    // LDS piece -> the tile piece it must hold (the swizzles below) -> its source address (loop-invariant 32-bit offsets; the
    // k-step advances the base).  KK: piece (row, q) at row * PPR + (q ^ sw(row)).  RR: piece (r, u) of contraction row r at
    // r * PR + (u + rot(r)) mod PR -- the four rows a transposing read touches then sit on different 16-byte slots (rot: p3_rot).
    unsigned offA[JA], offB[JB];
#pragma unroll
    for (int j = 0; j < JA; ++j) {
        const int P = (wave * JA + j) * 64 + lane;
        if (RR) { const int r = P / PRA, u = (P % PRA + PRA - p3_rot<NP>(r)) % PRA; offA[j] = (unsigned)((long long)r * a.rbA) + u * 16; }
        else { const int r = P / PPR, q = (P % PPR) ^ p3_sw<PPR>(r); offA[j] = (unsigned)((long long)r * a.rbA) + q * 16; }
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const int P = (wave * JB + j) * 64 + lane;
        if (RR) { const int r = P / PRB, u = (P % PRB + PRB - p3_rot<NP>(r)) % PRB; offB[j] = (unsigned)((long long)r * a.rbB) + u * 16; }
        else { const int r = P / PPR, q = (P % PPR) ^ p3_sw<PPR>(r); offB[j] = (unsigned)((long long)r * a.rbB) + q * 16; }
    }
    const long long kadvA = RR ? 16ll * KS * a.rbA : PPR * 16, kadvB = RR ? 16ll * KS * a.rbB : PPR * 16;
    const char* gA = RR ? a.A + kt0 * kadvA : a.A + (long long)m0 * a.rbA + kt0 * kadvA;
    const char* gB = RR ? a.B + (long long)(n0 / 8) * (NP * 16) + kt0 * kadvB : a.B + (long long)n0 * a.rbB + kt0 * kadvB;
    char* const ldsA = smem + (wave * JA) * 1024;                 // + stage * STAGE + j * 1024 (+ lane * 16 by the hardware)
    char* const ldsB = smem + PA * 16 + (wave * JB) * 1024;
    auto issue = [&](int st) {
#pragma unroll
        for (int j = 0; j < JA; ++j) p3_glds16(gA + offA[j], ldsA + st * STAGE + j * 1024);
#pragma unroll
        for (int j = 0; j < JB; ++j) p3_glds16(gB + offB[j], ldsB + st * STAGE + j * 1024);
        gA += kadvA; gB += kadvB;
    };

    // ---- fragments: lane l of a 32x32x16 operand holds row (l & 31), k = 8 (l >> 5) .. + 7 = ONE piece
    const int fr = lane & 31, fh = lane >> 5;
    const int rowA = wm * 64 + fr, rowB = wn * 128 + fr;
    const int swA = p3_sw<PPR>(rowA), swB = p3_sw<PPR>(rowB);          // (unchanged by + 32 i: see p3_sw)
    // RR: lane = 16 g + 4 q + p supplies the address of contraction row 8 (g >> 1) + 4 rd + q, columns 16 (g & 1) + 4 p .. + 3 of the
    // 32-column tile (rd = 0, 1: the two reads of a fragment) and receives column (lane & 31), rows 8 (lane >> 5) + 4 rd .. + 3
    const int tq = (lane >> 2) & 3, tp = lane & 3, tg = (lane >> 4) & 1;
    const int trr = 8 * fh + tq;                                       // + 4 rd (+ 16 ks)
    const int tcA = (wm * 64 + 16 * tg) / 8 + (tp >> 1), tcB = (wn * 128 + 16 * tg) / 8 + (tp >> 1);    // chunk (+ 4 i / + 4 j)
    const int thalf = (tp & 1) * 8;
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- the k-loop.  Three things run side by side in every k-step t, spread over one another by the scheduler hints:
    //   * the 8 x NPROD MFMAs of tile t, out of one register set of fragments (products smallest first, each product over all
    //     eight accumulator tiles before the next: dependent MFMAs are eight instructions apart);
    //   * the fragment reads of tile t + 1 into the OTHER register set (one wave per SIMD: nobody else would cover the ~400
    //     cycles between a k-step's barrier and its first fragment -- with the reads in front of the MFMAs the matrix pipe
    //     was busy 65 % of the time, rocprofv3 SQ_VALU_MFMA_BUSY_CYCLES);
    //   * the LDS-DMA of tile t + S - 1 (an LDS-DMA costs its wave ~60-100 cycles of issue; as a block in front of the MFMAs
    //     that was 9 x that per k-step with the matrix pipe idle: one DMA per five MFMAs hides in the issue slots MFMAs leave).
    // An LDS-DMA takes ~1.1 us from issue to landing under load (MI355X_MICROARCH.md, ldsdma-fill) -- longer than the 0.75 us a
    // wave multiplies on one k-step -- so a tile is requested S - 1 k-steps before its MFMAs, S - 2 before its fragment reads.
    // Step t: wait until this wave's DMA of tile t + 1 has landed (all but the younger tiles' instructions), barrier (tile t + 1
    // complete and published; everybody has read tile t, so the DMA of tile t + S - 1 -- into the buffer tile t - 1 had -- races
    // with nothing).  Raw s_barrier + counted vmcnt: __syncthreads() would drain the DMA queue (cdna_hip_programming.md).
    struct Frags { p3_bf16x8 a[KS][2][NP], b[KS][4][NP]; };
    constexpr int ND = 6 * NP * KS;                 // fragments (one ds_read_b128, or two transposing reads) per k-step
    constexpr int NM = 8 * NPROD * KS;              // MFMAs per k-step
    // fragment number u of a k-step: (ks, A tile i | B tile j, plane)
    auto read_one = [&](Frags& f, int st, int u) {
        const char* sA = smem + st * STAGE;
        const char* sB = sA + PA * 16;
        const int ks = u / (6 * NP), w = u % (6 * NP);
        const bool isA = w < 2 * NP;
        const int t = isA ? w / NP : (w - 2 * NP) / NP, pl = w % NP;
        p3_bf16x8 v;
        if (DBG == 3) asm volatile("" : "=v"(v));
        else if constexpr (RR) {
            p3_bf16x4 lo, hi;
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const int r = 16 * ks + trr + 4 * rd;
                const int PR = isA ? PRA : PRB;
                const int uu = (((isA ? tcA : tcB) + 4 * t) * NP + pl + p3_rot<NP>(r)) % PR;
                const p3_bf16x4 x = p3_tr_read((isA ? sA : sB) + (r * PR + uu) * 16 + thalf);
                if (rd == 0) lo = x; else hi = x;
            }
            v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
            const int q = (2 * ks + fh) * NP + pl;
            v = isA ? *reinterpret_cast<const p3_bf16x8*>(sA + ((rowA + 32 * t) * PPR + (q ^ swA)) * 16)
                    : *reinterpret_cast<const p3_bf16x8*>(sB + ((rowB + 32 * t) * PPR + (q ^ swB)) * 16);
        }
        if (isA) f.a[ks][t][pl] = v; else f.b[ks][t][pl] = v;
    };
    // MFMA number m of a k-step: products smallest first, each product over all eight accumulator tiles before the next
    auto mfma_one = [&](const Frags& f, int m) {
        constexpr int PRA_[6] = {2, 0, 1, 1, 0, 0}, PRB_[6] = {0, 2, 1, 0, 1, 0};     // a2b0 a0b2 a1b1 a1b0 a0b1 a0b0
        const int ks = m / (8 * NPROD), mm = m % (8 * NPROD), pr = (6 - NPROD) + mm / 8, i = (mm % 8) / 4, j = mm % 4;
        if (DBG == 2) { asm volatile("" :: "v"(f.a[ks][i][PRA_[pr]]), "v"(f.b[ks][j][PRB_[pr]])); return; }
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][i][PRA_[pr]], f.b[ks][j][PRB_[pr]], acc[i][j], 0, 0, 0);
    };
    auto dma_one = [&](int st, int g) {
        if (g < JA) p3_glds16(gA + offA[g < JA ? g : 0], ldsA + st * STAGE + g * 1024);
        else p3_glds16(gB + offB[g >= JA ? g - JA : 0], ldsB + st * STAGE + (g - JA) * 1024);
    };
    // one k-step in JW groups of (one DMA instruction, its share of the next tile's fragment reads, its share of the MFMAs),
    // the order pinned group by group (the scheduler otherwise clusters the reads behind the MFMAs and the DMAs in front)
    auto kstep = [&](const Frags& cur, Frags& nxt, int st_dma, int st_read, bool dma, bool rd) {
        constexpr int G = JW, RPG = (ND + G - 1) / G, MPG = NM / G;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (dma && DBG != 1) dma_one(st_dma, g);
#pragma unroll
            for (int d = 0; d < RPG; ++d)
                if (g * RPG + d < ND && rd) read_one(nxt, st_read, g * RPG + d);
#pragma unroll
            for (int m = 0; m < MPG; ++m) mfma_one(cur, g * MPG + m);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = G * MPG; m < NM; ++m) mfma_one(cur, m);
        if (dma) { gA += kadvA; gB += kadvB; }
    };
    const int nkt = kt1 - kt0;
    // tiles 0 .. S-2 requested; tile 0's fragments read
#pragma unroll
    for (int p = 0; p < S - 1; ++p)
        if (p < nkt) issue(p);
    {
        const int ahead = min(S - 2, nkt - 1);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * JW) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(JW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    Frags f0, f1;
#pragma unroll
    for (int u = 0; u < ND; ++u) read_one(f0, 0, u);
    int it = 0;
    // steady state, two k-steps per trip (the register sets alternate): at the top of step t tiles .. t + S - 2 are requested
    auto steady = [&](Frags& cur, Frags& nxt, int t) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 3) * JW) : "memory");      // tile t + 1 landed; t + 2 .. t + S - 2 may be in flight
        __builtin_amdgcn_s_barrier();
        kstep(cur, nxt, (t + S - 1) % S, (t + 1) % S, true, true);
    };
    for (; it + S < nkt; it += 2) {
        steady(f0, f1, it);
        steady(f1, f0, it + 1);
    }
    // drain: one step at a time, the sets swapped by copying (a handful of steps per tile)
    for (; it < nkt; ++it) {
        const bool more = it + 1 < nkt, req = it + S - 1 < nkt;
        if (more) {
            const int infl = min(nkt - 2 - it, S - 3);          // tiles younger than it + 1 still in flight
            if (infl >= 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(JW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        kstep(f0, f1, (it + S - 1) % S, (it + 1) % S, req, more);
        f0 = f1;
    }

    // ---- epilogue.  C/D map of a 32x32 tile: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5).
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int n = n0 + wn * 128 + j * 32 + fr;
            if (RR && a.colmap) n = a.colmap[n];
            const float bv = (a.bias && blockIdx.y == 0) ? a.bias[n] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ml = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                if (RR && ml >= mvalid) continue;
                const int m = crow0 + ml;
                float* cp = a.C + (size_t)m * a.ldc + n;
                float v = acc[i][j][e] + bv;
                if (a.splits > 1) {
                    if (a.slab) a.slab[((size_t)(blockIdx.z * a.splits + blockIdx.y) * a.M + (m0 + ml)) * a.N + (n0 + wn * 128 + j * 32 + fr)] = v;
                    else atomicAdd(cp, v);
                }
                else { if (a.accumulate) v += *cp; *cp = v; }
            }
        }
    (void)NPROD;
}

// ---------------------------------------------------------------------------------------------------------------------------
// fp32 -> P3.  transpose = 0: dst is the P3 image of src[R][C];  1: of src^T (dst logical [C][R]) -- the weights, once per step.
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t p3_cvt_pk(float lo, float hi) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f t = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(t, p3_bf16x2));
}
template <int NP>
__device__ __forceinline__ void p3_split8(const float* x, uint4* out) {     // out[p] = piece of plane p
    uint32_t pk[NP][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x0 = x[2 * j], x1 = x[2 * j + 1];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            pk[p][j] = p3_cvt_pk(x0, x1);
            if (p + 1 < NP) { x0 -= __uint_as_float(pk[p][j] << 16); x1 -= __uint_as_float(pk[p][j] & 0xffff0000u); }
        }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) out[p] = make_uint4(pk[p][0], pk[p][1], pk[p][2], pk[p][3]);
}
// dcols >= the logical column count: destination columns past it are zero (a tile-aligned image of a narrower matrix);
// umh > 0 (not transposed): destination column c = d * 4 umh + 4 u + g holds source column d * 4 umh + g * umh + u -- the
// gate-major columns of a TF LSTM kernel in the unit-major order the BPTT writes dG in.
template <int NP, bool TR>
__global__ __launch_bounds__(256) void p3_split_kernel(const float* __restrict__ src, int R, int C, int ld, char* __restrict__ dst,
                                                       long long rb, int dcols, int umh, bool vec) {
    // logical destination [DR][dcols]: one thread per 8-element chunk
    const int DR = TR ? C : R, DC = TR ? R : C, DC8 = dcols / 8;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)DR * DC8) return;
    float x[8];
    int dr, c8;
    if (TR) {       // consecutive threads: consecutive destination ROWS (= source columns): coalesced source reads
        dr = (int)(idx % DR); c8 = (int)(idx / DR);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (c8 * 8 + j < DC) ? src[(size_t)(c8 * 8 + j) * ld + dr] : 0.f;
    } else if (umh > 0) {
        dr = (int)(idx / DC8); c8 = (int)(idx % DC8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c8 * 8 + j, d = c / (4 * umh), w = c % (4 * umh);
            x[j] = c < DC ? src[(size_t)dr * ld + d * 4 * umh + (w & 3) * umh + (w >> 2)] : 0.f;
        }
    } else {
        dr = (int)(idx / DC8); c8 = (int)(idx % DC8);
        // vec: rows start 16-byte aligned (base aligned, ld % 4 == 0); otherwise (feat_length 39 / 43 / 83 ...) scalar loads
        if (vec && c8 * 8 + 8 <= DC) {
            const float4* p = reinterpret_cast<const float4*>(src + (size_t)dr * ld + c8 * 8);
            const float4 u = p[0], v = p[1];
            x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w; x[4] = v.x; x[5] = v.y; x[6] = v.z; x[7] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = (c8 * 8 + j < DC) ? src[(size_t)dr * ld + c8 * 8 + j] : 0.f;
        }
    }
    uint4 out[NP];
    p3_split8<NP>(x, out);
    uint4* d = reinterpret_cast<uint4*>(dst + (long long)dr * rb + (long long)c8 * (NP * 16));
#pragma unroll
    for (int p = 0; p < NP; ++p) d[p] = out[p];
}

}  // namespace asr

extern "C" size_t asr_p3_bytes(int rows, int cols, int np) {
    return (size_t)rows * (size_t)((cols + 7) / 8) * 16 * (size_t)np;
}

extern "C" int asr_p3_split_ex(void* stream, const float* src, int rows, int cols, int ld, void* dst, int np, int transpose,
                               int dst_cols, int unit_major_h) {
    using namespace asr;
    if (!src || !dst || rows <= 0 || cols <= 0 || ld < cols || np < 1 || np > 3) return ASR_EINVAL;
    const int DR = transpose ? cols : rows, DC = transpose ? rows : cols;
    if (dst_cols <= 0) dst_cols = (DC + 7) / 8 * 8;
    if (dst_cols % 8 || dst_cols < DC) return ASR_EINVAL;
    if (unit_major_h > 0 && (transpose || DC % (4 * unit_major_h))) return ASR_EINVAL;
    const bool vec = !(reinterpret_cast<uintptr_t>(src) & 15) && ld % 4 == 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long rb = (long long)(dst_cols / 8) * 16 * np;
    const long long n = (long long)DR * (dst_cols / 8);
    const unsigned grid = (unsigned)((n + 255) / 256);
    char* d = static_cast<char*>(dst);
#define P3_SPLIT(NP_) do { if (transpose) hipLaunchKernelGGL((p3_split_kernel<NP_, true>), dim3(grid), dim3(256), 0, s, src, rows, cols, ld, d, rb, dst_cols, 0, vec); \
                           else hipLaunchKernelGGL((p3_split_kernel<NP_, false>), dim3(grid), dim3(256), 0, s, src, rows, cols, ld, d, rb, dst_cols, unit_major_h, vec); } while (0)
    if (np == 3) P3_SPLIT(3); else if (np == 2) P3_SPLIT(2); else P3_SPLIT(1);
#undef P3_SPLIT
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// Up to 8 splits in ONE launch (blockIdx.y = job): the per-step weight images of all encoder layers (K_x^T for the forward
// projections, unit-major K_x for the data gradients) -- six launches of ~9 us each otherwise, back to back on the critical path.
struct asr_p3_split_job { const float* src; int rows, cols, ld; void* dst; int np, transpose, dst_cols, unit_major_h; };
namespace asr {
struct P3SplitJobs { asr_p3_split_job j[8]; };
template <int NP>
__device__ __forceinline__ void p3_split_job_body(const asr_p3_split_job& q, long long idx) {
    const bool TR = q.transpose != 0;
    const int R = q.rows, Cc = q.cols, ld = q.ld, umh = q.unit_major_h;
    const int DR = TR ? Cc : R, DC = TR ? R : Cc, DC8 = q.dst_cols / 8;
    if (idx >= (long long)DR * DC8) return;
    const long long rb = (long long)DC8 * 16 * NP;
    float x[8];
    int dr, c8;
    if (TR) {
        dr = (int)(idx % DR); c8 = (int)(idx / DR);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (c8 * 8 + j < DC) ? q.src[(size_t)(c8 * 8 + j) * ld + dr] : 0.f;
    } else {
        dr = (int)(idx / DC8); c8 = (int)(idx % DC8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c8 * 8 + j;
            int sc = c;
            if (umh > 0) { const int d = c / (4 * umh), w = c % (4 * umh); sc = d * 4 * umh + (w & 3) * umh + (w >> 2); }
            x[j] = c < DC ? q.src[(size_t)dr * ld + sc] : 0.f;
        }
    }
    uint4 out[NP];
    p3_split8<NP>(x, out);
    uint4* d = reinterpret_cast<uint4*>(static_cast<char*>(q.dst) + (long long)dr * rb + (long long)c8 * (NP * 16));
#pragma unroll
    for (int p = 0; p < NP; ++p) d[p] = out[p];
}
__global__ __launch_bounds__(256) void p3_split_multi_kernel(P3SplitJobs jobs) {
    const asr_p3_split_job& q = jobs.j[blockIdx.y];
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q.np == 3) p3_split_job_body<3>(q, idx); else if (q.np == 2) p3_split_job_body<2>(q, idx); else p3_split_job_body<1>(q, idx);
}
}  // namespace asr
extern "C" int asr_p3_split_multi(void* stream, int njobs, const asr_p3_split_job* jobs) {
    using namespace asr;
    if (njobs <= 0 || njobs > 8 || !jobs) return ASR_EINVAL;
    P3SplitJobs J;
    long long maxn = 0;
    for (int i = 0; i < njobs; ++i) {
        asr_p3_split_job q = jobs[i];
        if (!q.src || !q.dst || q.rows <= 0 || q.cols <= 0 || q.ld < q.cols || q.np < 1 || q.np > 3) return ASR_EINVAL;
        const int DR = q.transpose ? q.cols : q.rows, DC = q.transpose ? q.rows : q.cols;
        if (q.dst_cols <= 0) q.dst_cols = (DC + 7) / 8 * 8;
        if (q.dst_cols % 8 || q.dst_cols < DC) return ASR_EINVAL;
        if (q.unit_major_h > 0 && (q.transpose || DC % (4 * q.unit_major_h))) return ASR_EINVAL;
        J.j[i] = q;
        maxn = std::max(maxn, (long long)DR * (q.dst_cols / 8));
    }
    for (int i = njobs; i < 8; ++i) J.j[i] = J.j[0];
    hipLaunchKernelGGL(p3_split_multi_kernel, dim3((unsigned)((maxn + 255) / 256), njobs), dim3(256), 0, static_cast<hipStream_t>(stream), J);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

extern "C" int asr_p3_split_f32(void* stream, const float* src, int rows, int cols, int ld, void* dst, int np, int transpose) {
    if ((transpose ? rows : cols) % 8) return ASR_EINVAL;
    return asr_p3_split_ex(stream, src, rows, cols, ld, dst, np, transpose, 0, 0);
}

// C[M,N] (+)= A . B^T (+ bias) on P3 operands A [M][K], B [N][K] with `np` planes each; lda8 / ldb8: row pitch in 8-element
// chunks (>= K / 8).  M % 128 == 0, N % 256 == 0, K % 16 == 0.  splits > 1: K split over that many workgroups per tile with
// an atomicAdd epilogue (C is zeroed first unless `accumulate`); splits = 0 lets the library choose (1 for this form).
extern "C" int asr_gemm_p3_kk(void* stream, int M, int N, int K, const void* A, int lda8, const void* B, int ldb8, int np,
                              float* C, int ldc, const float* bias, int accumulate, int splits) {
    using namespace asr;
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || np < 1 || np > 3) return ASR_EINVAL;
    if (M % 128 || N % 256 || K % 16 || lda8 < K / 8 || ldb8 < K / 8 || ldc < N) return ASR_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return ASR_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    P3Args g;
    g.A = static_cast<const char*>(A); g.B = static_cast<const char*>(B); g.C = C; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.rbA = (long long)lda8 * 16 * np; g.rbB = (long long)ldb8 * 16 * np;
    g.ldc = ldc; g.accumulate = accumulate; g.splits = splits < 1 ? 1 : splits; g.colmap = nullptr;
    g.A2 = nullptr; g.rbA2 = 0; g.mA = M; g.mA_valid = M; g.zA2 = g.zB = g.zC = 0;
    if ((long long)M * g.rbA >= (1ll << 32) || (long long)N * g.rbB >= (1ll << 32)) return ASR_EUNSUPPORTED;   // 32-bit tile offsets
    g.slab = (g.splits > 1 && wgrad_slabs()) ? slab_arena(s, (size_t)g.splits * M * N * sizeof(float)) : nullptr;
    auto slab_end = [&](int ks) -> int {
        if (!g.slab) return ASR_OK;
        const int nk_all = K / (16 * ks), per = (nk_all + g.splits - 1) / g.splits;
        SlabMap q;
        q.M = M; q.N = N; q.Nvalid = N; q.nsl = (nk_all + per - 1) / per; q.nsl_alloc = g.splits; q.batch = 1; q.mA = M; q.mA_valid = M;
        q.colmap = nullptr; q.ldc = ldc; q.zC = 0; q.accumulate = accumulate;
        return slab_reduce(s, C, g.slab, q);
    };
    if (g.splits > 1 && !accumulate && !g.slab &&
        hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, s) != hipSuccess) return ASR_ELAUNCH;
    // One plane (bf16): a 128 x 256 tile asks the LDS-DMA for 48 bytes per CU and cycle at full MFMA rate -- more than it delivers;
    // the 256 x 256 tile of eight waves (two per SIMD, 249 registers) needs 32: 4096^3 870 -> 1 124 TF/s, the layer-2 projection
    // 543 -> 748 (ASR_P3_TILE256=0: the four-wave tile everywhere)
    static const int big = [] { const char* e = getenv("ASR_P3_TILE256"); return e ? atoi(e) : 1; }();
    if (big && np == 1 && M % 256 == 0 && K % 32 == 0 && (M / 256) * (N / 256) >= 128) {
        hipLaunchKernelGGL((gemm_p3_kernel<false, 1, 2, 4, 2>), dim3((M / 256) * (N / 256), g.splits, 1), dim3(512), 0, s, g);
        ASR_CHECK_LAUNCH();
        return slab_end(2);
    }
    const dim3 grid((M / 128) * (N / 256), g.splits, 1);
    static const int dbg = [] { const char* e = getenv("ASR_P3_DBG"); return e ? atoi(e) : 0; }();
    if (np == 3 && dbg == 1) hipLaunchKernelGGL((gemm_p3_kernel<false, 3, 1, 2, 2, 1>), grid, dim3(256), 0, s, g);
    else if (np == 3 && dbg == 2) hipLaunchKernelGGL((gemm_p3_kernel<false, 3, 1, 2, 2, 2>), grid, dim3(256), 0, s, g);
    else if (np == 3 && dbg == 3) hipLaunchKernelGGL((gemm_p3_kernel<false, 3, 1, 2, 2, 3>), grid, dim3(256), 0, s, g);
    else if (np == 3) hipLaunchKernelGGL((gemm_p3_kernel<false, 3, 1, 2, 2>), grid, dim3(256), 0, s, g);
    else if (np == 2) hipLaunchKernelGGL((gemm_p3_kernel<false, 2, 1, 2, 2>), grid, dim3(256), 0, s, g);
    else if (K % 32 == 0) hipLaunchKernelGGL((gemm_p3_kernel<false, 1, 2, 2, 2>), grid, dim3(256), 0, s, g);
    else              hipLaunchKernelGGL((gemm_p3_kernel<false, 1, 1, 2, 2>), grid, dim3(256), 0, s, g);
    ASR_CHECK_LAUNCH();
    return slab_end((np == 1 && K % 32 == 0) ? 2 : 1);
}

// RR form: C[M,N] (+)= A^T . B on P3 operands A [K][M], B [K][N] (K = rows of both: the B*T frames of a weight gradient X^T . dG).
// M % 128 == 0, N % 256 == 0, K % 16 == 0.  splits: K slices per output tile (0: the library fills the chip); with more than one
// slice the tiles meet in C through float atomics (C is zeroed first unless `accumulate`).  colmap (device, N ints or NULL):
// product column n is stored to column colmap[n] of C (a producer that writes its columns unit-major for a gate-major C).
extern "C" int asr_gemm_p3_rr(void* stream, int M, int N, int K, const void* A, int lda8, const void* B, int ldb8, int np,
                              float* C, int ldc, int accumulate, int splits, const int* colmap) {
    using namespace asr;
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || np < 1 || np > 3) return ASR_EINVAL;
    if (M % 128 || N % 256 || K % 16 || lda8 < M / 8 || ldb8 < N / 8 || ldc < N) return ASR_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return ASR_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    P3Args g;
    g.A = static_cast<const char*>(A); g.B = static_cast<const char*>(B); g.C = C; g.bias = nullptr;
    g.M = M; g.N = N; g.K = K; g.rbA = (long long)lda8 * 16 * np; g.rbB = (long long)ldb8 * 16 * np;
    g.ldc = ldc; g.accumulate = accumulate; g.colmap = colmap;
    g.A2 = nullptr; g.rbA2 = 0; g.mA = M; g.mA_valid = M; g.zA2 = g.zB = g.zC = 0;
    // one plane runs two MFMA k-steps per stage (KS = 2: 32 contraction rows) when K allows it; K % 32 == 16 takes the KS = 1
    // instantiation -- the KS = 2 kernel would drop the last 16 rows (nk_all = K / 32)
    const int ks = (np == 1 && K % 32 == 0) ? 2 : 1;
    const int tiles = (M / 128) * (N / 256), nk = K / (16 * ks);
    if (splits < 1) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        splits = std::max(1, std::min(cus / tiles, nk / 8));
    }
    g.splits = splits;
    const int per = (nk + splits - 1) / splits;
    if (16ll * ks * per * std::max(g.rbA, g.rbB) >= (1ll << 31)) return ASR_EUNSUPPORTED;       // 32-bit offsets inside a stage only: always true
    g.slab = (splits > 1 && wgrad_slabs()) ? slab_arena(s, (size_t)splits * M * N * sizeof(float)) : nullptr;
    if (splits > 1 && !accumulate && !g.slab &&
        hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, s) != hipSuccess) return ASR_ELAUNCH;
    const dim3 grid(tiles, splits, 1);
    if (np == 3)      hipLaunchKernelGGL((gemm_p3_kernel<true, 3, 1, 2, 2>), grid, dim3(256), 0, s, g);
    else if (np == 2) hipLaunchKernelGGL((gemm_p3_kernel<true, 2, 1, 2, 2>), grid, dim3(256), 0, s, g);
    else if (ks == 2) hipLaunchKernelGGL((gemm_p3_kernel<true, 1, 2, 2, 2>), grid, dim3(256), 0, s, g);
    else              hipLaunchKernelGGL((gemm_p3_kernel<true, 1, 1, 2, 2>), grid, dim3(256), 0, s, g);
    ASR_CHECK_LAUNCH();
    if (g.slab) {
        SlabMap q;
        q.M = M; q.N = N; q.Nvalid = N; q.nsl = (nk + per - 1) / per; q.nsl_alloc = splits; q.batch = 1; q.mA = M; q.mA_valid = M;
        q.colmap = colmap; q.ldc = ldc; q.zC = 0; q.accumulate = accumulate;
        return slab_reduce(s, C, g.slab, q);
    }
    return ASR_OK;
}

// Weight gradients of one (Bi)LSTM layer, dK_d += [X | Hprev_d]^T . dG_d for every direction d in ONE launch (called by
// asr_lstm_layer_bwd; declared in p3.h).  x_p3: P3 [rows][in_pad] (columns past in_valid zero), hprev_p3: P3 [rows][ndir*H],
// dg_p3: P3 [rows][ndir*4H] with unit-major columns; dk: the TF kernel gradient [in_valid + H][4H] of direction 0, direction 1 at
// + dk_stride elements; colmap: unit-major -> gate-major column inside one direction.  Accumulates (float atomics).
namespace asr {
int p3_lstm_wgrad(hipStream_t s, int rows, int in_pad, int in_valid, int H, int ndir, const void* x_p3, int x_ld8,
                  const void* hprev_p3, const void* dg_p3, int np, float* dk, long long dk_stride, const int* colmap) {
    const int H4 = 4 * H;
    if (in_pad % 128 || H % 128 || H4 % 256 || rows % 16 || in_valid > in_pad || np < 1 || np > 3) return ASR_EUNSUPPORTED;
    P3Args g;
    g.A = static_cast<const char*>(x_p3); g.rbA = (long long)x_ld8 * 16 * np;
    g.A2 = static_cast<const char*>(hprev_p3); g.rbA2 = (long long)(ndir * H / 8) * 16 * np; g.zA2 = (long long)(H / 8) * 16 * np;
    g.B = static_cast<const char*>(dg_p3); g.rbB = (long long)(ndir * H4 / 8) * 16 * np; g.zB = (long long)(H4 / 8) * 16 * np;
    g.C = dk; g.zC = dk_stride; g.bias = nullptr; g.ldc = H4; g.accumulate = 1; g.colmap = colmap;
    g.mA = in_pad; g.mA_valid = in_valid;
    g.M = in_pad + H; g.N = H4; g.K = rows;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    // (256 x 256 tiles double the bytes every K slice adds into dK with float atomics -- 61 instead of 30 MB per launch at ~1.3 TB/s:
    // measured in the bf16 train step, the larger tile LOSES here what it wins in the k-loop; opt-in ASR_P3_WGRAD256=1)
    static const int big = [] { const char* e = getenv("ASR_P3_WGRAD256"); return e ? atoi(e) : 0; }();
    auto with_slabs = [&](int nk_all) {          // g.splits is final: partial tiles through this stream's slab arena (csrc/splitk.hip)
        g.slab = (g.splits > 1 && wgrad_slabs()) ? slab_arena(s, (size_t)ndir * g.splits * g.M * g.N * sizeof(float)) : nullptr;
        (void)nk_all;
    };
    auto slab_end = [&](int nk_all) -> int {
        if (hipGetLastError() != hipSuccess) return ASR_ELAUNCH;
        if (!g.slab) return ASR_OK;
        const int per = (nk_all + g.splits - 1) / g.splits;
        SlabMap q;
        q.M = g.M; q.N = g.N; q.Nvalid = g.N; q.nsl = (nk_all + per - 1) / per; q.nsl_alloc = g.splits; q.batch = ndir;
        q.mA = g.mA; q.mA_valid = g.mA_valid; q.colmap = colmap; q.ldc = g.ldc; q.zC = dk_stride; q.accumulate = 1;
        return slab_reduce(s, dk, g.slab, q);
    };
    if (big && np == 1 && in_pad % 256 == 0 && H % 256 == 0 && rows % 32 == 0) {       // one plane: 256 x 256 tiles (see asr_gemm_p3_kk)
        const int tiles2 = (g.M / 256) * (g.N / 256) * ndir, nk2 = rows / 32;
        g.splits = std::max(1, std::min(cus / tiles2, nk2 / 8));
        with_slabs(nk2);
        hipLaunchKernelGGL((gemm_p3_kernel<true, 1, 2, 4, 2>), dim3(tiles2 / ndir, g.splits, ndir), dim3(512), 0, s, g);
        return slab_end(nk2);
    }
    const int ks = (np == 1 && rows % 32 == 0) ? 2 : 1;                    // (rows % 32 == 16: the KS = 1 kernel, see asr_gemm_p3_rr)
    const int tiles = (g.M / 128) * (g.N / 256) * ndir, nk = rows / (16 * ks);
    g.splits = std::max(1, std::min(cus / tiles, std::max(1, nk / 8)));    // one workgroup per CU and no second round
    const dim3 grid(tiles / ndir, g.splits, ndir);
    with_slabs(nk);
    if (np == 3)      hipLaunchKernelGGL((gemm_p3_kernel<true, 3, 1, 2, 2>), grid, dim3(256), 0, s, g);
    else if (np == 2) hipLaunchKernelGGL((gemm_p3_kernel<true, 2, 1, 2, 2>), grid, dim3(256), 0, s, g);
    else if (ks == 2) hipLaunchKernelGGL((gemm_p3_kernel<true, 1, 2, 2, 2>), grid, dim3(256), 0, s, g);
    else              hipLaunchKernelGGL((gemm_p3_kernel<true, 1, 1, 2, 2>), grid, dim3(256), 0, s, g);
    return slab_end(nk);
}
}  // namespace asr
