// Persistent (Bi)LSTM layer, forward -- the 1,500-step serial chain of the encoder.
//
// Reference semantics (encoder.py:55-91): tf.nn.bidirectional_dynamic_rnn /
// tf.nn.dynamic_rnn over BasicLSTMCell (arithmetic restated at basic_lstm.py:14-23:
// [x,h].K + b -> i,j,f,o; c' = c*sigmoid(f+1) + sigmoid(i)*tanh(j); h' = sigmoid(o)*tanh(c')),
// sequence_length masking (t >= len[b]: output 0, state copied through), backward
// direction = reverse_sequence o rnn o reverse_sequence (the cell walks t = len[b]-1 .. 0).
//
// MI355X design.  The input half of the contraction, X.K_x + b for ALL timesteps, is
// one MFMA GEMM per direction (gemm.hip) written to `gates` [B,T,ND,4H].  What is left
// per step is h_{t-1}.K_h (H x 4H) + the cell -- a dependent chain, so what matters is
// step LATENCY, not FLOPs.  Decomposition:
//   * utterances are independent => the batch is cut into groups of R rows; groups
//     never synchronise with each other (no grid barrier anywhere);
//   * inside a group, K_h is cut column-wise over G = H/HS workgroups, each owning HS
//     hidden units with all 4 of their gates.  Its K_h slice (H x 4HS fp32 = 128 KB at
//     H=256) lives in REGISTERS for the whole sequence (KPT*4 per thread); c stays in
//     registers too; h_{t-1} (R x H) is staged through LDS each step;
//   * the only inter-workgroup traffic is the all-gather of h_t inside a group: R*H
//     8-byte {tag = step+1, value} granules, each written by ONE write-through (sc1)
//     store and polled with sc1 loads -- the data is its own flag, one L2 round trip
//     per step, correct for any workgroup->XCD placement (guide: Guideline 16, R2).
//     Group members are placed on one XCD label (blockIdx % 8) for speed only.
//   * both directions and all groups run concurrently: ND * ceil(B/R) * G workgroups
//     (= 256 = one per CU at B=32, R=2, H=256).
// Thread map (16*HS = 512 threads): lane = 16*cgl + kq; unit u = 4*wave + cgl; the 16
// lanes of a DPP row hold the 16 K-chunks of one unit, so the K reduction is 4 DPP
// butterflies with no LDS, and lanes kq < R of each row run the cell for batch row kq.
#include "common.h"
#include <hip/hip_ext.h>
#include "p3.h"

namespace asr {

struct LstmRecArgs {
    const float* __restrict__ gates;   // [B][T][ND][4H]  x.Kx+b (read only here)
    float* __restrict__ act;           // [B][T][ND][H][8] saved {i,j,f,o | c, c_prev, -, -} or nullptr
    const float* kh[2];    // per direction: recurrent rows of the TF kernel, [H][4H]
    const int* len;        // [B]
    float* out;            // [B][Tout][ND*H]
    unsigned long long* dbg;  // diagnostic stamps (STAMP build) or nullptr
    float* hprev;          // [B][T][ND][H] or nullptr: h_{t-1} (undropped) for dK_h = Hprev^T.dG
    u64* hx;               // exchange granules [ND*NG][2][R][H]
    u64* xcc_slots;        // [ND*NG][16] XCC-ID agreement slots (zeroed with hx)
    int* err;              // set to 1 on poll timeout
    int B, T, Tout, ND, boff;
    float keep; uint32_t seed;
    // addressing: row of (b,t) in gates/act/hprev = b*sb + t*st; in out = b*osb + t*ost (leading dim ldo);
    // dropout counter = (boff+b)*dsb + (toff+t)*dst.  Batch-major encoder layer: sb = T, st = 1, osb = Tout,
    // ost = 1, ldo = ND*H, dsb = Tout, dst = 1.  Time-major (decoder LM chain): sb = 1, st = B, ...
    int sb, st, osb, ost, ldo, dsb, dst, toff;
    int ep0;               // granule tag base: tags are ep0 + step + 1 (segments of one call share a workspace zeroed once)
    const float* h0; const float* c0;      // [B][H] initial state (ND = 1) or nullptr = zeros
    float* h_last; float* c_last;          // [B][H] final state (ND = 1) or nullptr
    // lstm_rec_fwd4_kernel<IK > 0>: the input projection inside the recurrent kernel (`gates` is then not read): x [B][T][ldx],
    // kx[dir] = the input rows [in_dim][4H] of the TF kernel, bias[dir] [4H]
    const float* __restrict__ x; int ldx;
    const float* kx[2]; const float* bias[2];
    // lstm_rec_fwd4_kernel: outputs ALSO as bf16 planes (csrc/p3.h) for the GEMMs of csrc/gemm_p3.hip, or nullptr:
    // out_p3 = P3 image of out [B*Tout][ND*H] (the next layer's input), hprev_p3 = P3 image of hprev [B*T][ND*H] (then the
    // fp32 hprev is not written)
    char* out_p3; char* hprev_p3; int p3_np;
    // lstm_rec_fwd4_kernel / lstm_rec_bwd4_kernel: 20-byte records in two planes instead of the 32-byte record -- `act` holds the
    // activated gates {i,j,f,o} (16 bytes per unit-step), act_c the cell state c (4 bytes); c_prev(t) is the c of the record
    // the BPTT fetches next, and the record's 8 bytes of padding are gone (round-3 review: the record stream is what slows the
    // exchange of the BPTT most).  Same buffer: act_c = act + 4 * records.
    float* act_c;
};

__device__ __forceinline__ bool poll_granule(const u64* g, uint32_t epoch, float& val, int* err) {
    long long t0 = 0;
    ASR_RACE_HUNT_DELAY();
    for (uint32_t spins = 0;; ++spins) {
        u64 x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(x >> 32) == epoch) { val = __uint_as_float((uint32_t)x); return true; }
        ASR_POLL_BACKOFF();
        if ((spins & 1023) == 1023) {              // bounded spin: ~2 s of wall clock
            long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > 200000000LL) { *err = 11; val = 0.f; return false; }
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { val = 0.f; return false; }
        }
    }
}

// STAMP: diagnostic build only (ASR_LSTM_STAMP=1): per-phase s_memtime totals of workgroup 0
// go to a debug buffer that no other code reads; never used for timing claims.
#define ASR_STAMP(i) if (STAMP) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); stamp[i] += t__ - tlast; tlast = t__; }
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool poll_granule2(const u64* g, uint32_t epoch, float& v0, float& v1, int* err) {
    long long t0 = 0;
    const u32x4* p = reinterpret_cast<const u32x4*>(g);
    ASR_RACE_HUNT_DELAY();
    for (uint32_t spins = 0;; ++spins) {
        // 16-byte load with sc1 (system-coherent, bypasses the per-CU L1)
        u32x4 x;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(p) : "memory");
        if (x.y == epoch && x.w == epoch) { v0 = __uint_as_float(x.x); v1 = __uint_as_float(x.z); return true; }
        ASR_POLL_BACKOFF();
        if ((spins & 1023) == 1023) {
            long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > 200000000LL) { *err = 12; v0 = v1 = 0.f; return false; }
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { v0 = v1 = 0.f; return false; }
        }
    }
}

// MF (bf16 mode of the library, BASELINE config 3): the recurrent product h_{t-1}.K_h runs on the bf16 matrix pipe instead of
// the VALU -- K_h is rounded to bf16 once per launch into MFMA A-fragments (32 registers), h is rounded to bf16 by the pollers
// on its way into LDS, accumulation stays fp32: out^T[16 gate columns x batch rows] = K_h^T tile . h^T, one 16-column tile per
// wave, H/32 v_mfma_f32_16x16x32_bf16 per step (128 cycles) in place of 128 v_fma_f32 per thread plus the DPP reduction
// (1 000-1 400 of the step's ~3 300 cycles by the in-kernel stamps).  Same operand rounding as the bf16 GEMMs.
typedef __bf16 rbf16x8 __attribute__((ext_vector_type(8)));
template <int H, int HS, int R, bool STAMP = false, bool MF = false>
__global__ __launch_bounds__(16 * HS) void lstm_rec_fwd_kernel(LstmRecArgs a) {
    constexpr int KPT = H / 16;        // K values per lane
    constexpr int CS = KPT + 4;        // padded LDS chunk stride (conflict-free b128 reads)
    constexpr int G = H / HS;          // workgroups per group
    constexpr int NT = 16 * HS;
    constexpr int NCELL = R * HS;      // cell threads: the first NCELL threads
    constexpr int NCW = (NCELL + 63) / 64 * 64;   // ... rounded up to whole waves: those waves never poll
    static_assert(NCW < NT, "need at least one polling wave");
    constexpr int NPOLL = NT - NCW;    // the other waves poll; the cell waves own every store
    __shared__ __attribute__((aligned(16))) float hl[MF ? 4 : R * 16 * CS];
    __shared__ __attribute__((aligned(16))) float sums[HS * R * 4];
    __shared__ __attribute__((aligned(16))) unsigned short hb[MF ? (R + 1) * H : 8];     // MF: h as bf16, row R = zeros
    static_assert(!MF || (HS == 32 && H % 32 == 0 && R <= 16), "MF: eight waves x 16 gate columns, batch rows in the N = 16 of the MFMA");
    unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tlast = STAMP ? __builtin_amdgcn_s_memtime() : 0;

    // latency-critical serial chain: win issue arbitration against co-resident throughput kernels
    // (the weight-gradient GEMMs of the layer above run concurrently on the side stream)
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane & 15, cgl = lane >> 4;
    const int NG = (a.B + R - 1) / R;
    const int ngroups = a.ND * NG;
    int grp, mem;
    // The grid holds the groups, padded by the launcher to a multiple of 8 when the budget allows: then the members of a group
    // share blockIdx % 8 (one XCD label, the exchange stays in that XCD's L2) for ANY group count; padding workgroups leave.
    if (((gridDim.x / G) & 7) == 0) {
        mem = (blockIdx.x >> 3) % G;
        grp = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * G));
    } else {
        grp = blockIdx.x / G; mem = blockIdx.x % G;
    }
    if (grp >= ngroups) return;
    const int dir = grp / NG, bg = grp % NG;
    const int r0 = bg * R;
    const int u = wave * 4 + cgl;      // matvec role: unit of this DPP row
    const int j = mem * HS + u;
    const int H4 = 4 * H;

    // recurrent weights -> registers (once)
    float w[MF ? 1 : KPT][4];
    rbf16x8 wa[MF ? H / 32 : 1];          // MF: A fragments of this wave's 16 gate columns (row m = lane & 15 -> unit wave*4 + (m >> 2),
                                          // gate m & 3), k = 32 ks + 8 (lane >> 4) + jj
    if (MF) {
        const float* kh = a.kh[dir];
        const int m = lane & 15, kg = lane >> 4;
        const int col = (m & 3) * H + mem * HS + wave * 4 + (m >> 2);
#pragma unroll
        for (int ks = 0; ks < H / 32; ++ks)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) wa[ks][jj] = (__bf16)kh[(size_t)(ks * 32 + kg * 8 + jj) * H4 + col];
        for (int idx = tid; idx < (R + 1) * H; idx += NT) hb[idx] = 0;
    } else {
        const float* kh = a.kh[dir];
#pragma unroll
        for (int i = 0; i < KPT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) w[i][g] = kh[(size_t)(kq * KPT + i) * H4 + g * H + j];
    }
    int S = 0;
    for (int r = 0; r < R; ++r) S = max(S, (r0 + r < a.B) ? min(a.len[r0 + r], a.T) : 0);

    // cell role (threads < NCELL): row cr of the group, unit cu -> 32 consecutive units per row,
    // so every store of the step is a coalesced 128-byte (or 1 KiB for the records) segment
    const bool cell = tid < NCELL;
    // wave-uniform role flag through readfirstlane: a SCALAR branch, so the cell path carries no
    // exec-masked register initialisation (which costs a vmcnt(0) -- i.e. a drain of the
    // previous step's stores -- at every loop head)
    const bool cell_wave = __builtin_amdgcn_readfirstlane(tid) < NCW;
    const int cr = min(tid / HS, R - 1), cu = tid % HS;
    const int cb = r0 + cr;
    const int cj = mem * HS + cu;
    const int clen = (cell && cb < a.B) ? min(a.len[cb], a.T) : 0;
    const int cb_safe = min(cb, a.B - 1);
    float c = 0.f, h = 0.f;
    const bool has_init = a.h0 != nullptr;           // uniform
    if (has_init && cell && cb < a.B) { h = a.h0[(size_t)cb * H + cj]; c = a.c0[(size_t)cb * H + cj]; }
    u64* hxg = a.hx + (size_t)grp * 2 * R * H;
    const bool fast = group_shares_xcd(a.xcc_slots + (size_t)grp * 16, G, mem, tid, a.err, nullptr, (uint32_t)a.ep0, 4);
    // x.Kx+b of the NEXT step is loaded at the end of each cell phase (software pipelining): the
    // registers are loop-carried, never re-initialised, so the loop head needs no vmcnt wait and the
    // load latency hides under the next step's exchange.
    float gx0 = 0.f, gx1 = 0.f, gx2 = 0.f, gx3 = 0.f;
    auto prefetch = [&](int s) {
        const int t = dir ? (clen - 1 - s) : s;
        const int ts = min(max(t, 0), a.T - 1);
        const float* gp = a.gates + (((size_t)cb_safe * a.sb + (size_t)ts * a.st) * a.ND + dir) * H4 + cj;
        gx0 = gp[0]; gx1 = gp[H]; gx2 = gp[2 * H]; gx3 = gp[3 * H];
    };
    if (cell_wave) prefetch(0);

    for (int s = 0; s < S; ++s) {
        const bool live = cell && s < clen;
        const int t = dir ? (clen - 1 - s) : s;
        ASR_STAMP(0)
        float acc[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[r][g] = 0.f;
        if (s > 0 || has_init) {
            // (2) polling waves gather h_{s-1} (R x H granules) into LDS; they issue no stores, so
            //     their vmcnt(0) waits only for the poll itself
            if (!cell_wave) {
                // one 16-byte sc1 load = two adjacent granules (each 8-byte half is written by one
                // store and arrives untorn); R*H/2 pairs over the polling threads
                const u64* src = hxg + (size_t)((s - 1) & 1) * R * H;
                for (int pidx = tid - NCW; pidx < R * H / 2; pidx += NPOLL) {
                    const int idx = 2 * pidx;
                    const int r = idx / H, k = idx % H;
                    float v0 = 0.f, v1 = 0.f;
                    if (r0 + r < a.B) {
                        if (s > 0) poll_granule2(src + idx, (uint32_t)(a.ep0 + s), v0, v1, a.err);
                        else { v0 = a.h0[(size_t)(r0 + r) * H + k]; v1 = a.h0[(size_t)(r0 + r) * H + k + 1]; }
                    }
                    if (MF) {
                        union { __bf16 b[2]; uint32_t u; } pk;
                        pk.b[0] = (__bf16)v0; pk.b[1] = (__bf16)v1;
                        *reinterpret_cast<uint32_t*>(hb + r * H + k) = pk.u;
                    } else {
                        *reinterpret_cast<float2*>(hl + (r * 16 + k / KPT) * CS + (k % KPT)) = make_float2(v0, v1);
                    }
                }
            }
            ASR_STAMP(1)
            __syncthreads();
            ASR_STAMP(2)
            if constexpr (MF) {
                const int n = lane & 15;
                const unsigned short* hrow = hb + (n < R ? n : R) * H + 8 * (lane >> 4);
                f32x4 dacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < H / 32; ++ks) {
                    const rbf16x8 bfrag = *reinterpret_cast<const rbf16x8*>(hrow + ks * 32);
                    dacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks], bfrag, dacc, 0, 0, 0);
                }
                // D: col = lane & 15 = batch row, rows (lane >> 4) * 4 + reg = the four gates of unit wave*4 + (lane >> 4)
                if (n < R) *reinterpret_cast<float4*>(sums + (n * HS + wave * 4 + (lane >> 4)) * 4) = make_float4(dacc[0], dacc[1], dacc[2], dacc[3]);
            } else {
            // (3) h.K_h for this lane's K chunk, (4) DPP-row reduction over the 16 chunks
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float4* hp = reinterpret_cast<const float4*>(hl + (r * 16 + kq) * CS);
#pragma unroll
                for (int i4 = 0; i4 < KPT / 4; ++i4) {
                    const float4 hv = hp[i4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        acc[r][g] = fmaf(hv.x, w[4 * i4 + 0][g], acc[r][g]);
                        acc[r][g] = fmaf(hv.y, w[4 * i4 + 1][g], acc[r][g]);
                        acc[r][g] = fmaf(hv.z, w[4 * i4 + 2][g], acc[r][g]);
                        acc[r][g] = fmaf(hv.w, w[4 * i4 + 3][g], acc[r][g]);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[r][g] = row16_allreduce_sum(acc[r][g]);
            if (kq == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    *reinterpret_cast<float4*>(sums + (r * HS + u) * 4) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
            }
            }   // !MF
            ASR_STAMP(3)
            __syncthreads();   // sums complete; hl free for the next step's pollers
            ASR_STAMP(4)
        }
        // (5) the cell, on the cell waves only
        if (cell_wave && cell) {
            float4 pre = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s > 0 || has_init) pre = *reinterpret_cast<const float4*>(sums + (cr * HS + cu) * 4);
            float gi = 0.f, gj = 0.f, gf = 0.f, go = 0.f, h_old = h, c_old = c;
            if (live) {
                gi = fast_sigmoid(pre.x + gx0);
                gj = fast_tanh(pre.y + gx1);
                gf = fast_sigmoid(pre.z + gx2 + 1.0f);   // forget bias
                go = fast_sigmoid(pre.w + gx3);
                c = c * gf + gi * gj;
                h = go * fast_tanh(c);
            }
            // publish h_s FIRST (unchanged for rows past their length): it is the critical path of
            // every other workgroup of the group.  ONE 8-byte sc1 store.
            if (cb < a.B && s + 1 < S) {
                u64* dst = hxg + ((size_t)(s & 1) * R + cr) * H + cj;
                const u64 gv = ((u64)(uint32_t)(a.ep0 + s + 1) << 32) | __float_as_uint(h);
                ASR_RACE_HUNT_DELAY();
                if (fast) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst), "v"(gv) : "memory");   // stays in this XCD's L2
                else __hip_atomic_store(dst, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                 // write-through (sc1)
            }
            if (live) {       // bookkeeping stores, off the critical path
                const size_t ridx = (((size_t)cb * a.sb + (size_t)t * a.st) * a.ND + dir) * H + cj;
                float o = h;
                if (a.keep < 1.0f)
                    o *= keep_scale(a.seed, (uint32_t)((a.boff + cb) * a.dsb + (a.toff + t) * a.dst), (uint32_t)(dir * H + cj), a.keep);
                a.out[((size_t)cb * a.osb + (size_t)t * a.ost) * a.ldo + dir * H + cj] = o;
                if (a.hprev) a.hprev[ridx] = h_old;
                if (a.act) {      // one 32-byte record per (b,t,dir,unit): two 16-byte stores
                    float4* rp = reinterpret_cast<float4*>(a.act + ridx * 8);
                    rp[0] = make_float4(gi, gj, gf, go);
                    rp[1] = make_float4(c, c_old, 0.f, 0.f);
                }
            }
            if (s + 1 < S) prefetch(s + 1);
        }
        ASR_STAMP(5)
    }
    if (a.h_last && cell && cb < a.B) { a.h_last[(size_t)cb * H + cj] = h; a.c_last[(size_t)cb * H + cj] = c; }
    if (STAMP && a.dbg && blockIdx.x == 0 && (tid == 0 || tid == NT - 1)) {
        unsigned long long* d = a.dbg + (tid == 0 ? 0 : 8);
        for (int i = 0; i < 6; ++i) d[i] = stamp[i];
        d[6] = (unsigned long long)S;
    }
    // zero output past each row's length (dynamic_rnn zero-fill; also the pyramid pad frame)
    for (int r = 0; r < R; ++r) {
        if (r0 + r >= a.B) break;
        const int l = min(a.len[r0 + r], a.T);
        const int nz = a.Tout - l;
        for (int idx = tid; idx < nz * HS; idx += NT) {
            const int t = l + idx / HS, uu = idx % HS;
            a.out[((size_t)(r0 + r) * a.osb + (size_t)t * a.ost) * a.ldo + dir * H + mem * HS + uu] = 0.f;
            // hprev feeds dK_h = Hprev^T.dG over ALL rows: past the length dG is zero, so the value must merely
            // be finite (uninitialised memory may hold NaN bit patterns)
            if (a.hprev && t < a.T)
                a.hprev[(((size_t)(r0 + r) * a.sb + (size_t)t * a.st) * a.ND + dir) * H + mem * HS + uu] = 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Version 2 of the forward recurrence (round 3): one K-SLICE PER WAVE, consumed by SOURCE workgroup.
//
// The first kernel gathers all of h_{t-1} into LDS, barriers, and only then multiplies: every wave waits for the slowest
// granule of the slowest peer, and the own workgroup's eighth of h (known locally, a full exchange round trip earlier) waits
// with it.  Here wave w of a workgroup owns the 32 K-rows of K_h that belong to the units of ONE source workgroup
// ((mem + w) % G; wave 0 = the own units) for all 128 gate columns of the workgroup.  A polling wave reads exactly the
// granules its source's cell wave published with one store instruction (one 8-byte sc1 load per lane), passes them through a
// wave-private LDS row (no barrier: LDS operations of one wave execute in order) and multiplies at once.  The cell wave
// multiplies its own slice for the NEXT step right after publishing, in the shadow of the exchange.  ONE barrier per step
// (partials complete), partials double-buffered by step parity; the cell thread sums the G partials in fixed order.
// Lane map of the product: lane = 32*kh + u -- the four gate columns of unit u, K half kh (16 of the slice's 32 rows): 64
// weight registers, and only 16 x R h values read from LDS per lane (every lane reading all 32 made the step LDS-bandwidth
// bound: a broadcast ds_read_b128 still returns 1 KB).  The two K halves meet through v_permlane32_swap, which at the same
// time turns (K half, unit) lanes into the cell's (row, unit) lanes: one 16-byte partial per cell thread.
// The product runs on v_pk_fma_f32 (two fp32 FMAs per issue slot, bitwise two v_fma_f32; measured -11 % per step): with two
// rows the pair is (row 0, row 1) x one weight, with one row (gate g, gate g+1) x one h value.
// R*32 <= 64 (the cell is one wave): R in {1, 2}; larger batches per group keep the first kernel.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool poll_granule1(const u64* g, uint32_t epoch, float& val, int* err) {
    long long t0 = 0;
    ASR_RACE_HUNT_DELAY();
    for (uint32_t spins = 0;; ++spins) {
        u64 x;
        asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(g) : "memory");
        if ((uint32_t)(x >> 32) == epoch) { val = __uint_as_float((uint32_t)x); return true; }
        ASR_POLL_BACKOFF();
        if ((spins & 1023) == 1023) {
            long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > 200000000LL) { *err = 13; val = 0.f; return false; }
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { val = 0.f; return false; }
        }
    }
}

template <int H, int R, bool STAMP = false>
__global__ __launch_bounds__(2 * H) void lstm_rec_fwd2_kernel(LstmRecArgs a) {
    unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tlast = STAMP ? __builtin_amdgcn_s_memtime() : 0;
    constexpr int HS = 32;
    constexpr int NS = H / 32;         // K slices = waves = workgroups per group
    constexpr int G = NS;
    constexpr int NT = NS * 64;
    constexpr int NCELL = R * HS;      // cell threads: the first NCELL lanes of wave 0
    static_assert(R == 1 || R == 2, "the cell is one wave");
    // wave-private staging of the slice's h values: K half kh at kh*HST; R = 2: (h[0][k], h[1][k]) pairs, padded so that the two
    // halves' float4 reads fall into different banks; R = 1: 16 floats per half
    constexpr int HST = R == 2 ? 36 : 16;
    __shared__ __attribute__((aligned(16))) float hs[NS][2 * HST];
    __shared__ __attribute__((aligned(16))) float4 part[2][NS][NCELL];     // partial pre-activations by step parity

    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NG = (a.B + R - 1) / R;
    const int ngroups = a.ND * NG;
    int grp, mem;
    if (((gridDim.x / G) & 7) == 0) {
        mem = (blockIdx.x >> 3) % G;
        grp = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * G));
    } else {
        grp = blockIdx.x / G; mem = blockIdx.x % G;
    }
    if (grp >= ngroups) return;
    const int dir = grp / NG, bg = grp % NG;
    const int r0 = bg * R;
    const int H4 = 4 * H;
    const int src_wg = (mem + wave) % G;           // whose units this wave multiplies (wave 0: the own ones)
    const int u = lane & 31, kh = lane >> 5;       // product role: the four gates of unit u, K half kh of the slice

    // R = 2: wp[k][p] = (gate 2p, gate 2p+1) of K row k;  R = 1: the same pairs (B operand as it is)
    f32x2 wp[16][2];
    {
        const float* khp = a.kh[dir] + (size_t)(src_wg * 32 + kh * 16) * H4 + mem * HS + u;
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2)
                wp[k][p2] = f32x2{khp[(size_t)k * H4 + (2 * p2) * H], khp[(size_t)k * H4 + (2 * p2 + 1) * H]};
    }
    // where the h value (row r, K index k of the slice) is staged
    auto hs_slot = [&](int r, int k) { return (k >> 4) * HST + (R == 2 ? 2 * (k & 15) + r : (k & 15)); };
    int S = 0;
    for (int r = 0; r < R; ++r) S = max(S, (r0 + r < a.B) ? min(a.len[r0 + r], a.T) : 0);

    // cell role: wave 0, lane = cr*32 + cu
    const bool cell_wave = wave == 0;
    const bool cell = tid < NCELL;
    const int cr = min(tid / HS, R - 1), cu = tid % HS;
    const int cb = r0 + cr;
    const int cj = mem * HS + cu;
    const int clen = (cell && cb < a.B) ? min(a.len[cb], a.T) : 0;
    const int cb_safe = min(cb, a.B - 1);
    float c = 0.f, h = 0.f;
    const bool has_init = a.h0 != nullptr;           // uniform
    if (has_init && cell && cb < a.B) { h = a.h0[(size_t)cb * H + cj]; c = a.c0[(size_t)cb * H + cj]; }
    u64* hxg = a.hx + (size_t)grp * 2 * R * H;
    const bool fast = group_shares_xcd(a.xcc_slots + (size_t)grp * 16, G, mem, tid, a.err, nullptr, (uint32_t)a.ep0, 4);
    float gx0 = 0.f, gx1 = 0.f, gx2 = 0.f, gx3 = 0.f;
    auto prefetch = [&](int s) {
        const int t = dir ? (clen - 1 - s) : s;
        const int ts = min(max(t, 0), a.T - 1);
        const float* gp_ = a.gates + (((size_t)cb_safe * a.sb + (size_t)ts * a.st) * a.ND + dir) * H4 + cj;
        gx0 = gp_[0]; gx1 = gp_[H]; gx2 = gp_[2 * H]; gx3 = gp_[3 * H];
    };
    if (cell_wave) prefetch(0);

    // this wave's slice of h (already in hs[wave]) times its weights -> the partial of the cell thread (row, unit)
    auto slice_partial = [&](int par) {
        const f32x4* hq = reinterpret_cast<const f32x4*>(&hs[wave][kh * HST]);
        float4 out;
        if constexpr (R == 2) {
            f32x4 hall[8];               // all LDS reads in flight before the first FMA
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) hall[k2] = hq[k2];
            f32x2 pa[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};      // gate g: (row 0, row 1)
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                const f32x4 hv = hall[k2];
                const f32x2 h0 = __builtin_shufflevector(hv, hv, 0, 1), h1 = __builtin_shufflevector(hv, hv, 2, 3);
                pk_fma_blo(pa[0], h0, wp[2 * k2][0]);
                pk_fma_bhi(pa[1], h0, wp[2 * k2][0]);
                pk_fma_blo(pa[2], h0, wp[2 * k2][1]);
                pk_fma_bhi(pa[3], h0, wp[2 * k2][1]);
                pk_fma_blo(pa[0], h1, wp[2 * k2 + 1][0]);
                pk_fma_bhi(pa[1], h1, wp[2 * k2 + 1][0]);
                pk_fma_blo(pa[2], h1, wp[2 * k2 + 1][1]);
                pk_fma_bhi(pa[3], h1, wp[2 * k2 + 1][1]);
            }
            // K halves meet; lane (row = lane >> 5, unit) ends with the gate of ITS row
            out = make_float4(swap32_add(pa[0].x, pa[0].y), swap32_add(pa[1].x, pa[1].y),
                              swap32_add(pa[2].x, pa[2].y), swap32_add(pa[3].x, pa[3].y));
        } else {
            f32x4 hall[4];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) hall[k4] = hq[k4];
            f32x2 pa[2][2] = {{f32x2{0.f, 0.f}, f32x2{0.f, 0.f}}, {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}}};   // [gate pair][chain]
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const f32x4 hv = hall[k4];
                const f32x2 h0 = __builtin_shufflevector(hv, hv, 0, 1), h1 = __builtin_shufflevector(hv, hv, 2, 3);
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    pk_fma_alo(pa[p2][0], h0, wp[4 * k4 + 0][p2]);
                    pk_fma_ahi(pa[p2][1], h0, wp[4 * k4 + 1][p2]);
                    pk_fma_alo(pa[p2][0], h1, wp[4 * k4 + 2][p2]);
                    pk_fma_ahi(pa[p2][1], h1, wp[4 * k4 + 3][p2]);
                }
            }
            const f32x2 g01 = pa[0][0] + pa[0][1], g23 = pa[1][0] + pa[1][1];
            out = make_float4(swap32_add(g01.x, g01.x), swap32_add(g01.y, g01.y), swap32_add(g23.x, g23.x), swap32_add(g23.y, g23.y));
        }
        if (lane < NCELL) part[par][wave][lane] = out;
    };

    if (has_init && S > 0) {          // partials of step 0 from the given initial state
        if (lane < NCELL) {
            const int r = lane >> 5;
            hs[wave][hs_slot(r, lane & 31)] = (r0 + r < a.B) ? a.h0[(size_t)(r0 + r) * H + src_wg * 32 + (lane & 31)] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        slice_partial(0);
    }

    for (int s = 0; s < S; ++s) {
        const bool live = cell && s < clen;
        const int t = dir ? (clen - 1 - s) : s;
        const int par = s & 1;
        if (s > 0 || has_init) {
            if (!cell_wave && s > 0) {
                ASR_STAMP(0)
                // h_{s-1} of the source's units: the 8-byte granules one store instruction of its cell wave published
                if (lane < NCELL) {
                    const int r = lane >> 5;
                    float v = 0.f;
                    if (r0 + r < a.B)
                        poll_granule1(hxg + ((size_t)((s - 1) & 1) * R + r) * H + src_wg * 32 + (lane & 31), (uint32_t)(a.ep0 + s), v, a.err);
                    hs[wave][hs_slot(r, lane & 31)] = v;
                }
                ASR_STAMP(1)
                __builtin_amdgcn_wave_barrier();
                slice_partial(par);
                ASR_STAMP(2)
            }
            if (cell_wave) { ASR_STAMP(0) }
            __syncthreads();          // all G partials of this step are in part[par]
            if (cell_wave) { ASR_STAMP(1) } else { ASR_STAMP(3) }
        }
        if (cell_wave) {
            float4 pre = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((s > 0 || has_init) && cell) {
#pragma unroll
                for (int ww = 0; ww < NS; ++ww) {
                    const float4 p = part[par][ww][lane];
                    pre.x += p.x; pre.y += p.y; pre.z += p.z; pre.w += p.w;
                }
            }
            float gi = 0.f, gj = 0.f, gf = 0.f, go = 0.f, h_old = h, c_old = c;
            if (live) {
                gi = fast_sigmoid(pre.x + gx0);
                gj = fast_tanh(pre.y + gx1);
                gf = fast_sigmoid(pre.z + gx2 + 1.0f);   // forget bias
                go = fast_sigmoid(pre.w + gx3);
                c = c * gf + gi * gj;
                h = go * fast_tanh(c);
            }
            const bool more = s + 1 < S;
            if (cell && cb < a.B && more) {       // publish h_s FIRST: the critical path of every peer
                u64* dst = hxg + ((size_t)(s & 1) * R + cr) * H + cj;
                const u64 gv = ((u64)(uint32_t)(a.ep0 + s + 1) << 32) | __float_as_uint(h);
                ASR_RACE_HUNT_DELAY();
                if (fast) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst), "v"(gv) : "memory");
                else __hip_atomic_store(dst, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            ASR_STAMP(2)
            if (live) {       // bookkeeping stores, off the critical path
                const size_t ridx = (((size_t)cb * a.sb + (size_t)t * a.st) * a.ND + dir) * H + cj;
                float o = h;
                if (a.keep < 1.0f)
                    o *= keep_scale(a.seed, (uint32_t)((a.boff + cb) * a.dsb + (a.toff + t) * a.dst), (uint32_t)(dir * H + cj), a.keep);
                a.out[((size_t)cb * a.osb + (size_t)t * a.ost) * a.ldo + dir * H + cj] = o;
                // saved for the backward pass, not read again before it: streaming stores (measured -0.05 us per step)
                if (a.hprev) __builtin_nontemporal_store(h_old, a.hprev + ridx);
                if (a.act) {
                    f32x4* rp = reinterpret_cast<f32x4*>(a.act + ridx * 8);
                    __builtin_nontemporal_store(f32x4{gi, gj, gf, go}, rp);
                    __builtin_nontemporal_store(f32x4{c, c_old, 0.f, 0.f}, rp + 1);
                }
            }
            if (more) {
                prefetch(s + 1);
                ASR_STAMP(3)
                // the own units' slice for the next step, while the peers' values travel
                if (cell) hs[0][hs_slot(cr, cu)] = (cb < a.B) ? h : 0.f;
                __builtin_amdgcn_wave_barrier();
                slice_partial(par ^ 1);
                ASR_STAMP(4)
            }
        }
    }
    if (a.h_last && cell && cb < a.B) { a.h_last[(size_t)cb * H + cj] = h; a.c_last[(size_t)cb * H + cj] = c; }
    if (STAMP && a.dbg && blockIdx.x == 0 && (tid == 0 || tid == 64)) {
        unsigned long long* d = a.dbg + (tid == 0 ? 0 : 8);
        for (int i = 0; i < 6; ++i) d[i] = stamp[i];
        d[6] = (unsigned long long)S;
    }
    if (STAMP && a.dbg && blockIdx.x == 0 && lane == 0 && wave > 0) {       // every polling wave: exit->hit, product, barrier wait
        a.dbg[16 + wave] = stamp[0] + stamp[1]; a.dbg[32 + wave] = stamp[2]; a.dbg[48 + wave] = stamp[3];
    }
    // zero output past each row's length (dynamic_rnn zero-fill; also the pyramid pad frame)
    for (int r = 0; r < R; ++r) {
        if (r0 + r >= a.B) break;
        const int l = min(a.len[r0 + r], a.T);
        const int nz = a.Tout - l;
        for (int idx = tid; idx < nz * HS; idx += NT) {
            const int tt = l + idx / HS, uu = idx % HS;
            a.out[((size_t)(r0 + r) * a.osb + (size_t)tt * a.ost) * a.ldo + dir * H + mem * HS + uu] = 0.f;
            if (a.hprev && tt < a.T)
                a.hprev[(((size_t)(r0 + r) * a.sb + (size_t)tt * a.st) * a.ND + dir) * H + mem * HS + uu] = 0.f;
        }
    }
}

extern "C" int asr_get_gemm_precision(void);
// bf16 mode only: recurrent products of the FIRST-version kernels on the bf16 matrix pipe (round 2).  Off by default since
// round 3 (the fp32 version-2 recurrences are faster and exact); ASR_LSTM_MFMA=1 or asr_set_lstm_mfma(1) selects them.
static int g_lstm_mfma = -1;
// ---------------------------------------------------------------------------------------------------------------
// Groups of FOUR workgroups (H = 256: 64 units each, one batch row per group) -- the `HS = 64` candidate of the round-2 review,
// priced by scripts/micro/allgather.hip at -9 % of the bare exchange (three peers to hear from instead of seven).  Wave w
// multiplies K half (w & 1) of the 64-unit slice of source workgroup (mem + (w >> 1)) % 4: lane = own unit, 32 K rows x 4 gates
// = 128 weight registers, the 32 h values broadcast from a wave-private LDS row -- no cross-lane reduction at all; the cell
// wave (wave 0) sums the eight partials.  Wave 0 takes the first half of the own slice straight from its registers, wave 1
// polls the second half of the own slice like any other source.  Forward 1.19 -> 1.08 us per step at B = 32, T = 800 (round 3).
// ---------------------------------------------------------------------------------------------------------------
// IK > 0: the INPUT PROJECTION runs in here too (in_dim = 8 * IK; the first encoder layer, 80 log-mel inputs).  Wave w holds the
// K_x rows of inputs [w*IK, (w+1)*IK) for its lanes' units (4 * IK registers) and adds x_t . K_x to its partial; x_t is the same
// for the whole workgroup and is fetched one step ahead with scalar loads.  The product costs a wave 2 * IK packed FMAs in
// front of its poll, where it would only wait; what it saves is the 25 600 x 2 048 projection GEMM (K = 80: 89 us, bound by
// writing 210 MB of gates) and the recurrence's own reading them back -- the streaming operand that costs the exchange most.
// XPRE (round 5, with IK > 0; ASR_LSTM_XPRE=0 keeps the kernel of rounds 3-4): (i) the x part of a step's product does not
// depend on the exchange, so the polling waves form it BEFORE they poll (xpa) and its 20 packed FMAs leave the chain; (ii) the x row
// is fetched with real scalar loads (constant address space: s_load_dwordx8 + x2) into 10 SGPRs -- the "scalar" loads above were
// global_load into 10 VGPRs, counted by the vmcnt they shared with the step's record stores; (iii) the loads of the prologue
// (bias, initial state) are waited for once, in the prologue: left in flight at the loop's entry they made the compiler's merged
// wait in front of the cell a vmcnt(0) on EVERY step, i.e. a wait for the previous step's stores.  Layer 1 of config 2:
// 1.10 -> 1.02 us per step (scripts/bench_lstm.py, same box).  The instantiation with gate rows: GXL below.
// GXL (round 5, the instantiation with gate rows; ASR_LSTM_GXL=0 keeps rounds 2-4): the gate rows x_t.K_x + b of a step were
// requested one step ahead, behind that step's record stores, and waited for in front of the cell: a knock-out build that read
// the same (cached) row every step ran 0.89 instead of 1.01 us per step at T = 400 (105 MB of rows from HBM), no difference at
// T = 100 (rows still in the Infinity Cache).  Now as in the BPTT: the rows are requested TWO steps ahead and handed to the next
// cell through LDS (gxs) at the END of a step, behind the own slice's product -- the cell reads LDS and never waits on vmcnt (the
// prologue's loads are pinned as for XPRE).  1.00 -> 0.95 us per step at T = 400.  (Two other forms were measured slower: the rows
// copied between register sets right behind the publishing store, 1.007 -> 1.07-1.09; that plus the own slice's product in front
// of the record stores, the same.)
template <int IK = 0, bool XPRE = false, bool GXL = false>
__global__ __launch_bounds__(512) void lstm_rec_fwd4_kernel(LstmRecArgs a) {
    static_assert(!XPRE || IK > 0, "XPRE belongs to the instantiation with the input projection inside");
    static_assert(!GXL || IK == 0, "GXL belongs to the instantiation with gate rows");
    constexpr int H = 256, HS = 64, G = 4, NW = 8, NT = 512, H4 = 4 * H;
    constexpr bool XIN = IK > 0;
    __shared__ __attribute__((aligned(16))) float hs[NW][32];
    __shared__ __attribute__((aligned(16))) float4 part[2][NW][HS];
    __shared__ __attribute__((aligned(16))) float4 gxs[GXL ? 2 : 1][GXL ? HS : 1];      // GXL: gate rows of the next step

    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NG = a.B;
    const int ngroups = a.ND * NG;
    int grp, mem;
    if (((gridDim.x / G) & 7) == 0) { mem = (blockIdx.x >> 3) % G; grp = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * G)); }
    else { grp = blockIdx.x / G; mem = blockIdx.x % G; }
    if (grp >= ngroups) return;
    const int dir = grp / NG, cb = grp % NG;
    const int src_wg = (mem + (wave >> 1)) % G, khh = wave & 1;
    const int kbase = src_wg * HS + khh * 32;          // first K row (= hidden unit of the source) of this wave's half slice

    f32x2 wp[32][2];                                   // wp[k][p] = (gate 2p, gate 2p+1) of K row kbase + k for unit mem*64 + lane
    {
        const float* khp = a.kh[dir] + (size_t)kbase * H4 + mem * HS + lane;
#pragma unroll
        for (int k = 0; k < 32; ++k)
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2)
                wp[k][p2] = f32x2{khp[(size_t)k * H4 + (2 * p2) * H], khp[(size_t)k * H4 + (2 * p2 + 1) * H]};
    }
    const int S = min(a.len[cb], a.T);
    const bool cell_wave = wave == 0;
    const int cj = mem * HS + lane;
    f32x2 wx[XIN ? IK : 1][2];                         // wx[k][p] = (gate 2p, gate 2p+1) of input row wave*IK + k for unit cj
    float xb[XIN ? IK : 1];                            // x_t of the NEXT product, inputs [wave*IK, wave*IK + IK)
    if constexpr (XIN) {
        const float* kxp = a.kx[dir] + (size_t)(wave * IK) * H4 + cj;
#pragma unroll
        for (int k = 0; k < IK; ++k)
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2)
                wx[k][p2] = f32x2{kxp[(size_t)k * H4 + (2 * p2) * H], kxp[(size_t)k * H4 + (2 * p2 + 1) * H]};
    }
    // x row of step s (uniform over the workgroup: scalar loads)
    auto load_x = [&](int s) {
        if constexpr (XIN) {
            const int t = dir ? (S - 1 - s) : s;
            const int ts = min(max(t, 0), a.T - 1);
            const float* xp = a.x + ((size_t)cb * a.T + ts) * a.ldx + wave * IK;
            if constexpr (XPRE) {        // (x is not written during the launch)
                typedef const float __attribute__((address_space(4))) cfloat;
                const cfloat* xc = (const cfloat*)xp;
#pragma unroll
                for (int k = 0; k < IK; ++k) xb[k] = xc[k];
            } else {
#pragma unroll
                for (int k = 0; k < IK; ++k) xb[k] = xp[k];
            }
        }
    };
    // x part of a step's pre-activations (two chains per gate pair, as the h part)
    auto x_partial = [&](f32x2 (&pa)[2][2]) {
        if constexpr (XIN) {
#pragma unroll
            for (int k = 0; k + 1 < IK + 1; k += 2) {
                const f32x2 x2 = f32x2{xb[k], k + 1 < IK ? xb[k + 1] : 0.f};
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    if constexpr (XPRE) {
                        pk_fma_alo_s(pa[p2][0], x2, wx[k][p2]);
                        if (k + 1 < IK) pk_fma_ahi_s(pa[p2][1], x2, wx[k + 1][p2]);
                    } else {
                        pk_fma_alo(pa[p2][0], x2, wx[k][p2]);
                        if (k + 1 < IK) pk_fma_ahi(pa[p2][1], x2, wx[k + 1][p2]);
                    }
                }
            }
        }
    };
    float c = 0.f, h = 0.f;
    const bool has_init = a.h0 != nullptr;
    if (has_init && cell_wave) { h = a.h0[(size_t)cb * H + cj]; c = a.c0[(size_t)cb * H + cj]; }
    if constexpr (XPRE || GXL) asm volatile("" : "+v"(c), "+v"(h));       // (iii): waited for here, by every wave
    u64* hxg = a.hx + (size_t)grp * 2 * H;             // [2 parities][H] granules
    const bool fast = group_shares_xcd(a.xcc_slots + (size_t)grp * 16, G, mem, tid, a.err, nullptr, (uint32_t)a.ep0, 4);
    float gx0 = 0.f, gx1 = 0.f, gx2 = 0.f, gx3 = 0.f;
    // per-thread base pointers + 32-bit element offsets per step (the cell wave is the wave the others wait for: 64-bit index
    // arithmetic for each of its five or six addresses per step was measurable on the BPTT's cell wave, csrc/lstm_bwd.hip)
    const unsigned rstr = (unsigned)(a.st * a.ND * H);                 // elements of a [.., ND, H] array per time step
    const size_t rb0 = (((size_t)cb * a.sb) * a.ND + dir) * H + cj;     // (b, t = 0, dir, unit)
    const float* const gates_b = XIN ? nullptr : a.gates + rb0 * 4 - (size_t)cj * 3;     // ((b,0,dir) * 4H + cj)
    // (not in the instantiation with the input projection inside: its 248 registers leave no room for four more pointers)
    float* const out_b = XIN ? nullptr : a.out + ((size_t)cb * a.osb) * a.ldo + dir * H + cj;
    const unsigned ostr = (unsigned)(a.ost * a.ldo);
    float* const hprev_b = !XIN && a.hprev ? a.hprev + rb0 : nullptr;
    float* const actg_b = !XIN && a.act ? a.act + rb0 * 4 : nullptr;
    float* const actc_b = !XIN && a.act_c ? a.act_c + rb0 : nullptr;
    auto prefetch = [&](int s) {
        if constexpr (XIN) return;            // (gx = the bias, loaded once below)
        const int t = dir ? (S - 1 - s) : s;
        const unsigned ts = (unsigned)min(max(t, 0), a.T - 1);
        const float* gp_ = gates_b + ts * rstr * 4u;
        gx0 = gp_[0]; gx1 = gp_[H]; gx2 = gp_[2 * H]; gx3 = gp_[3 * H];
    };
    if (cell_wave) {
        if constexpr (XIN) {
            const float* bp = a.bias[dir] + cj; gx0 = bp[0]; gx1 = bp[H]; gx2 = bp[2 * H]; gx3 = bp[3 * H];
            if constexpr (XPRE) asm volatile("" : "+v"(gx0), "+v"(gx1), "+v"(gx2), "+v"(gx3));      // (iii)
        } else {
            prefetch(0);
            if constexpr (GXL) {
                gxs[0][lane] = make_float4(gx0, gx1, gx2, gx3);
                if (S > 1) prefetch(1);
                asm volatile("" : "+v"(gx0), "+v"(gx1), "+v"(gx2), "+v"(gx3));
            }
        }
    }

    f32x2 xpa[2][2] = {{f32x2{0.f, 0.f}, f32x2{0.f, 0.f}}, {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}}};   // XPRE: x part of the NEXT product
    auto slice_partial = [&](int par, bool xpre = false) {
        const f32x4* hq = reinterpret_cast<const f32x4*>(&hs[wave][0]);
        f32x4 hall[8];
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) hall[k4] = hq[k4];
        f32x2 pa[2][2] = {{f32x2{0.f, 0.f}, f32x2{0.f, 0.f}}, {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}}};   // [gate pair][chain]
        if (xpre) { pa[0][0] = xpa[0][0]; pa[0][1] = xpa[0][1]; pa[1][0] = xpa[1][0]; pa[1][1] = xpa[1][1]; }
        else x_partial(pa);
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
            const f32x4 hv = hall[k4];
            const f32x2 h0 = __builtin_shufflevector(hv, hv, 0, 1), h1 = __builtin_shufflevector(hv, hv, 2, 3);
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                pk_fma_alo(pa[p2][0], h0, wp[4 * k4 + 0][p2]);
                pk_fma_ahi(pa[p2][1], h0, wp[4 * k4 + 1][p2]);
                pk_fma_alo(pa[p2][0], h1, wp[4 * k4 + 2][p2]);
                pk_fma_ahi(pa[p2][1], h1, wp[4 * k4 + 3][p2]);
            }
        }
        const f32x2 g01 = pa[0][0] + pa[0][1], g23 = pa[1][0] + pa[1][1];
        part[par][wave][lane] = make_float4(g01.x, g01.y, g23.x, g23.y);
    };
    auto x_only_partial = [&](int par) {      // step 0 without an initial state: the pre-activations are x_0 . K_x alone
        f32x2 pa[2][2] = {{f32x2{0.f, 0.f}, f32x2{0.f, 0.f}}, {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}}};
        x_partial(pa);
        const f32x2 g01 = pa[0][0] + pa[0][1], g23 = pa[1][0] + pa[1][1];
        part[par][wave][lane] = make_float4(g01.x, g01.y, g23.x, g23.y);
    };

    if (S > 0) load_x(0);
    if (has_init && S > 0) {
        if (lane < 32) hs[wave][lane] = a.h0[(size_t)cb * H + kbase + lane];
        __builtin_amdgcn_wave_barrier();
        slice_partial(0);
    } else if (XIN && S > 0) x_only_partial(0);
    if (XIN && S > 1) load_x(1);          // xb always holds the x row of the NEXT product
    for (int s = 0; s < S; ++s) {
        const int t = dir ? (S - 1 - s) : s;
        const int par = s & 1;
        if (s > 0 || has_init || XIN) {
            if (!cell_wave && s > 0) {
                if constexpr (XPRE) {        // (i): x_s . K_x in front of the poll; then the request for the next row
                    xpa[0][0] = xpa[0][1] = xpa[1][0] = xpa[1][1] = f32x2{0.f, 0.f};
                    x_partial(xpa);
                    if (s + 1 < S) load_x(s + 1);
                }
                if (lane < 32) {
                    float v = 0.f;
                    poll_granule1(hxg + (size_t)((s - 1) & 1) * H + kbase + lane, (uint32_t)(a.ep0 + s), v, a.err);
                    hs[wave][lane] = v;
                }
                __builtin_amdgcn_wave_barrier();
                slice_partial(par, XPRE);
                if (XIN && !XPRE && s + 1 < S) load_x(s + 1);
            }
            __syncthreads();
        }
        if (cell_wave) {
            float4 pre = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s > 0 || has_init || XIN) {
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) {
                    const float4 p = part[par][ww][lane];
                    pre.x += p.x; pre.y += p.y; pre.z += p.z; pre.w += p.w;
                }
            }
            const float h_old = h, c_old = c;
            float4 gxv = make_float4(gx0, gx1, gx2, gx3);
            if constexpr (GXL) gxv = gxs[par][lane];
            const float gi = fast_sigmoid(pre.x + gxv.x);
            const float gj = fast_tanh(pre.y + gxv.y);
            const float gf = fast_sigmoid(pre.z + gxv.z + 1.0f);
            const float go = fast_sigmoid(pre.w + gxv.w);
            c = c * gf + gi * gj;
            h = go * fast_tanh(c);
            const bool more = s + 1 < S;
            if (more) {
                u64* dst = hxg + (size_t)(s & 1) * H + cj;
                const u64 gv = ((u64)(uint32_t)(a.ep0 + s + 1) << 32) | __float_as_uint(h);
                ASR_RACE_HUNT_DELAY();
                if (fast) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst), "v"(gv) : "memory");
                else __hip_atomic_store(dst, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if constexpr (XPRE) {        // the own slice's product in front of the record stores
                if (more) {
                    if (lane < 32) hs[0][lane] = h;
                    __builtin_amdgcn_wave_barrier();
                    slice_partial(par ^ 1);
                    if (s + 2 < S) load_x(s + 2);
                }
            }
            {
                const unsigned roff = (unsigned)t * rstr;
                float o = h;
                if (a.keep < 1.0f)
                    o *= keep_scale(a.seed, (uint32_t)((a.boff + cb) * a.dsb + (a.toff + t) * a.dst), (uint32_t)(dir * H + cj), a.keep);
                if constexpr (XIN) {
                    const size_t ridx = (((size_t)cb * a.sb + (size_t)t * a.st) * a.ND + dir) * H + cj;
                    a.out[((size_t)cb * a.osb + (size_t)t * a.ost) * a.ldo + dir * H + cj] = o;
                    if (a.hprev) __builtin_nontemporal_store(h_old, a.hprev + ridx);
                    if (a.act) {
                        __builtin_nontemporal_store(f32x4{gi, gj, gf, go}, reinterpret_cast<f32x4*>(a.act + ridx * 4));
                        __builtin_nontemporal_store(c, a.act_c + ridx);
                    }
                } else {
                    out_b[(unsigned)t * ostr] = o;
                    if (a.hprev) __builtin_nontemporal_store(h_old, hprev_b + roff);
                    if (a.act) {
                        __builtin_nontemporal_store(f32x4{gi, gj, gf, go}, reinterpret_cast<f32x4*>(actg_b + roff * 4u));
                        __builtin_nontemporal_store(c, actc_b + roff);
                    }
                }
                // the same values as bf16 planes for the GEMMs that consume them (no split inside their k-loops)
                if (a.out_p3) p3_store1(a.out_p3, p3_elem_off((size_t)cb * a.osb + (size_t)t * a.ost, dir * H + cj, a.ldo >> 3, a.p3_np), o, a.p3_np, false);
                if (a.hprev_p3) p3_store1(a.hprev_p3, p3_elem_off((size_t)cb * a.sb + (size_t)t * a.st, dir * H + cj, (a.ND * H) >> 3, a.p3_np), h_old, a.p3_np, true);
            }
            if (!XPRE && more) {
                if constexpr (!GXL) prefetch(s + 1);
                if (lane < 32) hs[0][lane] = h;          // the first half of the own slice, for the next step
                __builtin_amdgcn_wave_barrier();
                slice_partial(par ^ 1);
                if constexpr (GXL) {       // the rows requested a step ago -> LDS for the next cell; then the request for the step after it
                    gxs[par ^ 1][lane] = make_float4(gx0, gx1, gx2, gx3);
                    if (s + 2 < S) prefetch(s + 2);
                }
                if (XIN && s + 2 < S) load_x(s + 2);
            }
        }
    }
    if (a.h_last && cell_wave) { a.h_last[(size_t)cb * H + cj] = h; a.c_last[(size_t)cb * H + cj] = c; }
    {   // zero output past the row's length (dynamic_rnn zero-fill; also the pyramid pad frame)
        const int nz = a.Tout - S;
        for (int idx = tid; idx < nz * HS; idx += NT) {
            const int tt = S + idx / HS, uu = idx % HS;
            a.out[((size_t)cb * a.osb + (size_t)tt * a.ost) * a.ldo + dir * H + mem * HS + uu] = 0.f;
            if (a.hprev && tt < a.T)
                a.hprev[(((size_t)cb * a.sb + (size_t)tt * a.st) * a.ND + dir) * H + mem * HS + uu] = 0.f;
        }
        // ... and in the plane images: the workgroup's 64 columns of a row are HS / 8 chunks = HS / 8 * np contiguous pieces
        const int ppr = HS / 8 * a.p3_np;
        if (a.out_p3)
            for (int idx = tid; idx < nz * ppr; idx += NT) {
                const int tt = S + idx / ppr;
                char* rowp = a.out_p3 + p3_elem_off((size_t)cb * a.osb + (size_t)tt * a.ost, dir * H + mem * HS, a.ldo >> 3, a.p3_np);
                reinterpret_cast<uint4*>(rowp)[idx % ppr] = make_uint4(0u, 0u, 0u, 0u);
            }
        if (a.hprev_p3)
            for (int idx = tid; idx < (a.T - S) * ppr; idx += NT) {
                const int tt = S + idx / ppr;
                char* rowp = a.hprev_p3 + p3_elem_off((size_t)cb * a.sb + (size_t)tt * a.st, dir * H + mem * HS, (a.ND * H) >> 3, a.p3_np);
                reinterpret_cast<uint4*>(rowp)[idx % ppr] = make_uint4(0u, 0u, 0u, 0u);
            }
    }
}

extern "C" int asr_get_lstm_mfma(void) {
    if (g_lstm_mfma < 0) { const char* e = getenv("ASR_LSTM_MFMA"); g_lstm_mfma = (e && e[0] == '1') ? 1 : 0; }
    return g_lstm_mfma;
}
extern "C" int asr_set_lstm_mfma(int on) { g_lstm_mfma = on ? 1 : 0; return ASR_OK; }
unsigned long long* g_lstm_dbg = nullptr;      // diagnostic stamp buffer (also read by decoder_chain_bwd.hip)
extern "C" int asr_debug_set_buffer(void* p) { g_lstm_dbg = static_cast<unsigned long long*>(p); return ASR_OK; }

}  // namespace asr
int asr_lstm_max_wgs();
namespace asr {
template <int H, int R>
static int launch_rec(hipStream_t s, const LstmRecArgs& a0) {
    constexpr int HS = 32;
    LstmRecArgs a = a0;
    a.dbg = g_lstm_dbg;
    const int NG = (a.B + R - 1) / R;
    int grid = a.ND * NG * (H / HS);
    {   // whole octets of groups (see the kernel's group mapping) when they still fit the co-residency budget
        const int padded = ((a.ND * NG + 7) & ~7) * (H / HS);
        if (padded <= asr_lstm_max_wgs()) grid = padded;
    }
    // bf16 mode of the library (asr_set_gemm_precision(1)): recurrent product on the bf16 matrix pipe (H = 256 instantiation)
    const bool mf_env = asr_get_lstm_mfma() != 0;      // opt-in since round 3: the fp32 version-2 kernels are faster
    if (H == 256 && R == 2 && g_lstm_dbg && getenv("ASR_LSTM_STAMP")) {
        const char* e2 = getenv("ASR_LSTM_V2");
        if (e2 && e2[0] == '0') hipLaunchKernelGGL((lstm_rec_fwd_kernel<256, HS, 2, true>), dim3(grid), dim3(16 * HS), 0, s, a);
        else hipLaunchKernelGGL((lstm_rec_fwd2_kernel<256, 2, true>), dim3(grid), dim3(512), 0, s, a);
    }
    else if (H == 256 && mf_env && asr_get_gemm_precision() == 1)
        hipLaunchKernelGGL((lstm_rec_fwd_kernel<256, HS, R, false, true>), dim3(grid), dim3(16 * HS), 0, s, a);
    else if constexpr (R <= 2 && H <= 256) {
        // version 2 (slice per wave, one barrier per step) for groups of one or two rows (H = 512 would need 1024 threads at
        // <= 128 VGPRs and spills); ASR_LSTM_V2=0 keeps version 1
        static const bool v2 = [] { const char* e = getenv("ASR_LSTM_V2"); return !(e && e[0] == '0'); }();
        if (v2) hipLaunchKernelGGL((lstm_rec_fwd2_kernel<H, R>), dim3(grid), dim3(2 * H), 0, s, a);
        else hipLaunchKernelGGL((lstm_rec_fwd_kernel<H, HS, R>), dim3(grid), dim3(16 * HS), 0, s, a);
    } else
    hipLaunchKernelGGL((lstm_rec_fwd_kernel<H, HS, R>), dim3(grid), dim3(16 * HS), 0, s, a);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

template <int H>
static int launch_rec_h(hipStream_t s, const LstmRecArgs& a, int R) {
    switch (R) {
        case 1: return launch_rec<H, 1>(s, a);
        case 2: return launch_rec<H, 2>(s, a);
        case 4: return launch_rec<H, 4>(s, a);
        case 8:
            if constexpr (H == 512) return ASR_EUNSUPPORTED;     // (would spill: the caller runs four rows per group in twice the launches)
            else return launch_rec<H, 8>(s, a);
    }
    return ASR_EINVAL;
}

}  // namespace asr

extern "C" int asr_gemm_f32(void*, int, int, int, int, int, const float*, int, const float*, int,
                            float*, int, const float*, int);

// rows per group: smallest R whose grid fits one workgroup per CU (256 CUs); override for tuning
// resident-workgroup budget of one recurrent launch (tuning knob; default one workgroup per CU)
int asr_lstm_max_wgs() { return asr::resident_wg_budget(); }
int asr_lstm_pick_rows(int B, int ND, int G) {
    if (const char* e = getenv("ASR_LSTM_R")) { int r = atoi(e); if (r == 1 || r == 2 || r == 4 || r == 8) return r; }
    for (int R : {1, 2, 4, 8})
        if (ND * ((B + R - 1) / R) * G <= asr_lstm_max_wgs()) return R;
    return 8;
}

static size_t lstm_hx_bytes(int B, int H, int ndir) {
    // worst case R=1: ND * B groups * 2 parities * H granules; R>1 never needs more
    return (size_t)ndir * (size_t)((B + 7) / 8 * 8) * 2 * H * sizeof(u64);
}
extern "C" size_t asr_lstm_ws_bytes(int B, int H, int ndir) {
    return lstm_hx_bytes(B, H, ndir) + (size_t)ndir * (size_t)((B + 7) / 8 * 8) * 16 * sizeof(u64);   // + XCC slots
}

// bf16-plane operands of one layer (include/e2e_asr_hip.h: asr_lstm_p3)
struct asr_lstm_p3 {
    int np; const void* x_p3; int x_cols; const void* kxT_p3; void* out_p3; void* hprev_p3;
    void* dg_p3; const void* kxu_p3; const int* colmap;
};
extern "C" int asr_gemm_p3_kk(void* stream, int M, int N, int K, const void* A, int lda8, const void* B, int ldb8, int np,
                              float* C, int ldc, const float* bias, int accumulate, int splits);
bool asr_lstm_g4_selected(int B, int H, int ndir);
// 1 when asr_lstm_layer_fwd_p3 / _bwd_p3 take plane operands for this shape (the groups-of-four recurrent kernels write planes)
extern "C" int asr_lstm_p3_supported(int B, int T, int in_dim, int H, int ndir) {
    return H == 256 && ndir == 2 && (B * T) % 128 == 0 && asr_lstm_g4_selected(B, H, ndir) ? 1 : 0;
}
extern "C" int asr_lstm_layer_fwd_p3(void* stream, const float* x, int B, int T, int in_dim, int ldx,
                                  const int* len, int H, int ndir,
                                  const float* kernel_fw, const float* bias_fw,
                                  const float* kernel_bw, const float* bias_bw,
                                  float* out, int Tout, float* gates, float* act, float* hprev,
                                  void* hx_ws, size_t hx_bytes, int* err_flag,
                                  float keep_prob, unsigned seed, const float* kx_cat, const float* bias_cat,
                                  const asr_lstm_p3* p3);
extern "C" int asr_lstm_layer_fwd(void* stream, const float* x, int B, int T, int in_dim, int ldx,
                                  const int* len, int H, int ndir,
                                  const float* kernel_fw, const float* bias_fw,
                                  const float* kernel_bw, const float* bias_bw,
                                  float* out, int Tout, float* gates, float* act, float* hprev,
                                  void* hx_ws, size_t hx_bytes, int* err_flag,
                                  float keep_prob, unsigned seed, const float* kx_cat, const float* bias_cat) {
    return asr_lstm_layer_fwd_p3(stream, x, B, T, in_dim, ldx, len, H, ndir, kernel_fw, bias_fw, kernel_bw, bias_bw, out, Tout, gates,
                                 act, hprev, hx_ws, hx_bytes, err_flag, keep_prob, seed, kx_cat, bias_cat, nullptr);
}
extern "C" int asr_race_hunt_build(void) { return ASR_RACE_HUNT; }
bool asr_lstm_g4_selected(int B, int H, int ndir) {
    const char* e = getenv("ASR_LSTM_G4");
    const char* v2e = getenv("ASR_LSTM_V2");
    const char* ce = getenv("ASR_LSTM_G4_CHUNKS");
    const int rpl = asr_lstm_max_wgs() / (4 * ndir);
    const char* ag = getenv("ASR_BPTT_AG");       // (the forward and the BPTT of a layer must agree: they share a record format)
    return !(e && e[0] == '0') && !(v2e && v2e[0] == '0') && !(ag && ag[0] == '0') && !(asr::g_lstm_dbg && getenv("ASR_LSTM_STAMP")) &&
           !(asr::asr_get_lstm_mfma() != 0 && asr::asr_get_gemm_precision() == 1) && H == 256 && rpl >= 1 && (B + rpl - 1) / rpl <= (ce ? atoi(ce) : 4);
}
extern "C" int asr_lstm_layer_fwd_p3(void* stream, const float* x, int B, int T, int in_dim, int ldx,
                                  const int* len, int H, int ndir,
                                  const float* kernel_fw, const float* bias_fw,
                                  const float* kernel_bw, const float* bias_bw,
                                  float* out, int Tout, float* gates, float* act, float* hprev,
                                  void* hx_ws, size_t hx_bytes, int* err_flag,
                                  float keep_prob, unsigned seed, const float* kx_cat, const float* bias_cat,
                                  const asr_lstm_p3* p3) {
    using namespace asr;
    if (!x || !len || !kernel_fw || !bias_fw || !out || !gates || !hx_ws || !err_flag) return ASR_EINVAL;
    if (ndir != 1 && ndir != 2) return ASR_EINVAL;
    if (ndir == 2 && (!kernel_bw || !bias_bw)) return ASR_EINVAL;
    if (B <= 0 || T <= 0 || in_dim <= 0 || Tout < T || ldx < in_dim) return ASR_EINVAL;
    if (H != 64 && H != 128 && H != 256 && H != 512) return ASR_EUNSUPPORTED;
    const bool prezeroed = (hx_bytes & ASR_WS_PREZEROED) != 0;
    hx_bytes &= ~ASR_WS_PREZEROED;
    if (hx_bytes < asr_lstm_ws_bytes(B, H, ndir)) return ASR_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int H4 = 4 * H;
    // groups of four workgroups (lstm_rec_fwd4_kernel) -- and with 80 inputs (the first encoder layer) the input projection
    // inside that kernel: no projection GEMM, no gates written or read (ASR_LSTM_XIN=0: the GEMM, as for every other width)
    bool g4 = false, xin = false;
    {
        const char* xe = getenv("ASR_LSTM_XIN");
        g4 = asr_lstm_g4_selected(B, H, ndir);
        xin = g4 && in_dim == 80 && !(xe && xe[0] == '0');
    }
    if (p3 && (p3->out_p3 || p3->hprev_p3) && !g4) return ASR_EUNSUPPORTED;       // only the groups-of-four kernel writes planes
    if (p3 && (p3->np < 1 || p3->np > 3)) return ASR_EINVAL;
    // input projection for all timesteps: gates[b,t,dir,:] = x[b,t,:] . K_x + bias
    if (xin) {
    } else if (p3 && p3->x_p3 && p3->kxT_p3 && bias_cat && (B * T) % 128 == 0 && (ndir * H4) % 256 == 0 && in_dim % 16 == 0 &&
               p3->x_cols >= in_dim && p3->x_cols % 8 == 0) {
        // both operands arrive as bf16 planes (x: written by the layer below; K_x^T: split once per step): csrc/gemm_p3.hip
        int rc = asr_gemm_p3_kk(stream, B * T, ndir * H4, in_dim, p3->x_p3, p3->x_cols / 8, p3->kxT_p3, in_dim / 8, p3->np,
                                gates, ndir * H4, bias_cat, 0, 1);
        if (rc) return rc;
    } else if (ndir == 2 && kx_cat && bias_cat) {
        // both directions as ONE product with N = 8H: gates rows are [fw 4H | bw 4H] and the caller supplies the input rows of
        // the two kernels side by side ([in, 8H]) -- twice the tiles per launch (less tile quantisation: 1 600 instead of
        // 2 x 800 on 512 resident slots at layer 2) and X streamed once
        int rc = asr_gemm_f32(stream, 0, 0, B * T, 2 * H4, in_dim, x, ldx, kx_cat, 2 * H4, gates, 2 * H4, bias_cat, 0);
        if (rc) return rc;
    } else
    for (int d = 0; d < ndir; ++d) {
        int rc = asr_gemm_f32(stream, 0, 0, B * T, H4, in_dim, x, ldx, d ? kernel_bw : kernel_fw, H4,
                              gates + (size_t)d * H4, ndir * H4, d ? bias_bw : bias_fw, 0);
        if (rc) return rc;
    }
    if (!prezeroed && hipMemsetAsync(hx_ws, 0, asr_lstm_ws_bytes(B, H, ndir), s) != hipSuccess) return ASR_ELAUNCH;
    LstmRecArgs a;
    a.gates = gates;
    a.kh[0] = kernel_fw + (size_t)in_dim * H4;
    a.kh[1] = ndir == 2 ? kernel_bw + (size_t)in_dim * H4 : nullptr;
    a.len = len; a.out = out; a.act = act; a.hprev = hprev; a.boff = 0; a.hx = static_cast<u64*>(hx_ws); a.err = err_flag;
    a.xcc_slots = reinterpret_cast<u64*>(static_cast<char*>(hx_ws) + lstm_hx_bytes(B, H, ndir));
    a.B = B; a.T = T; a.Tout = Tout; a.ND = ndir; a.keep = keep_prob; a.seed = seed;
    a.sb = T; a.st = 1; a.osb = Tout; a.ost = 1; a.ldo = ndir * H; a.dsb = Tout; a.dst = 1; a.toff = 0;
    a.h0 = a.c0 = nullptr; a.h_last = a.c_last = nullptr; a.ep0 = 0;
    a.x = nullptr; a.ldx = 0; a.kx[0] = a.kx[1] = nullptr; a.bias[0] = a.bias[1] = nullptr;
    a.out_p3 = p3 ? static_cast<char*>(p3->out_p3) : nullptr;
    a.hprev_p3 = p3 ? static_cast<char*>(p3->hprev_p3) : nullptr;
    a.p3_np = p3 ? p3->np : 0; a.act_c = nullptr;
    if (a.hprev_p3) a.hprev = nullptr;          // (the fp32 copy has no reader then)
    if (g4) {   // one row per group; larger batches as consecutive launches over ranges of 32 rows (measured against the first-version
                // kernels with four / eight rows per group: B = 64 2.25 vs 2.69 us per step of a layer, B = 128 4.85 vs 4.97)
        const int rpl = asr_lstm_max_wgs() / (4 * ndir);
        a.dbg = nullptr;
        a.x = x; a.ldx = ldx; a.kx[0] = kernel_fw; a.kx[1] = kernel_bw; a.bias[0] = bias_fw; a.bias[1] = bias_bw;
        for (int b0 = 0; b0 < B; b0 += rpl) {
            LstmRecArgs c = a;
            c.B = (B - b0 < rpl) ? (B - b0) : rpl;
            c.gates = a.gates + (size_t)b0 * T * ndir * H4;
            c.x = x + (size_t)b0 * T * ldx;
            c.len = len + b0;
            c.out = out + (size_t)b0 * Tout * ndir * H;
            c.act = act ? act + (size_t)b0 * T * ndir * H * 4 : nullptr;                       // gates plane, then the c plane
            c.act_c = act ? act + (size_t)B * T * ndir * H * 4 + (size_t)b0 * T * ndir * H : nullptr;
            c.hprev = a.hprev ? a.hprev + (size_t)b0 * T * ndir * H : nullptr;
            if (a.out_p3) c.out_p3 = a.out_p3 + (size_t)b0 * Tout * (ndir * H / 8) * 16 * a.p3_np;
            if (a.hprev_p3) c.hprev_p3 = a.hprev_p3 + (size_t)b0 * T * (ndir * H / 8) * 16 * a.p3_np;
            c.boff = b0;
            const int groups = ndir * c.B;
            const int padded = ((groups + 7) & ~7) * 4;
            const int grid = padded <= asr_lstm_max_wgs() ? padded : groups * 4;
            // (timing events = the dispatch's own start / stop signals, as in csrc/lstm_bwd.hip)
            static const bool ext_ev = [] { const char* e = getenv("ASR_EXT_EVENTS"); return !(e && e[0] == '0'); }();
            hipEvent_t ev_a = nullptr, ev_b = nullptr;
            if (ext_ev) prof_launch_events(ASR_PROF_LSTM_REC_FWD, &ev_a, &ev_b);
            else prof_begin(ASR_PROF_LSTM_REC_FWD, s);
            static const bool xpre = [] { const char* e = getenv("ASR_LSTM_XPRE"); return !(e && e[0] == '0'); }();
            if (xin && xpre) hipExtLaunchKernelGGL((asr::lstm_rec_fwd4_kernel<10, true>), dim3(grid), dim3(512), 0, s, ev_a, ev_b, 0, c);
            else if (xin) hipExtLaunchKernelGGL(asr::lstm_rec_fwd4_kernel<10>, dim3(grid), dim3(512), 0, s, ev_a, ev_b, 0, c);
            else {
                static const bool gxl = [] { const char* e = getenv("ASR_LSTM_GXL"); return !(e && e[0] == '0'); }();
                if (gxl) hipExtLaunchKernelGGL((asr::lstm_rec_fwd4_kernel<0, false, true>), dim3(grid), dim3(512), 0, s, ev_a, ev_b, 0, c);
                else hipExtLaunchKernelGGL(asr::lstm_rec_fwd4_kernel<0>, dim3(grid), dim3(512), 0, s, ev_a, ev_b, 0, c);
            }
            if (!ext_ev) prof_end(ASR_PROF_LSTM_REC_FWD, s);
            ASR_CHECK_LAUNCH();
            if (b0 + rpl < B && hipMemsetAsync(hx_ws, 0, asr_lstm_ws_bytes(B, H, ndir), s) != hipSuccess) return ASR_ELAUNCH;
        }
        return ASR_OK;
    }
    int R = asr_lstm_pick_rows(B, ndir, H / 32);
    // H = 512: eight rows per group need 256 registers + scratch; four rows in twice the launches take the same time (B = 64:
    // 5.80 against 5.85 ms for an 800-step layer) without it
    if (H == 512 && R > 4) R = 4;
    // batches too large for one resident grid run as consecutive launches over row ranges
    const int max_groups = asr_lstm_max_wgs() / (H / 32) / ndir;
    if (max_groups < 1) return ASR_EUNSUPPORTED;      // one group (both directions) cannot be co-resident on this device
    const int rows_per_launch = max_groups * R;
    for (int b0 = 0; b0 < B; b0 += rows_per_launch) {
        LstmRecArgs c = a;
        const int nb = (B - b0 < rows_per_launch) ? (B - b0) : rows_per_launch;
        c.B = nb;
        c.gates = a.gates + (size_t)b0 * T * ndir * H4;
        c.len = len + b0;
        c.out = out + (size_t)b0 * Tout * ndir * H;
        c.act = act ? act + (size_t)b0 * T * ndir * H * 8 : nullptr;
        c.hprev = hprev ? hprev + (size_t)b0 * T * ndir * H : nullptr;
        c.boff = b0;
        int rc;
        prof_begin(ASR_PROF_LSTM_REC_FWD, s);
        switch (H) {
            case 64: rc = launch_rec_h<64>(s, c, R); break;
            case 128: rc = launch_rec_h<128>(s, c, R); break;
            case 256: rc = launch_rec_h<256>(s, c, R); break;
            default: rc = launch_rec_h<512>(s, c, R); break;
        }
        prof_end(ASR_PROF_LSTM_REC_FWD, s);
        if (rc) return rc;
        if (b0 + rows_per_launch < B &&
            hipMemsetAsync(hx_ws, 0, asr_lstm_ws_bytes(B, H, ndir), s) != hipSuccess) return ASR_ELAUNCH;
    }
    return ASR_OK;
}


// Time-major single-direction launch with an initial state: the decoder's LM cell chain over the steps
// [toff, toff + T) of a segment (attn_decoder.py:116-121 lm_cell; all rows run every step).  Pointers are
// already offset to the segment's first step; rows of step t are at t*B + b.  gates holds x.K_x + b on entry.
// full_len: device [B] ints >= T.  Requires one resident grid: asr_lstm_tm_supported(B, H).
bool asr_lstm_tm_supported(int B, int H) {
    if (H != 64 && H != 128 && H != 256 && H != 512) return false;
    const int G = H / 32;
    int R = asr_lstm_pick_rows(B, 1, G);
    if (H == 512 && R > 2) R = 2;          // the BPTT's cap (csrc/lstm_bwd.hip): forward and backward take the same batches
    return ((B + R - 1) / R) * G <= asr_lstm_max_wgs();
}
int asr_lstm_rec_fwd_tm(hipStream_t s, const float* gates, const float* kh, const int* full_len, float* out, int ldo,
                        float* act, float* hprev, const float* h0, const float* c0, float* h_last, float* c_last,
                        void* hx_ws, int* err, int B, int T, int H, int toff, float keep, unsigned seed) {
    using namespace asr;
    if (!asr_lstm_tm_supported(B, H)) return ASR_EUNSUPPORTED;
    // the workspace is zeroed once per sequence (first segment); later segments use fresh granule tags (ep0 = toff)
    if (toff == 0 && hipMemsetAsync(hx_ws, 0, asr_lstm_ws_bytes(B, H, 1), s) != hipSuccess) return ASR_ELAUNCH;
    LstmRecArgs a;
    a.gates = gates; a.act = act; a.kh[0] = kh; a.kh[1] = nullptr; a.len = full_len; a.out = out; a.dbg = nullptr;
    a.hprev = hprev; a.hx = static_cast<u64*>(hx_ws); a.err = err;
    a.xcc_slots = reinterpret_cast<u64*>(static_cast<char*>(hx_ws) + lstm_hx_bytes(B, H, 1));
    a.B = B; a.T = T; a.Tout = T; a.ND = 1; a.boff = 0; a.keep = keep; a.seed = seed;
    a.sb = 1; a.st = B; a.osb = 1; a.ost = B; a.ldo = ldo; a.dsb = 1; a.dst = B; a.toff = toff;
    a.h0 = h0; a.c0 = c0; a.h_last = h_last; a.c_last = c_last; a.ep0 = toff;
    a.x = nullptr; a.ldx = 0; a.kx[0] = a.kx[1] = nullptr; a.bias[0] = a.bias[1] = nullptr;
    a.out_p3 = a.hprev_p3 = nullptr; a.p3_np = 0; a.act_c = nullptr;
    int R = asr_lstm_pick_rows(B, 1, H / 32);
    if (H == 512 && R > 4) R = 4;          // (asr_lstm_tm_supported has checked that the batch fits)
    switch (H) {
        case 64: return launch_rec_h<64>(s, a, R);
        case 128: return launch_rec_h<128>(s, a, R);
        case 256: return launch_rec_h<256>(s, a, R);
        default: return launch_rec_h<512>(s, a, R);
    }
}
