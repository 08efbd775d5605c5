// Persistent attention-decoder chain, backward: all T steps of the reverse-time recursion in ONE
// launch (the gradient of csrc/decoder_chain.hip; replaces the per-step launches K1/S1/S2 of
// decoder_bwd.hip when the shape is supported).
//
// Per step i (T-1 .. 0), for a group of R = 2 utterances over G = 16 workgroups (same ownership as the
// forward: H/16 units, A/16 attention columns, ceil(Te/16) positions, D/16 context columns each):
//   AG  gather dG of step i+1 (all 4H positions of both rows; all-gather, published by the owners of the units)
//       [dh_i | dctx_carry] for my units / my D-slice = dG_{i+1} . [K_h ; W_inp[P:].K_x]^T over ALL gate columns, with the
//       rows of my own H/16 + D/16 outputs resident in registers
//       dctx_tot = dctx_ap[i] + dctx_carry   (saved: the caller turns sum_i alpha_i^T.dctx_i into denc); my partial of the
//       softmax scalar S = sum_tau alpha.dalpha = dctx_tot . ctx_i  (identity: ctx = sum alpha.enc)
//   AG  gather dctx_tot over all D columns (+ the 16 partials of S)
//       dalpha for MY positions over all D (enc rows of my positions in LDS); de = alpha (dalpha - S)
//   (d) tanh backward on my positions: ds = de v (1 - th^2); dhf slice += ds (LDS, written once at the
//       end); dv += de th (registers); partial dy[a] over my positions               -> X2 publish
//   AR  all-reduce of dy over the group's 16 position slices (round 5, ARED: one hop; every workgroup publishes its partial
//       [R][A] and gathers all 16 -- rounds 2-4: X2 reduce-scatter to the owners of the A-slices, then an all-gather);
//       my A-slice is saved (dW_att = q^T.dy after the loop); dq_att for MY units = dy . W_att[unit, :]
//       cell pointwise backward -> dG (in place over the saved gates), dc carry; the 4 dG values of each unit are
//       published first (AG of the next step)
// Three of the four exchanges are all-gathers of the SMALL vector a workgroup owns (64 / 64+1 / 16 publishing stores) with
// the contraction done by the consumer; the first version reduce-scattered partial sums of everything (2848 publishing
// stores per workgroup and step by one wave, 12.6 us per step; 6.3 with the four exchanges of rounds 2-4, 5.7 with three).  Granules are tagged 8-byte {step, value} words;
// gathering threads keep all their loads in flight and sum in fixed order (reproducible).  Wave 0 owns every global store.
// dx = dG.K_x^T and dlm_out = dx.W_inp[:P]^T are GEMMs after the loop.
#include "common.h"
#include <algorithm>
#include <cstdlib>

extern "C" int asr_decoder_chain_supported(int B, int Te, int D, int A, int H);
namespace asr { extern unsigned long long* g_lstm_dbg; }

namespace asr {

struct ChainBwdArgs {
    float* gates;              // [T][B][4H] in: activated gates; out: dG
    const float* dec_c;        // [T][B][H]
    const float* alpha;        // [T][B][Te]
    const float* y;            // [T][B][A]
    const float* ctx;          // [T][B][D]
    const float* dqc;          // [T][B][H+D]  dq_ap | dctx_ap (hoisted GEMMs)
    const float* wh;           // [H][4H]
    const float* wc;           // [D][4H]
    const float* w_att; const float* v;    // [H][A], [A]
    const float* hf;           // [B][Te][A]
    const float* enc;          // [B][Te][D]
    const int* enc_len;
    float* dY;                 // [T][B][A]
    float* dctx;               // [T][B][D]
    float* dhf;                // [B][Te][A]   (written once)
    float* dv_part;            // [groups*16][A]
    u64* gx; u64* xcc_slots; int* err;
    int B, Te, T;
    unsigned long long* dbg;   // STAMP build only
    int g0, ng;                // groups [g0, g0 + ng) of the batch in this launch
};

typedef unsigned int u32x4d __attribute__((ext_vector_type(4)));

// Sum over the 16 sources of one slot: all 16 loads in flight (compiler-tracked relaxed agent-scope
// atomic loads = global_load_dwordx2 sc1), re-polled together, fixed summation order (reproducible).
template <int src_stride>
__device__ __forceinline__ bool gather16_one(const u64* base, uint32_t epoch, float& s0, int* err) {
    long long t0 = 0;
    ASR_RACE_HUNT_DELAY();
    for (uint32_t spins = 0;; ++spins) {
        u64 x[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) x[m] = __hip_atomic_load(base + m * src_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool all = true;
#pragma unroll
        for (int m = 0; m < 16; ++m) all &= (uint32_t)(x[m] >> 32) == epoch;
        if (all) {
            float a0 = 0.f;
#pragma unroll
            for (int m = 0; m < 16; ++m) a0 += __uint_as_float((uint32_t)x[m]);
            s0 = a0;
            return true;
        }
        ASR_POLL_BACKOFF();
        if ((spins & 1023) == 1023) {
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > 200000000LL) { *err = 21; s0 = 0.f; return false; }
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { s0 = 0.f; return false; }
        }
    }
}
__device__ __forceinline__ void pubg(u64* dst, uint32_t epoch, float v, bool fast) {
    const u64 gv = ((u64)epoch << 32) | __float_as_uint(v);
    ASR_RACE_HUNT_DELAY();
    if (fast) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst), "v"(gv) : "memory");
    else __hip_atomic_store(dst, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// STAMP: diagnostic instantiation (ASR_CHAIN_STAMP=1 + asr_debug_set_buffer): per-phase s_memtime totals of wave 0 of
// workgroup 0 (phase = code between two consecutive barriers of a step); never used for timing claims.
// R: utterances per group (2: up to 16 encoder positions per workgroup, Te <= 256; 1: 32 positions, Te <= 512).
// PASSES = 2 (round 5, R = 2 only): two utterances per group with up to 32 positions each -- the 64 (utterance, position) slots
// of a workgroup take the 32 DPP rows of the position phases in two passes (pass p = utterance p); the hf slice and the dhf
// accumulator of a slot live in the REGISTERS of the 16 lanes that own it in both phases (8 + 8 values per pass) instead of
// 2 x 32 KB of LDS, which leaves room for both utterances' enc rows up to 400 positions (config 4's phone task: one launch of
// 16 groups instead of two launches of one utterance per group).
// ARED (round 5; ASR_CHAIN_BWD_ARED=0 keeps the two hops): the reduce-scatter of the partial dy to the owners of the A-slices and the
// all-gather of the reduced dy that followed it -- two store -> L2 -> poll hops and two barriers -- as ONE all-reduce: every
// workgroup publishes its partial dy of the group's rows ([R][A] floats = one wave instruction of tagged quads, summed over its
// DPP rows by the publishing lane itself) and gathers all 16 partials (four lanes per quad, four sources each, summed in a fixed
// order and joined by two DPP adds).  16 KB polled per workgroup and step instead of 2 x 2 KB, one hop (~1 us) less.
template <int H, int D, int A, int R = 2, bool STAMP = false, int PASSES = 1, bool ARED = false>
__global__ __launch_bounds__(512) void decoder_chain_bwd_kernel(ChainBwdArgs a) {
    unsigned int stamp[16] = {0};
    unsigned long long tlast = 0;
    int sph = 0;
#define CHAIN_STAMP() if (STAMP) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); stamp[sph & 15] += (unsigned int)(t__ - tlast); tlast = t__; ++sph; }
    constexpr int G = 16, NT = 512;
    static_assert(R == 1 || R == 2, "rows per group");
    static_assert(PASSES == 1 || (PASSES == 2 && R == 2), "two passes: two utterances of up to 32 positions");
    constexpr bool WIDE = PASSES == 2;
    constexpr int HS = H / G, AS = A / G, DS = D / G;
    constexpr int NGT = 192;                          // gathering threads of the all-gathers: waves 1-3
    constexpr int N4 = 4 * H;                         // dG positions per row: p = 4*unit + gate
    constexpr int PC = N4 / 64;                       // positions per lane in the [dh|dctx] contraction (64 chunks = one wave)
    constexpr int CSB = PC + 4;                       // padded LDS chunk stride
    constexpr int NOUT = HS + DS;                     // outputs owned by this workgroup: dh of my units | dctx of my columns
    constexpr int OPW = (NOUT + 7) / 8;               // outputs per wave
    // dG travels as 16-byte quads (the four gate values of a unit), each value carrying a 1-bit tag in its lowest mantissa bit
    // -- the format of the encoder BPTT (csrc/lstm_bwd.hip): half the bytes and loads of {tag32, value32} granules
    constexpr int NQUAD4 = R * N4 / 4;                // quads gathered per step
    constexpr int NPP4 = (NQUAD4 + NGT - 1) / NGT;    // ... per gathering thread
    static_assert(NPP4 >= 1 && NPP4 <= 3, "quads per gathering thread");
    constexpr int MAXTS = 32 * PASSES / R;            // position slots per utterance and workgroup
    constexpr int DYR = WIDE ? 8 : MAXTS;             // partial-dy rows per utterance in dyrow (WIDE: one per wave)
    constexpr int AL = A / 16;                        // a values per lane in the tanh phase
    constexpr int H4 = 4 * H;
    // slots per (dst, src): even counts so that pairs never straddle
    constexpr int D1 = D + G;                         // all-gather row of dctx_tot: [D values | G partials of S]
    constexpr int NPAIR1 = R * D1 / 2;
    constexpr int NPP1 = (NPAIR1 + NGT - 1) / NGT;    // pairs per gathering thread (waves 1-3)
    constexpr int DL = D / 16;                        // context columns per lane in the dalpha contraction
    constexpr int S2 = R * ((AS + 1) & ~1);           // X2: [r][dy a-slice]
    constexpr int AS2 = (AS + 1) & ~1, HS2 = (HS + 1) & ~1;
    static_assert(HS * G == H && AS * G == A && DS * G == D && PC % 4 == 0, "sizes");
    constexpr int NPAR = R * D1 + G * G * S2 + R * A + R * N4;  // granules per parity per group (dctx_tot, dy, dG all-gathers + the dy reduce-scatter X2)

    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* lds_flag = reinterpret_cast<int*>(smem);
    float* dhl = smem + 4;                            // dh for my units [R][HS]
    float* sp = dhl + R * HS2;                        // S [2 + r] (+pad)
    float* del = sp + 4;                              // de for my positions [R][MAXTS]
    // One region, two lives: [dctall | dga | fpart] serve the first phases of a step (dG gather, [dh|dctx] contraction,
    // dctx_tot gather, dalpha), dyrow the tanh / dy-reduce phases after them (and the dv reduction of the epilogue); every
    // hand-over between the two is separated by barriers.  (Without the overlay Te = 256 needs 168 KB of LDS.)
    float* uni = del + R * MAXTS;
    float* dctall = uni;                              // dctx_tot of both rows over ALL context columns + the G partials of S [R][D1]
    float* dga = dctall + ((R * D1 + 3) & ~3);        // gathered dG of the later step [R][64 chunks][CSB]
    float* fpart = dga + R * 64 * CSB;                // [dh|dctx] partial sums [NOUT][R][4 DPP rows]
    float* dyrow = uni;                               // per DPP row partial dy [32 rows][A]
    constexpr int UNI_A = ((R * D1 + 3) & ~3) + R * 64 * CSB + ((NOUT * R * 4 + 3) & ~3);
    constexpr int UNI = UNI_A > 32 * A ? UNI_A : 32 * A;
    float* dyp = uni + UNI;                           // partial dy [R][A]                            -> X2 publish
    float* dys = dyp + R * A;                         // dy for my a-slice [R][AS2]
    float* dyall = dys + R * AS2;                     // dy of both rows over ALL attention columns [R][A] (all-gather)
    float* dql = dyall + R * A;                       // dq_att for my units [R][HS2]
    float* hfl = dql + R * HS2;                       // hf slice [R][MAXTS][A]          (registers when WIDE)
    float* dhfl = hfl + (WIDE ? 0 : R * MAXTS * A);   // dhf accumulator [R][MAXTS][A]   (registers when WIDE)
    const int Te = a.Te;
    const int TS = (Te + G - 1) / G;
    float* wal = dhfl + (WIDE ? 0 : R * MAXTS * A);   // W_att rows of my units [HS][A]
    float* vl = wal + H * AS;                         // v [A]
    float* encl = vl + A;                             // enc rows of my positions, all context columns [R][TS][D]
    // operands of the CURRENT step that depend on no exchange: fetched one step ahead by the prefetch waves
    // (threads >= 256, which neither poll nor store) -- item order below = LDS order
    const int TeP = (Te + 1) & ~1;
    // double-buffered by step parity: the LAST phase of a step (cell) still reads its copy while the prefetch waves hand the
    // next step's operands over at the top of the next step (a single buffer there was a race; behind the step's first
    // barrier it put the prefetch issue on the critical path)
    float* pfl_base = encl + R * TS * D;
    const int nitems = R * A + R * TeP + 2 * R * DS + R * HS * 4 + 3 * R * HS;
    const int nitemsP = (nitems + 3) & ~3;

    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane & 15, row = tid >> 4;
    const int NG = a.ng;
    int grp, mem;
    if (((gridDim.x / G) & 7) == 0) { mem = (blockIdx.x >> 3) % G; grp = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * G)); }   // (grid padded to whole
    else { grp = blockIdx.x / G; mem = blockIdx.x % G; }                                                                      //  octets of groups: lstm.hip)
    if (grp >= NG) return;
    grp += a.g0;
    const int r0 = grp * R;
    const bool wave0 = __builtin_amdgcn_readfirstlane(tid) < 64;
    const bool wave1 = !wave0 && __builtin_amdgcn_readfirstlane(tid) < 128;     // the gathering wave
    const int brow0 = min(r0, a.B - 1), brow1 = min(r0 + 1, a.B - 1);
    const int blen0 = min(max(a.enc_len[brow0], 0), Te);
    const int blen1 = (r0 + 1 < a.B) ? min(max(a.enc_len[brow1], 0), Te) : 0;
    auto browf = [&](int r) { return r ? brow1 : brow0; };
    auto blenf = [&](int r) { return r ? blen1 : blen0; };
    auto rok = [&](int r) { return r < R && r0 + r < a.B; };
    u64* gbase = a.gx + (size_t)grp * 2 * NPAR;
    const bool fast = group_shares_xcd(a.xcc_slots + (size_t)grp * 16, G, mem, tid, a.err, lds_flag, 0, 2);

    // ---- resident operands
    // [dh|dctx] of the EARLIER step = dG . [K_h ; WK_c]^T for my own outputs: wave w owns outputs w*OPW .. +OPW-1 (unit
    // rows of K_h, then context rows of WK_c), lane l the positions [l*PC, l*PC + PC) of all 4H gate columns
    // (outputs in PAIRS: the contraction runs on v_pk_fma_f32 with the dG value broadcast to both halves -- half the issue slots
    //  of the 2 x OPW x PC scalar FMAs per thread, same order of summation, same bits; round 5)
    constexpr int OPW2 = (OPW + 1) / 2;
    f32x2 wo[OPW2][PC];
#pragma unroll
    for (int i = 0; i < 2 * OPW2; ++i) {
        const int o = wave * OPW + i;
        const bool ook = i < OPW && o < NOUT;
        const float* wr = (o < HS) ? a.wh + (size_t)(mem * HS + (ook ? o : 0)) * H4 : a.wc + (size_t)(mem * DS + (ook ? o - HS : 0)) * H4;
#pragma unroll
        for (int q = 0; q < PC; ++q) {
            const int pos = lane * PC + q;
            const float w = ook ? wr[(pos & 3) * H + (pos >> 2)] : 0.f;
            if (i & 1) wo[i >> 1][q].y = w; else wo[i >> 1][q].x = w;
        }
    }
    for (int idx = tid; idx < HS * A; idx += NT) wal[idx] = a.w_att[(size_t)(mem * HS + idx / A) * A + idx % A];
    for (int idx = tid; idx < A; idx += NT) vl[idx] = a.v[idx];
    // hf / enc slices -> LDS, dhf accumulator = 0
    const int tau0 = mem * TS;
    if constexpr (!WIDE) {
        for (int idx = tid; idx < R * MAXTS * A; idx += NT) {
            const int r = idx / (MAXTS * A), rem = idx % (MAXTS * A), tl = rem / A, aa = rem % A;
            const int tau = tau0 + tl;
            hfl[idx] = (tl < TS && tau < Te) ? a.hf[((size_t)browf(r) * Te + tau) * A + aa] : 0.f;
            dhfl[idx] = 0.f;
        }
    }
    for (int idx = tid; idx < R * TS * D; idx += NT) {
        const int r = idx / (TS * D), tl = (idx / D) % TS, dcol = idx % D, tau = mem * TS + tl;
        encl[idx] = tau < Te ? a.enc[((size_t)browf(r) * Te + tau) * D + dcol] : 0.f;
    }
    // position-phase mapping: slot = pass * 32 + DPP row -> (tl = slot % MAXTS, r = slot / MAXTS), lane kq -> AL columns a
    // WIDE: hf / dhf of my slots in registers (element q of lane kq = column (q / 4) * 64 + kq * 4 + (q & 3), as the LDS form)
    static_assert(!WIDE || AL % 4 == 0, "register-resident hf: float4 chunks");
    float hreg[WIDE ? PASSES : 1][AL], greg[WIDE ? PASSES : 1][AL];
    if constexpr (WIDE) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int slot = p * 32 + row, r = slot / MAXTS, tl = slot % MAXTS, tau = tau0 + tl;
            const bool ok = tl < TS && tau < Te;
#pragma unroll
            for (int c = 0; c < AL / 4; ++c) {
                float4 h4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) h4 = *reinterpret_cast<const float4*>(a.hf + ((size_t)browf(r) * Te + tau) * A + c * 64 + kq * 4);
                hreg[p][4 * c] = h4.x; hreg[p][4 * c + 1] = h4.y; hreg[p][4 * c + 2] = h4.z; hreg[p][4 * c + 3] = h4.w;
                greg[p][4 * c] = 0.f; greg[p][4 * c + 1] = 0.f; greg[p][4 * c + 2] = 0.f; greg[p][4 * c + 3] = 0.f;
            }
        }
    }
    float dvacc[AL];
#pragma unroll
    for (int q = 0; q < AL; ++q) dvacc[q] = 0.f;
    // cell threads (wave 0): (unit = tid % HS, r = tid / HS)
    const bool cell = tid < R * HS;
    const int cr = cell ? tid / HS : 0, cuu = tid % HS;
    const int cj = mem * HS + cuu;
    const int cb = r0 + cr;
    const bool cb_ok = cell && cb < a.B;
    float dc = 0.f;
    constexpr int NPF = WIDE ? 6 : 5, PF0 = 256, PFN = NT - PF0;      // waves 4-7 prefetch (the host checks nitems <= NPF * PFN)
    const bool pfw = __builtin_amdgcn_readfirstlane(tid) >= PF0;
    // Per (thread, slot) descriptors of the prefetched items, built once: address of the item at time index 0 and its
    // stride per time step (low bit: "the item of step t lives at t-1 and is zero at t = 0", the c_{t-1} rows).  The
    // per-step fetch is then one LDS read + one multiply-add per item (decoding the item list every step put ~2000
    // cycles of integer work of the prefetch waves in front of the step's first barrier).
    unsigned long long* pfa = reinterpret_cast<unsigned long long*>(pfl_base + 2 * nitemsP);
    int* pfs = reinterpret_cast<int*>(pfa + NPF * PFN);
    if (pfw) {
        for (int j = 0; j < NPF; ++j) {
            int idx = tid - PF0 + PFN * j;
            const float* ptr = nullptr; long long stride = 0; int flag = 0;
            if (idx < nitems) {
                if (idx < R * A) { const int r = idx / A; if (rok(r)) { ptr = a.y + (size_t)browf(r) * A + idx % A; stride = (long long)a.B * A; } }
                else if ((idx -= R * A) < R * TeP) {
                    const int r = idx / TeP, tau = idx % TeP;
                    if (rok(r) && tau < Te) { ptr = a.alpha + (size_t)browf(r) * Te + tau; stride = (long long)a.B * Te; }
                } else if ((idx -= R * TeP) < R * DS) {
                    const int r = idx / DS; if (rok(r)) { ptr = a.dqc + (size_t)browf(r) * (H + D) + H + mem * DS + idx % DS; stride = (long long)a.B * (H + D); }
                } else if ((idx -= R * DS) < R * DS) {
                    const int r = idx / DS; if (rok(r)) { ptr = a.ctx + (size_t)browf(r) * D + mem * DS + idx % DS; stride = (long long)a.B * D; }
                } else if ((idx -= R * DS) < R * HS * 4) {
                    const int r = idx / (HS * 4), uu = (idx >> 2) % HS, g = idx & 3;
                    if (rok(r)) { ptr = a.gates + (size_t)browf(r) * H4 + g * H + mem * HS + uu; stride = (long long)a.B * H4; }
                } else {
                    idx -= R * HS * 4;
                    const int which = idx / (R * HS), rem = idx % (R * HS), r = rem / HS, uu = rem % HS;
                    if (rok(r)) {
                        if (which == 2) { ptr = a.dqc + (size_t)browf(r) * (H + D) + mem * HS + uu; stride = (long long)a.B * (H + D); }
                        else { ptr = a.dec_c + (size_t)browf(r) * H + mem * HS + uu; stride = (long long)a.B * H; flag = which; }
                    }
                }
            }
            unsigned long long ad = reinterpret_cast<unsigned long long>(ptr);
            if (ptr && flag) ad -= (unsigned long long)stride * 4ull;                  // item of step t lives at t - 1
            pfa[j * PFN + tid - PF0] = ad;
            pfs[j * PFN + tid - PF0] = (int)(stride * 2) | flag;
        }
    }
    auto pf_fetch = [&](int t, int j) -> float {
        const unsigned long long ad = pfa[j * PFN + tid - PF0];
        const int st = pfs[j * PFN + tid - PF0];
        if (ad == 0 || ((st & 1) && t == 0)) return 0.f;
        return *reinterpret_cast<const float*>(ad + (unsigned long long)t * (unsigned long long)(st >> 1) * 4ull);
    };
    float pfr[NPF];
#pragma unroll
    for (int j = 0; j < NPF; ++j) pfr[j] = pfw ? pf_fetch(a.T - 1, j) : 0.f;
    __syncthreads();

    if (STAMP) tlast = __builtin_amdgcn_s_memtime();
    const int tid_outer = tid;
    for (int s = 0; s < a.T; ++s) {
        sph = 0;
        // thread indices re-derived per step from an opaque copy: addresses built from them are then not hoisted out of the loop
        // and spilled (csrc/decoder_greedy.hip has the measurement: scratch reloads in front of the critical stores)
        int tz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(tz));
        const int tid = tid_outer + tz, lane = tid & 63;
        const int kq = lane & 15, row = tid >> 4;
        float* const pfl = pfl_base + (s & 1) * nitemsP;
        float* const yl = pfl;                            // y_i [R][A]
        float* const alf = yl + R * A;                    // alpha_i [R][TeP]
        float* const dqcx = alf + R * TeP;                // dctx_ap slice [R][DS]
        float* const ctxl = dqcx + R * DS;                // ctx_i slice [R][DS]
        float* const gl = ctxl + R * DS;                  // activated gates of my units [R][HS][4]
        float* const cl = gl + R * HS * 4;                // c_i [R][HS]
        float* const cpl = cl + R * HS;                   // c_{i-1} [R][HS]
        float* const dqa = cpl + R * HS;                  // dq_ap of my units [R][HS]
        const int i = a.T - 1 - s;
        const uint32_t ep = (uint32_t)(s + 1);
        u64* gpar = gbase + (size_t)(s & 1) * NPAR;
        u64* g1 = gpar; u64* g2 = g1 + R * D1; u64* g3 = g2 + G * G * S2; u64* g4 = g3 + R * A;
        const u64* g4prev = gbase + (size_t)((s - 1) & 1) * NPAR + (NPAR - R * N4);     // dG of the previous (later-time) step
        // ---- operands of this step that do not depend on any exchange (waves >= 2 fetch them)
        // ---- hand the prefetched operands of THIS step over to this parity's LDS copy, and fetch the next step's
        if (pfw) {
#pragma unroll
            for (int j = 0; j < NPF; ++j) {
                const int idx = tid - PF0 + PFN * j;
                if (idx < nitems) pfl[idx] = pfr[j];
            }
            if (s + 1 < a.T) {
#pragma unroll
                for (int j = 0; j < NPF; ++j) pfr[j] = pf_fetch(i - 1, j);
            }
        }
        // ---- gather dG of the later step (all 4H positions of both rows; published by their owners), waves 1-2:
        // all of a thread's granule loads in flight, re-polled together until every tag matches
        if (s > 0 && tid >= 64 && tid < 64 + NGT) {
            typedef unsigned int u32x4q __attribute__((ext_vector_type(4)));
            const uint32_t* src = reinterpret_cast<const uint32_t*>(g4prev);
            const uint32_t want = ((((uint32_t)(s - 1)) >> 1) & 1u) ^ 1u;
            bool need[NPP4];
            const u32x4q* qp[NPP4];
#pragma unroll
            for (int j = 0; j < NPP4; ++j) {
                const int qidx = tid - 64 + NGT * j;
                need[j] = qidx < NQUAD4 && rok((4 * qidx) / N4);
                qp[j] = reinterpret_cast<const u32x4q*>(src) + min(qidx, NQUAD4 - 1);
                if (qidx < NQUAD4 && !need[j]) {
                    const int idx = 4 * qidx, r = idx / N4, pos = idx % N4;
                    *reinterpret_cast<float4*>(dga + (r * 64 + pos / PC) * CSB + (pos % PC)) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            long long t0w = 0;
            ASR_RACE_HUNT_DELAY();
            for (uint32_t spins = 0;; ++spins) {
                u32x4q x[NPP4];
                if constexpr (NPP4 == 1) {
                    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x[0]) : "v"(qp[0]) : "memory");
                } else if constexpr (NPP4 == 2) {
                    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(x[0]), "=&v"(x[1]) : "v"(qp[0]), "v"(qp[1]) : "memory");
                } else {
                    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\t"
                                 "global_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]) : "v"(qp[0]), "v"(qp[1]), "v"(qp[2]) : "memory");
                }
                bool pending = false;
#pragma unroll
                for (int j = 0; j < NPP4; ++j) {
                    if (!need[j]) continue;
                    const uint32_t bits = (x[j].x & 1u) + (x[j].y & 1u) + (x[j].z & 1u) + (x[j].w & 1u);
                    if (bits == 4u * want) {
                        const int idx = 4 * (tid - 64 + NGT * j), r = idx / N4, pos = idx % N4;
                        *reinterpret_cast<float4*>(dga + (r * 64 + pos / PC) * CSB + (pos % PC)) =
                            make_float4(__uint_as_float(x[j].x & ~1u), __uint_as_float(x[j].y & ~1u),
                                        __uint_as_float(x[j].z & ~1u), __uint_as_float(x[j].w & ~1u));
                        need[j] = false;
                    } else pending = true;
                }
                if (!pending) break;
                ASR_POLL_BACKOFF();
                if ((spins & 1023) == 1023) {
                    const long long now = wall_clock64();
                    if (t0w == 0) t0w = now;
                    else if (now - t0w > 200000000LL) { *a.err = 52; break; }
                    if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                }
            }
        }
        __syncthreads();
        CHAIN_STAMP()
        // ---- [dh_i | dctx_carry_i] for my outputs = dG_{i+1} . [K_h ; WK_c]^T: 64 position chunks per wave, DPP-row
        // butterflies, the 4 rows of the wave meet in LDS (summed in fixed order by the consumers below)
        if (s > 0) {
            float acc[R][2 * OPW2];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                f32x2 ap[OPW2];
#pragma unroll
                for (int i = 0; i < OPW2; ++i) ap[i] = f32x2{0.f, 0.f};
                const f32x4* dp = reinterpret_cast<const f32x4*>(dga + (r * 64 + lane) * CSB);
#pragma unroll
                for (int q4 = 0; q4 < PC / 4; ++q4) {
                    const f32x4 dv = dp[q4];
                    const f32x2 lo = __builtin_shufflevector(dv, dv, 0, 1), hi = __builtin_shufflevector(dv, dv, 2, 3);
#pragma unroll
                    for (int i = 0; i < OPW2; ++i) {
                        pk_fma_alo(ap[i], lo, wo[i][4 * q4 + 0]);
                        pk_fma_ahi(ap[i], lo, wo[i][4 * q4 + 1]);
                        pk_fma_alo(ap[i], hi, wo[i][4 * q4 + 2]);
                        pk_fma_ahi(ap[i], hi, wo[i][4 * q4 + 3]);
                    }
                }
#pragma unroll
                for (int i = 0; i < OPW2; ++i) { acc[r][2 * i] = ap[i].x; acc[r][2 * i + 1] = ap[i].y; }
            }
            {   // butterflies over the 16 lanes of a DPP row, four values per fused block (common.h)
                constexpr int NV = R * 2 * OPW2;
                float* f = &acc[0][0];
#pragma unroll
                for (int g = 0; g + 4 <= NV; g += 4) row16_allreduce_sum4(f[g], f[g + 1], f[g + 2], f[g + 3]);
#pragma unroll
                for (int g = NV & ~3; g < NV; ++g) f[g] = row16_allreduce_sum(f[g]);
            }
            if ((lane & 15) == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int i = 0; i < OPW; ++i)
                        if (wave * OPW + i < NOUT) fpart[((wave * OPW + i) * R + r) * 4 + (lane >> 4)] = acc[r][i];
            }
        }
        __syncthreads();
        CHAIN_STAMP()
        // dh for my units (threads 64 ..) and dctx_tot = dctx_ap + carry (D-slice; wave 0), saved for the denc GEMM;
        // partial S = dctx_tot . ctx_i
        if (tid >= 64 && tid < 64 + R * HS) {
            const int r = (tid - 64) / HS, u = (tid - 64) % HS;
            float x = 0.f;
            if (s > 0) { const float4 v = *reinterpret_cast<const float4*>(fpart + (u * R + r) * 4); x = (v.x + v.y) + (v.z + v.w); }
            dhl[r * HS2 + u] = x;
        }
        if (wave0) {
            float sprt = 0.f;
            const int r = lane / 32, dd0 = lane % 32;      // 2 row slots x 32 lanes (R = 1: the second slot idles)
            for (int dd = dd0; dd < DS; dd += 32) {
                float x = 0.f;
                if (s > 0 && r < R) { const float4 v = *reinterpret_cast<const float4*>(fpart + ((HS + dd) * R + r) * 4); x = (v.x + v.y) + (v.z + v.w); }
                if (rok(r)) {
                    const size_t rowi = (size_t)i * a.B + r0 + r;
                    x += dqcx[r * DS + dd];
                    pubg(g1 + (size_t)r * D1 + mem * DS + dd, ep, x, fast);           // all-gather of dctx_tot: published first
                    a.dctx[rowi * D + mem * DS + dd] = x;
                    sprt = fmaf(x, ctxl[r * DS + dd], sprt);
                }
            }
            // reduce the 32 lanes of each half-wave: this workgroup's partial of S = dctx_tot . ctx_i, gathered with dctx_tot
            sprt += lane_xor16(sprt); sprt = row16_allreduce_sum(sprt);
            if (dd0 == 0 && rok(r)) pubg(g1 + (size_t)r * D1 + D + mem, ep, sprt, fast);
        }
        // ---- gather dctx_tot of both rows over all D columns (+ the G partials of S): waves 1-3, all loads in flight
        if (tid >= 64 && tid < 64 + NGT) {
            bool need[NPP1];
#pragma unroll
            for (int j = 0; j < NPP1; ++j) {
                const int pidx = tid - 64 + NGT * j;
                need[j] = pidx < NPAIR1 && rok((2 * pidx) / D1);
                if (pidx < NPAIR1 && !need[j]) *reinterpret_cast<float2*>(dctall + 2 * pidx) = make_float2(0.f, 0.f);
            }
            long long t0w = 0;
            ASR_RACE_HUNT_DELAY();
            for (uint32_t spins = 0;; ++spins) {
                u64 x[NPP1][2];
#pragma unroll
                for (int j = 0; j < NPP1; ++j) {
                    const int pidx = min(tid - 64 + NGT * j, NPAIR1 - 1);
                    x[j][0] = __hip_atomic_load(g1 + 2 * pidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    x[j][1] = __hip_atomic_load(g1 + 2 * pidx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                bool pending = false;
#pragma unroll
                for (int j = 0; j < NPP1; ++j) {
                    if (!need[j]) continue;
                    if ((uint32_t)(x[j][0] >> 32) == ep && (uint32_t)(x[j][1] >> 32) == ep) {
                        *reinterpret_cast<float2*>(dctall + 2 * (tid - 64 + NGT * j)) =
                            make_float2(__uint_as_float((uint32_t)x[j][0]), __uint_as_float((uint32_t)x[j][1]));
                        need[j] = false;
                    } else pending = true;
                }
                if (!pending) break;
                ASR_POLL_BACKOFF();
                if ((spins & 1023) == 1023) {
                    const long long now = wall_clock64();
                    if (t0w == 0) t0w = now;
                    else if (now - t0w > 200000000LL) { *a.err = 53; break; }
                    if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                }
            }
        }
        __syncthreads();
        CHAIN_STAMP()
        // ---- dalpha for my positions over ALL context columns (DPP row = one (row r, position tl), 16 lanes over D) and the
        // softmax scalar S = sum of the G partials (fixed order): del = dalpha, sp[2 + r] = S
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int slot = p * 32 + row, r = slot / MAXTS, tl = slot % MAXTS, tau = tau0 + tl;
            float x = 0.f;
            if (tl < TS && tau < blenf(r)) {
                const float* dr = dctall + r * D1;
                const float* er = encl + ((size_t)r * TS + tl) * D;
                if (DL % 4 == 0) {
#pragma unroll
                    for (int c = 0; c < DL / 4; ++c) {
                        const int d0 = c * 64 + kq * 4;
                        const float4 d4 = *reinterpret_cast<const float4*>(dr + d0);
                        const float4 e4 = *reinterpret_cast<const float4*>(er + d0);
                        x = fmaf(d4.x, e4.x, x); x = fmaf(d4.y, e4.y, x); x = fmaf(d4.z, e4.z, x); x = fmaf(d4.w, e4.w, x);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < DL; ++q) x = fmaf(dr[kq * DL + q], er[kq * DL + q], x);
                }
            }
            x = row16_allreduce_sum(x);
            if (kq == 0) {
                del[r * MAXTS + tl] = x;
                if (tl == 0) {
                    float st = 0.f;
#pragma unroll
                    for (int m = 0; m < G; ++m) st += dctall[r * D1 + D + m];
                    sp[2 + r] = st;
                }
            }
        }
        __syncthreads();
        CHAIN_STAMP()
        // ---- (d) tanh backward on my positions
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int slot = p * 32 + row, r = slot / MAXTS, tl = slot % MAXTS, tau = tau0 + tl;
            float de = 0.f;
            if (tl < TS && tau < blenf(r)) de = alf[r * TeP + tau] * (del[r * MAXTS + tl] - sp[2 + r]);
            // element q of lane kq is column acol(q): float4 chunks 64 columns apart, so that the 16 lanes of a DPP row
            // touch 16 consecutive 16-byte words (the strided kq*AL+q map was a 4-way LDS bank conflict on every access)
            float* hrow = hfl + (r * MAXTS + tl) * A;
            float* grow = dhfl + (r * MAXTS + tl) * A;
            const float* yrow = yl + r * A;
            if (AL % 4 == 0) {
#pragma unroll
                for (int c = 0; c < AL / 4; ++c) {
                    const int a0 = c * 64 + kq * 4;
                    float4 h4, g4;
                    if constexpr (WIDE) {
                        h4 = make_float4(hreg[p][4 * c], hreg[p][4 * c + 1], hreg[p][4 * c + 2], hreg[p][4 * c + 3]);
                        g4 = make_float4(greg[p][4 * c], greg[p][4 * c + 1], greg[p][4 * c + 2], greg[p][4 * c + 3]);
                    } else {
                        h4 = *reinterpret_cast<const float4*>(hrow + a0);
                        g4 = *reinterpret_cast<float4*>(grow + a0);
                    }
                    const float4 y4 = *reinterpret_cast<const float4*>(yrow + a0);
                    const float4 v4 = *reinterpret_cast<const float4*>(vl + a0);
                    float4 d4;
                    float th;
#define ASR_TBW(f, j) th = fast_tanh(h4.f + y4.f); d4.f = de * v4.f * (1.f - th * th); g4.f += d4.f; dvacc[4 * c + j] = fmaf(de, th, dvacc[4 * c + j]);
                    ASR_TBW(x, 0) ASR_TBW(y, 1) ASR_TBW(z, 2) ASR_TBW(w, 3)
#undef ASR_TBW
                    if constexpr (WIDE) {
                        greg[p][4 * c] = g4.x; greg[p][4 * c + 1] = g4.y; greg[p][4 * c + 2] = g4.z; greg[p][4 * c + 3] = g4.w;
                        // the four DPP rows of a wave hold four positions of the SAME utterance (r = p): summed here in a
                        // fixed order, one partial-dy row per wave
                        d4.x += lane_xor16(d4.x); d4.y += lane_xor16(d4.y); d4.z += lane_xor16(d4.z); d4.w += lane_xor16(d4.w);
                        d4.x += lane_xor32(d4.x); d4.y += lane_xor32(d4.y); d4.z += lane_xor32(d4.z); d4.w += lane_xor32(d4.w);
                        if (lane < 16) *reinterpret_cast<float4*>(dyrow + (p * DYR + wave) * A + a0) = d4;
                    } else {
                        *reinterpret_cast<float4*>(grow + a0) = g4;
                        *reinterpret_cast<float4*>(dyrow + row * A + a0) = d4;
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < AL; ++q) {
                    const int a0 = kq * AL + q;
                    const float th = fast_tanh(hrow[a0] + yrow[a0]);
                    const float ds = de * vl[a0] * (1.f - th * th);
                    grow[a0] += ds;
                    dvacc[q] = fmaf(de, th, dvacc[q]);
                    dyrow[row * A + a0] = ds;
                }
            }
        }
        __syncthreads();
        CHAIN_STAMP()
        if constexpr (ARED) {
            constexpr int NQ = R * A / 4;                 // quads of the partial dy of the group's rows
            static_assert(A % 4 == 0 && NQ <= 64 && 4 * NQ <= NT - 64, "one publishing wave instruction, four polling lanes per quad");
            uint32_t* const ar = reinterpret_cast<uint32_t*>(g2);          // [source workgroup][R * A] tagged floats
            const uint32_t tb = ((((uint32_t)s) >> 1) & 1u) ^ 1u;
            typedef unsigned int u32x4a __attribute__((ext_vector_type(4)));
            if (wave0 && lane < NQ) {      // my partial: quad `lane` summed over the DPP rows (fixed order), published at once
                const int idx = 4 * lane, r = idx / A, aa = idx % A;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t < DYR; ++t) {
                    const float4 v = *reinterpret_cast<const float4*>(dyrow + (r * DYR + t) * A + aa);
                    x.x += v.x; x.y += v.y; x.z += v.z; x.w += v.w;
                }
                const u32x4a q0 = {(__float_as_uint(x.x) & ~1u) | tb, (__float_as_uint(x.y) & ~1u) | tb,
                                   (__float_as_uint(x.z) & ~1u) | tb, (__float_as_uint(x.w) & ~1u) | tb};
                uint32_t* dst = ar + (size_t)mem * (R * A) + idx;
                ASR_RACE_HUNT_DELAY();
                if (fast) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(q0) : "memory");
                else {
                    __hip_atomic_store(dst + 0, q0.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 1, q0.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 2, q0.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 3, q0.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (tid >= 64 && tid < 64 + 4 * NQ) {      // gather: lane (quad q, part p) takes sources 4 p .. 4 p + 3
                const int q = (tid - 64) >> 2, part = (tid - 64) & 3;
                const u32x4a* qp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) qp[j] = reinterpret_cast<const u32x4a*>(ar + (size_t)(4 * part + j) * (R * A) + 4 * q);
                u32x4a xq[4];
                long long t0w = 0;
                ASR_RACE_HUNT_DELAY();
                for (uint32_t spins = 0;; ++spins) {
                    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                                 "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(xq[0]), "=&v"(xq[1]), "=&v"(xq[2]), "=&v"(xq[3])
                                 : "v"(qp[0]), "v"(qp[1]), "v"(qp[2]), "v"(qp[3]) : "memory");
                    uint32_t bits = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) bits += (xq[j].x & 1u) + (xq[j].y & 1u) + (xq[j].z & 1u) + (xq[j].w & 1u);
                    if (bits == 16u * tb) break;
                    ASR_POLL_BACKOFF();
                    if ((spins & 1023) == 1023) {
                        const long long now = wall_clock64();
                        if (t0w == 0) t0w = now;
                        else if (now - t0w > 200000000LL) { *a.err = 55; break; }
                        if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    }
                }
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x.x += __uint_as_float(xq[j].x & ~1u); x.y += __uint_as_float(xq[j].y & ~1u);
                    x.z += __uint_as_float(xq[j].z & ~1u); x.w += __uint_as_float(xq[j].w & ~1u);
                }
                // the four parts of a quad sit in four adjacent lanes: (p0 + p1) + (p2 + p3)
                x.x += dpp_mov<0xB1>(x.x); x.y += dpp_mov<0xB1>(x.y); x.z += dpp_mov<0xB1>(x.z); x.w += dpp_mov<0xB1>(x.w);
                x.x += dpp_mov<0x4E>(x.x); x.y += dpp_mov<0x4E>(x.y); x.z += dpp_mov<0x4E>(x.z); x.w += dpp_mov<0x4E>(x.w);
                if (part == 0) *reinterpret_cast<float4*>(dyall + 4 * q) = x;
            }
            __syncthreads();
            CHAIN_STAMP()
            if (wave0 && tid < R * AS) {      // my A-slice of dy, saved for dW_att = q^T . dy after the loop
                const int r = tid / AS, al = tid % AS;
                if (rok(r)) a.dY[((size_t)i * a.B + r0 + r) * A + mem * AS + al] = dyall[r * A + mem * AS + al];
            }
        } else {
        for (int idx = tid; idx < R * A; idx += NT) {        // partial dy[r][a] = sum over my positions (MAXTS DPP rows; WIDE: 8 waves)
            const int r = idx / A, aa = idx % A;
            float x = 0.f;
#pragma unroll
            for (int t = 0; t < DYR; ++t) x += dyrow[(r * DYR + t) * A + aa];
            dyp[idx] = x;
        }
        __syncthreads();
        CHAIN_STAMP()
        if (wave0) {     // X2 publish: to the owner of each A-slice
            for (int idx = lane; idx < R * A; idx += 64) {
                const int r = idx / A, aa = idx % A, md = aa / AS, al = aa % AS;
                if (rok(r)) pubg(g2 + ((size_t)md * G + mem) * S2 + r * AS2 + al, ep, dyp[idx], fast);
            }
        }
        // ---- X2 gather: dy for my A-slice
        if (wave1 && tid - 64 < S2) {
            const int slot = tid - 64, r = slot / AS2, q = slot % AS2;
            float v0 = 0.f;
            if (rok(r) && q < AS) gather16_one<S2>(g2 + ((size_t)mem * G) * S2 + slot, ep, v0, a.err);
            dys[r * AS2 + q] = v0;
        }
        __syncthreads();
        CHAIN_STAMP()
        if (wave0 && tid < R * AS) {      // save dy (dW_att = q^T . dy after the loop) and publish it (all-gather of dy)
            const int r = tid / AS, al = tid % AS;
            if (rok(r)) {
                const float x = dys[r * AS2 + al];
                a.dY[((size_t)i * a.B + r0 + r) * A + mem * AS + al] = x;
                pubg(g3 + (size_t)r * A + mem * AS + al, ep, x, fast);
            }
        }
        // ---- gather dy over all attention columns (R*A granules, pairs; wave 1)
        if (wave1) {
            constexpr int NPY = (R * A / 2 + 63) / 64;
            for (int j = 0; j < NPY; ++j) {
                const int pidx = tid - 64 + 64 * j;
                if (pidx < R * A / 2) {
                    const int idx = 2 * pidx, r = idx / A;
                    float v0 = 0.f, v1 = 0.f;
                    if (rok(r)) {
                        long long t0w = 0;
                        ASR_RACE_HUNT_DELAY();
                        for (uint32_t spins = 0;; ++spins) {
                            const u64 x0 = __hip_atomic_load(g3 + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const u64 x1 = __hip_atomic_load(g3 + idx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((uint32_t)(x0 >> 32) == ep && (uint32_t)(x1 >> 32) == ep) {
                                v0 = __uint_as_float((uint32_t)x0); v1 = __uint_as_float((uint32_t)x1); break;
                            }
                            ASR_POLL_BACKOFF();
                            if ((spins & 1023) == 1023) {
                                const long long now = wall_clock64();
                                if (t0w == 0) t0w = now;
                                else if (now - t0w > 200000000LL) { *a.err = 54; break; }
                                if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                            }
                        }
                    }
                    *reinterpret_cast<float2*>(dyall + idx) = make_float2(v0, v1);
                }
            }
        }
        __syncthreads();
        CHAIN_STAMP()
        }
        // ---- dq_att for my units: dq[r][u] = dy[r][:] . W_att[unit u][:]  (DPP row = one (r, u), 16 lanes over the columns)
        {
            const int o = row;                       // 0 .. 31
            float x = 0.f;
            if (o < R * HS) {
                const int r = o / HS, u = o % HS;
                const float* dyr = dyall + r * A;
                const float* wr = wal + u * A;
                if (AL % 4 == 0) {
#pragma unroll
                    for (int c = 0; c < AL / 4; ++c) {
                        const int a0c = c * 64 + kq * 4;
                        const float4 d4 = *reinterpret_cast<const float4*>(dyr + a0c);
                        const float4 w4 = *reinterpret_cast<const float4*>(wr + a0c);
                        x = fmaf(d4.x, w4.x, x); x = fmaf(d4.y, w4.y, x); x = fmaf(d4.z, w4.z, x); x = fmaf(d4.w, w4.w, x);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < AL; ++q) x = fmaf(dyr[kq * AL + q], wr[kq * AL + q], x);
                }
            }
            x = row16_allreduce_sum(x);
            if (kq == 0 && o < R * HS) dql[(o / HS) * HS2 + (o % HS)] = x;
        }
        __syncthreads();
        CHAIN_STAMP()
        // ---- cell pointwise backward (wave 0) -> dG slice
        if (wave0 && cell) {
            float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cb_ok) {
                const size_t rowi = (size_t)i * a.B + cb;
                float* gp = a.gates + rowi * H4 + cj;
                const float* gv = gl + (cr * HS + cuu) * 4;
                const float gi = gv[0], gj = gv[1], gf = gv[2], go = gv[3];
                const float cc = cl[cr * HS + cuu];
                const float cp = cpl[cr * HS + cuu];
                const float dq = dqa[cr * HS + cuu] + dql[cr * HS2 + cuu] + dc;
                const float dh = dhl[cr * HS2 + cuu];
                const float tc = fast_tanh(cc);
                const float dct = dq + dh * go * (1.f - tc * tc);
                dg.x = dct * gj * gi * (1.f - gi);
                dg.y = dct * gi * (1.f - gj * gj);
                dg.z = dct * cp * gf * (1.f - gf);
                dg.w = dh * tc * go * (1.f - go);
                dc = dct * gf;
                // publish dG_i of this unit FIRST: one tagged quad (all-gather; the peers contract it with their rows)
                if (s + 1 < a.T) {
                    uint32_t* dst = reinterpret_cast<uint32_t*>(g4) + (size_t)cr * N4 + 4 * cj;
                    const uint32_t tb = ((((uint32_t)s) >> 1) & 1u) ^ 1u;
                    typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
                    const u32x4s q0 = {(__float_as_uint(dg.x) & ~1u) | tb, (__float_as_uint(dg.y) & ~1u) | tb,
                                       (__float_as_uint(dg.z) & ~1u) | tb, (__float_as_uint(dg.w) & ~1u) | tb};
                    ASR_RACE_HUNT_DELAY();
                    if (fast) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(q0) : "memory");
                    else {
                        __hip_atomic_store(dst + 0, q0.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dst + 1, q0.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dst + 2, q0.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dst + 3, q0.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                gp[0] = dg.x; gp[H] = dg.y; gp[2 * H] = dg.z; gp[3 * H] = dg.w;
            }
        }
        CHAIN_STAMP()
        // (LDS buffers written by wave 1 / the compute phases are rewritten only after later barriers)
    }
    if (STAMP && a.dbg && blockIdx.x == 0 && threadIdx.x == 0) { for (int i = 0; i < 16; ++i) a.dbg[i] = stamp[i]; }
#undef CHAIN_STAMP
    // ---- epilogue: dhf slice and dv partial
    __syncthreads();
    if constexpr (WIDE) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int slot = p * 32 + row, r = slot / MAXTS, tl = slot % MAXTS, tau = tau0 + tl;
            if (rok(r) && tl < TS && tau < Te) {
#pragma unroll
                for (int c = 0; c < AL / 4; ++c)
                    *reinterpret_cast<float4*>(a.dhf + ((size_t)(r0 + r) * Te + tau) * A + c * 64 + kq * 4) =
                        make_float4(greg[p][4 * c], greg[p][4 * c + 1], greg[p][4 * c + 2], greg[p][4 * c + 3]);
            }
        }
    } else {
        for (int idx = tid; idx < R * MAXTS * A; idx += NT) {
            const int r = idx / (MAXTS * A), rem = idx % (MAXTS * A), tl = rem / A, aa = rem % A;
            const int tau = tau0 + tl;
            if (rok(r) && tl < TS && tau < Te) a.dhf[((size_t)(r0 + r) * Te + tau) * A + aa] = dhfl[idx];
        }
    }
#pragma unroll
    for (int q = 0; q < AL; ++q) dyrow[row * A + (AL % 4 == 0 ? (q / 4) * 64 + kq * 4 + (q & 3) : kq * AL + q)] = dvacc[q];
    __syncthreads();
    for (int aa = tid; aa < A; aa += NT) {
        float x = 0.f;
        for (int t = 0; t < 32; ++t) x += dyrow[t * A + aa];
        a.dv_part[((size_t)grp * G + mem) * A + aa] = x;
    }
}

}  // namespace asr

extern "C" int asr_decoder_chain_rows(int Te);

static size_t chain_bwd_ws_bytes_r(int B, int D, int A, int H, int Rr) {
    const size_t R = Rr, groups = ((size_t)B + R - 1) / R, G = 16, AS = A / 16;
    const size_t s2 = R * ((AS + 1) & ~(size_t)1);
    return groups * 2 * (R * ((size_t)D + G) + G * G * s2 + R * (size_t)A + R * 4 * (size_t)H) * sizeof(u64) + groups * 16 * sizeof(u64);
}
extern "C" size_t asr_decoder_chain_bwd_ws_bytes(int B, int D, int A, int H) {      // serves either decomposition
    return std::max(chain_bwd_ws_bytes_r(B, D, A, H, 1), chain_bwd_ws_bytes_r(B, D, A, H, 2));
}

// Dynamic LDS of the backward chain kernel (floats, in carve order) for R utterances per group in `passes` passes over the
// position slots; the hardware limit is 160 KB per workgroup.  *npf_ok: the prefetched items fit the kernel's descriptor slots.
static size_t chain_bwd_lds_bytes_r(int Te, int D, int A, int H, int Rr, int passes, bool* npf_ok) {
    const size_t R = Rr;
    const bool wide = passes == 2;
    const size_t G = 16, HS = H / G, AS = A / G, DS = D / G, HS2 = (HS + 1) & ~(size_t)1, AS2 = (AS + 1) & ~(size_t)1;
    const size_t D1 = D + G, CSB = (size_t)4 * H / 64 + 4, NOUT = HS + DS, TS = ((size_t)Te + G - 1) / G, TeP = ((size_t)Te + 1) & ~(size_t)1;
    const size_t MAXTS = 32 * (size_t)passes / R, NPF = wide ? 6 : 5;
    const size_t uni_a = ((R * D1 + 3) & ~(size_t)3) + R * 64 * CSB + ((NOUT * R * 4 + 3) & ~(size_t)3);
    const size_t uni = uni_a > 32 * (size_t)A ? uni_a : 32 * (size_t)A;
    const size_t nitems = R * A + R * TeP + 2 * R * DS + R * HS * 4 + 3 * R * HS, nitemsP = (nitems + 3) & ~(size_t)3;
    if (npf_ok) *npf_ok = nitems <= NPF * 256 && TS <= MAXTS;
    const size_t floats = 4 + R * HS2 + 4 + R * MAXTS + uni + R * A + R * AS2 + R * A + R * HS2 + (wide ? 0 : 2 * R * MAXTS * A) + HS * A + A +
                          R * TS * D + 2 * nitemsP + 3 * NPF * 256;
    return floats * sizeof(float) + 64;
}
static const size_t kChainBwdLdsMax = 160 * 1024 - 64;
// The backward chain's decomposition for this shape: two utterances per group up to 256 encoder positions (16 per workgroup and
// utterance); beyond, two utterances in two passes with hf / dhf in registers while both utterances' enc rows fit the LDS
// (400 positions at config-2 widths; ASR_CHAIN_BWD_WIDE=0: never), else one utterance per group (up to 512).
static void chain_bwd_mode(int Te, int D, int A, int H, int* R, int* passes) {
    *passes = 1;
    if (Te <= 256) { *R = 2; return; }
    const char* e = getenv("ASR_CHAIN_BWD_WIDE");
    bool ok = false;
    if (!(e && e[0] == '0') && (A / 16) % 4 == 0 && chain_bwd_lds_bytes_r(Te, D, A, H, 2, 2, &ok) <= kChainBwdLdsMax && ok) { *R = 2; *passes = 2; return; }
    *R = 1;
}
// utterances per 16-workgroup group of the backward chain (dv_part holds one partial per workgroup: ceil(B / rows) * 16 rows)
extern "C" int asr_decoder_chain_bwd_rows(int Te, int D, int A, int H) {
    int R, passes;
    chain_bwd_mode(Te, D, A, H, &R, &passes);
    return R;
}
size_t asr_decoder_chain_bwd_lds_bytes(int Te, int D, int A, int H) {
    int R, passes;
    chain_bwd_mode(Te, D, A, H, &R, &passes);
    return chain_bwd_lds_bytes_r(Te, D, A, H, R, passes, nullptr);
}
bool asr_decoder_chain_bwd_fits(int Te, int D, int A, int H) {
    int R, passes;
    chain_bwd_mode(Te, D, A, H, &R, &passes);
    bool ok = false;
    return Te <= 512 && chain_bwd_lds_bytes_r(Te, D, A, H, R, passes, &ok) <= kChainBwdLdsMax && ok;
}

template <int H, int D, int A, int R, int PASSES = 1>
static int chain_bwd_launch(hipStream_t s, asr::ChainBwdArgs& a) {
    constexpr int G = 16;
    const int groups = a.ng;
    const int grid_groups = (((groups + 7) & ~7) * G <= asr::resident_wg_budget()) ? ((groups + 7) & ~7) : groups;
    bool ok = false;
    const size_t lds = chain_bwd_lds_bytes_r(a.Te, D, A, H, R, PASSES, &ok);
    if (lds > kChainBwdLdsMax || !ok) return ASR_EUNSUPPORTED;
    static const bool ared = [] { const char* e = getenv("ASR_CHAIN_BWD_ARED"); return !(e && e[0] == '0'); }();
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asr::decoder_chain_bwd_kernel<H, D, A, R, false, PASSES, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asr::decoder_chain_bwd_kernel<H, D, A, R, false, PASSES, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (H == 256 && R == 2 && PASSES == 1 && a.dbg) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asr::decoder_chain_bwd_kernel<256, 512, 128, 2, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((asr::decoder_chain_bwd_kernel<256, 512, 128, 2, true>), dim3(grid_groups * G), dim3(512), lds, s, a);
        return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
    }
    if (ared) hipLaunchKernelGGL((asr::decoder_chain_bwd_kernel<H, D, A, R, false, PASSES, true>), dim3(grid_groups * G), dim3(512), lds, s, a);
    else hipLaunchKernelGGL((asr::decoder_chain_bwd_kernel<H, D, A, R, false, PASSES, false>), dim3(grid_groups * G), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

// All T steps of the decoder chain backward.  gates holds the activated gates on entry and dG on exit.
int asr_decoder_chain_bwd(void* stream, float* gates, const float* dec_c, const float* alpha, const float* y,
                          const float* ctx, const float* dqc, const float* wh, const float* wc, const float* w_att,
                          const float* v, const float* hf, const float* enc, const int* enc_len, float* dY, float* dctx,
                          float* dhf, float* dv_part, void* ws, int* err, int B, int Te, int D, int A, int H, int T) {
    using namespace asr;
    if (!asr_decoder_chain_supported(B, Te, D, A, H) || T <= 0) return ASR_EUNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t bytes = asr_decoder_chain_bwd_ws_bytes(B, D, A, H);
    if (hipMemsetAsync(ws, 0, bytes, s) != hipSuccess) return ASR_ELAUNCH;
    ChainBwdArgs a;
    a.gates = gates; a.dec_c = dec_c; a.alpha = alpha; a.y = y; a.ctx = ctx; a.dqc = dqc; a.wh = wh; a.wc = wc;
    a.w_att = w_att; a.v = v; a.hf = hf; a.enc = enc; a.enc_len = enc_len; a.dY = dY; a.dctx = dctx; a.dhf = dhf;
    a.dv_part = dv_part; a.gx = static_cast<u64*>(ws);
    int R, passes;
    chain_bwd_mode(Te, D, A, H, &R, &passes);
    const int groups = (B + R - 1) / R;
    a.xcc_slots = reinterpret_cast<u64*>(static_cast<char*>(ws) + chain_bwd_ws_bytes_r(B, D, A, H, R)) - ((size_t)groups * 16);
    a.err = err; a.B = B; a.Te = Te; a.T = T;
    a.dbg = getenv("ASR_CHAIN_STAMP") ? asr::g_lstm_dbg : nullptr;
    const int gpl = std::min(16, asr::resident_wg_budget() / 16);     // 16 groups = 256 workgroups per launch on a whole MI355X
    for (int g0 = 0; g0 < groups; g0 += gpl) {
        a.g0 = g0; a.ng = groups - g0 < gpl ? groups - g0 : gpl;
        int rc;
        if (H == 256) rc = passes == 2 ? chain_bwd_launch<256, 512, 128, 2, 2>(s, a)
                              : (R == 2 ? chain_bwd_launch<256, 512, 128, 2>(s, a) : chain_bwd_launch<256, 512, 128, 1>(s, a));
        else rc = passes == 2 ? ASR_EUNSUPPORTED       // (chain_bwd_mode never picks two passes at these widths: A / 16 = 1 column per lane)
                              : (R == 2 ? chain_bwd_launch<64, 128, 16, 2>(s, a) : chain_bwd_launch<64, 128, 16, 1>(s, a));
        if (rc) return rc;
    }
    return ASR_OK;
}
