// Persistent attention-decoder chain, forward: steps [t0,t1) of the raw_rnn loop
// (attn_decoder.py:76-162) in ONE launch.
//
// What is on the per-step dependency chain of the decoder is short:
//     [h_{i-1}, ctx_{i-1}] -> outer cell (gates = preG_i + h.K_h + ctx.(W2.K_x)) -> q = c_i
//       -> y = q.W_att + b -> e[tau] = v.tanh(hf[tau] + y) -> alpha = softmax -> ctx_i
// everything else is hoisted by the caller (decoder.hip): the LM cell chain (side stream),
// preG_i = (lm_out_i.W1 + b_inp).K_x + b_dec and W2.K_x as MFMA GEMMs before the launch,
// x_i (needed only by the backward), AttnProjection and OutputProjection as GEMMs after it.
// As separate launches those chain stages cost ~27 us per step (start/drain latency of three
// dependent kernels); here a step is four granule exchanges inside a persistent kernel.
//
// Decomposition (same recipe as csrc/lstm.hip): groups of R = 2 utterances never synchronise
// with each other; inside a group G = 16 workgroups each own H/16 hidden units (all 4 gates),
// A/16 attention columns, a slice of ceil(Te/16) encoder positions and D/16 context columns.
// Weights stay on chip for the whole segment: the [(H+D) x 4H/16] slice of [K_h ; W2.K_x] and the
// W_att slice in registers, the hf / enc slices in LDS.  Exchanges are all-gathers of tagged
// 8-byte granules (tag = step+1, one store each, polled with 16-byte sc1 loads), double-buffered;
// wave 0 is the cell/publisher wave (owns every global store, never polls), waves 1-7 poll.
// The same-XCD plain-store fast path is used when the group's XCC ids agree.
#include "common.h"
#include "granule.h"
#include <algorithm>
#include <cstdlib>

namespace asr { extern unsigned long long* g_lstm_dbg; }
extern "C" int asr_decoder_chain_supported(int B, int Te, int D, int A, int H);
extern "C" size_t asr_decoder_chain_ws_bytes(int B, int D, int A, int H);

namespace asr {

struct ChainArgs {
    float* gates;            // [T][B][4H] in: preG rows of the segment; out: activated i,j,f,o
    const float* wh;         // [H][4H]  K_h  (rows E.. of the outer cell's TF kernel)
    const float* wc;         // [D][4H]  W2.K_x
    const float4* wrl;       // [16 members][KC][512 threads] float4: [K_h ; W2.K_x] re-laid in the kernel's register order
    const float* w_att; const float* b_att; const float* v;     // [H][A], [A], [A]
    const float* hf;         // [B][Te][A]
    const float* enc;        // [B][Te][D]
    const int* enc_len;      // [B]
    float* dec_c; float* dec_h; float* alpha; float* ctx; float* y;   // [T][B][.] saved activations
    u64* gx;                 // granules: [groups][2 parities][S | Q | Y | E]
    u64* xcc_slots;          // [groups][16]
    int* err;
    int B, Te, t0, t1;
    unsigned long long* dbg; // STAMP build only
    int g0, ng;              // this launch covers groups [g0, g0 + ng) of the batch (<= 16 groups = 256 workgroups)
};

// H: decoder hidden; D: encoder state width; A: attention width.  R = 2 rows, G = 16 workgroups.
// [K_h ; W2.K_x] -> the forward kernel's register order: out[((mem*KC + i)*512 + tid)] = the 4 gate weights of state row
// k = ((row/16)*16 + kq)*KC + i for unit mem*HS + row%16 (tid = 16*row + kq); zero for padding rows / idle unit slots.
template <int H, int D>
__global__ __launch_bounds__(512) void chain_relayout_kernel(const float* wh, const float* wc, float4* out) {
    constexpr int G = 16, HS = H / G, KS = H + D, KSP = (KS + 127) / 128 * 128, KC = KSP / 32, H4 = 4 * H;
    const int mem = blockIdx.x / KC, i = blockIdx.x % KC, tid = threadIdx.x;
    const int kq = tid & 15, row = tid >> 4, cu = row % 16, cpart_id = row / 16;
    const int k = (cpart_id * 16 + kq) * KC + i;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cu < HS && k < KS) {
        const float* wr = (k < H) ? wh + (size_t)k * H4 : wc + (size_t)(k - H) * H4;
        const int j = mem * HS + cu;
        v = make_float4(wr[j], wr[H + j], wr[2 * H + j], wr[3 * H + j]);
    }
    out[(size_t)blockIdx.x * 512 + tid] = v;
}

// STAMP: diagnostic instantiation (ASR_CHAIN_STAMP=1 + asr_debug_set_buffer): s_memtime totals of wave 0 of workgroup 0
// per phase (code between two consecutive barriers of a step; slot 15 = prologue), accumulated over the launches of a call;
// never used for timing claims.
// R: utterances per group, MAXTS: encoder positions per workgroup (Te <= 16 * MAXTS).  R * MAXTS (utterance, position) pairs are
// scored by the 32 DPP rows of a workgroup in R * MAXTS / 32 passes: (2, 16) for Te <= 256; (1, 32) up to 512 positions; and
// (2, 32), two passes, wherever the LDS still holds two utterances' enc / hf slices (Te <= 432 at config-2 widths: the depth-2
// tap of the phone decoder, 400 positions -- ONE launch for 32 utterances instead of two launches of 16, round 3).
template <int H, int D, int A, int R = 2, bool STAMP = false, int MAXTS_ = 32 / R>
__global__ __launch_bounds__(512) void decoder_chain_fwd_kernel(ChainArgs a) {
    unsigned int stamp[16] = {0};
    unsigned long long tlast = STAMP ? __builtin_amdgcn_s_memtime() : 0;
    int sph = 0;
#define CHAINF_STAMP() if (STAMP) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); stamp[sph & 15] += (unsigned int)(t__ - tlast); tlast = t__; ++sph; }
    constexpr int G = 16, NT = 512;
    static_assert(R == 1 || R == 2, "rows per group");
    constexpr int HS = H / G;            // hidden units per workgroup
    constexpr int AS = A / G;            // attention columns per workgroup
    constexpr int DS = D / G;            // context columns per workgroup
    constexpr int KS = H + D;            // state width [h | ctx]
    constexpr int KSP = (KS + 127) / 128 * 128;   // ... padded so that 32 chunks of a multiple of 4 cover it
    constexpr int KC = KSP / 32;         // state values per lane in the cell matvec (32 chunks)
    constexpr int KCP = KC + 4;          // padded chunk stride in LDS
    constexpr int QP = (H + 127) / 128 * 128;     // padded query length
    constexpr int QC = QP / 32;          // q values per lane in the y matvec
    constexpr int MAXTS = MAXTS_;        // encoder positions per workgroup (Te <= 16 * MAXTS)
    constexpr int PASSES = R * MAXTS / 32;
    static_assert(R * MAXTS == 32 * PASSES && (PASSES == 1 || PASSES == 2) && R * MAXTS <= 64, "(utterance, position) pairs per workgroup");
    static_assert(HS * G == H && AS * G == A && DS * G == D && HS <= 16 && AS <= 8, "sizes");
    static_assert(KC % 4 == 0 && QC % 4 == 0 && KS % 2 == 0 && H % 2 == 0 && A % 2 == 0, "mapping");
    constexpr int H4 = 4 * H;
    constexpr int NTP = (NT / (DS * R)) < 8 ? (NT / (DS * R)) : 8;     // tau parts of the context sum

    extern __shared__ __attribute__((aligned(16))) float smem[];
    // LDS carve (floats)
    int* lds_flag = reinterpret_cast<int*>(smem);   // 4 words reserved (XCC agreement)
    float* sl = smem + 4;                           // state [R][32 chunks][KCP]
    float* sums = sl + R * 32 * KCP;                // [2 parts][HS][R][4]
    float* ql = sums + 2 * HS * R * 4;              // q [R][QP]
    float* ysum = ql + R * QP;                      // [2 parts][AS][R]
    float* yl = ysum + 2 * AS * R + 4;              // y [R][A]
    float* el = yl + R * A;                         // e / alpha [R][G*MAXTS]
    float* eout = el + R * G * MAXTS;               // [R * MAXTS] scores of this workgroup
    float* cpart = eout + 32 * PASSES;              // [8][R][DS]
    float* pgl = cpart + 8 * R * DS;                // preG of the current step for my units [R][HS][4]
    float* vl = pgl + R * HS * 4;                   // v [A]
    float* hfl = vl + A;                            // hf slice [R][TS][A]
    const int Te = a.Te;
    const int TS = (Te + G - 1) / G;
    float* encl = hfl + R * MAXTS * A;              // enc slice [R][Te][DS]

    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane & 15, row = tid >> 4;       // 32 DPP rows
    const int NG = a.ng;
    int grp, mem;
    if (((gridDim.x / G) & 7) == 0) { mem = (blockIdx.x >> 3) % G; grp = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * G)); }   // (grid padded to whole
    else { grp = blockIdx.x / G; mem = blockIdx.x % G; }                                                                      //  octets of groups: lstm.hip)
    if (grp >= NG) return;
    grp += a.g0;
    const int r0 = grp * R;
    const bool wave0 = __builtin_amdgcn_readfirstlane(tid) < 64;
    // wave 7 neither polls nor stores: it fetches the next step's preG one step ahead (its loads never wait
    // behind a publish, and the cell wave never waits on a load)
    const bool wave7 = __builtin_amdgcn_readfirstlane(tid) >= NT - 64;
    constexpr int NPOLL = NT - 128;                 // waves 1..6 poll
    const int brow0 = min(r0, a.B - 1), brow1 = min(r0 + 1, a.B - 1);
    const int blen0 = min(max(a.enc_len[brow0], 0), Te);
    const int blen1 = (r0 + 1 < a.B) ? min(max(a.enc_len[brow1], 0), Te) : 0;
    auto browf = [&](int r) { return r ? brow1 : brow0; };
    auto blenf = [&](int r) { return r ? blen1 : blen0; };
    // granule areas of this group
    constexpr int NS = R * KS, NQ = R * H, NY = R * A, NE = R * G * MAXTS;
    constexpr int NPAR = NS + NQ + NY + NE;
    u64* gbase = a.gx + (size_t)grp * 2 * NPAR;
    const bool fast = group_shares_xcd(a.xcc_slots + (size_t)grp * 16, G, mem, tid, a.err, lds_flag, (uint32_t)a.t0, 1);

    // ---- resident operands -----------------------------------------------------------------
    // cell matvec: DPP row -> (unit u = row % HS..., part): rows [0,16) take chunks 0..15, rows [16,32) chunks 16..31
    // generic in HS: pair index p = row % 16 -> unit up = p % HS (rows beyond HS*... idle when HS < 16)
    const int cu = row % 16, cpart_id = row / 16;            // unit slot (0..15), K part
    const bool cact = cu < HS;
    const int cchunk = cpart_id * 16 + kq;                   // chunk 0..31 of the state vector
    // weights in register order, re-laid once per call by chain_relayout_kernel: one coalesced 16-byte load per (i, thread)
    // (read straight from the TF layout this took 96 strided 4-byte loads per thread: most of a 29 us prologue per launch)
    float wb[KC][4];
    {
        const float4* wp = a.wrl + (size_t)mem * KC * NT + tid;
#pragma unroll
        for (int i = 0; i < KC; ++i) { const float4 v4 = wp[(size_t)i * NT]; wb[i][0] = v4.x; wb[i][1] = v4.y; wb[i][2] = v4.z; wb[i][3] = v4.w; }
    }
    // y matvec: DPP row -> (col ya = row % 8 ..., r, part)
    const int ya = row % 8, yslot = (row / 8) % 2, ypart = row / 16;
    const bool yact = ya < AS && yslot < R;        // (R = 1: the second row slot idles)
    const int yr = yslot < R ? yslot : 0;
    float wy[QC];
    {
        const int acol = mem * AS + (yact ? ya : 0);
#pragma unroll
        for (int i = 0; i < QC; ++i) {
            const int k = (ypart * 16 + kq) * QC + i;
            wy[i] = (yact && k < H) ? a.w_att[(size_t)k * A + acol] : 0.f;
        }
    }
    // hf / enc slices -> LDS (once)
    const int tau0 = mem * TS;
    for (int idx = tid; idx < R * TS * A; idx += NT) {
        const int r = idx / (TS * A), rem = idx % (TS * A), tl = rem / A, aa = rem % A;
        const int tau = tau0 + tl;
        hfl[(r * MAXTS + tl) * A + aa] = (tau < Te) ? a.hf[((size_t)browf(r) * Te + tau) * A + aa] : 0.f;
    }
    for (int idx = tid; idx < R * Te * DS; idx += NT) {
        const int r = idx / (Te * DS), rem = idx % (Te * DS), tau = rem / DS, dd = rem % DS;
        encl[idx] = a.enc[((size_t)browf(r) * Te + tau) * D + mem * DS + dd];
    }
    // cell threads (wave 0): (unit = tid % HS, r = tid / HS) for tid < R*HS
    const bool cell = tid < R * HS;
    const int cr = cell ? tid / HS : 0, cuu = tid % HS;
    const int cj = mem * HS + cuu;
    const int cb = r0 + cr;
    const bool cb_ok = cell && cb < a.B;
    float c_state = 0.f, h_state = 0.f;
    if (cb_ok && a.t0 > 0) {
        c_state = a.dec_c[((size_t)(a.t0 - 1) * a.B + cb) * H + cj];
        h_state = a.dec_h[((size_t)(a.t0 - 1) * a.B + cb) * H + cj];
    }
    // initial state vector [h | ctx] of step t0-1 -> LDS (plain loads: written by earlier launches)
    for (int idx = tid; idx < R * 32 * KCP; idx += NT) sl[idx] = 0.f;
    for (int idx = tid; idx < R * QP; idx += NT) ql[idx] = 0.f;
    __syncthreads();
    for (int idx = tid; idx < R * KS; idx += NT) {
        const int r = idx / KS, k = idx % KS;
        float v = 0.f;
        if (a.t0 > 0 && r0 + r < a.B) {
            const size_t rowi = (size_t)(a.t0 - 1) * a.B + browf(r);
            v = (k < H) ? a.dec_h[rowi * H + k] : a.ctx[rowi * D + (k - H)];
        }
        sl[(r * 32 + k / KC) * KCP + (k % KC)] = v;
    }
    // preG of the next step, in wave 7's registers (software-pipelined): item = lane + 64*j -> (r, unit, gate)
    constexpr int NPG = (R * HS * 4 + 63) / 64;
    float pgr[NPG];
    auto prefetch = [&](int i) {
#pragma unroll
        for (int j = 0; j < NPG; ++j) {
            const int idx = lane + 64 * j, r = idx / (HS * 4), uu = (idx >> 2) % HS, g = idx & 3;
            pgr[j] = (idx < R * HS * 4 && r0 + r < a.B) ? a.gates[((size_t)i * a.B + r0 + r) * H4 + g * H + mem * HS + uu] : 0.f;
        }
    };
#pragma unroll
    for (int j = 0; j < NPG; ++j) pgr[j] = 0.f;
    if (wave7) prefetch(a.t0);
    for (int idx = tid; idx < A; idx += NT) vl[idx] = a.v[idx];
    const float batt = (tid < R * AS) ? a.b_att[mem * AS + tid % AS] : 0.f;
    __syncthreads();

    const int nsteps = a.t1 - a.t0;
    if (STAMP) { sph = 15; CHAINF_STAMP() }
    const int tid_outer = tid;
    for (int s = 0; s < nsteps; ++s) {
        sph = 0;
        // thread indices re-derived per step from an opaque copy (csrc/decoder_greedy.hip): fewer loop-invariant addresses held
        int tz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(tz));
        const int tid = tid_outer + tz, lane = tid & 63;
        const int kq = lane & 15, row = tid >> 4;
        const int i = a.t0 + s;
        const uint32_t ep = (uint32_t)(a.t0 + s + 1);      // tags unique over the segments of one call (workspace zeroed once)
        u64* gpar = gbase + (size_t)(s & 1) * NPAR;
        u64* gS = gpar; u64* gQ = gpar + NS; u64* gY = gQ + NQ; u64* gE = gY + NY;
        if (wave7) {
#pragma unroll
            for (int j = 0; j < NPG; ++j)
                if (lane + 64 * j < R * HS * 4) pgl[lane + 64 * j] = pgr[j];
            if (s + 1 < nsteps) prefetch(i + 1);
        }
        // ---- (1) gather the state [h_{i-1} | ctx_{i-1}] published at the previous step
        if (s > 0) {
            if (!wave0 && !wave7) {
                const u64* src = gbase + (size_t)((s - 1) & 1) * NPAR;
                for (int p = tid - 64; p < NS / 2; p += NPOLL) {
                    const int idx = 2 * p, r = idx / KS, k = idx % KS;
                    float v0 = 0.f, v1 = 0.f;
                    if (r0 + r < a.B) chain_poll2(src + idx, (uint32_t)(a.t0 + s), v0, v1, a.err);
                    *reinterpret_cast<float2*>(sl + (r * 32 + k / KC) * KCP + (k % KC)) = make_float2(v0, v1);
                }
            }
            __syncthreads();
        CHAINF_STAMP()
        }
        CHAINF_STAMP()
        // ---- (2) outer cell: gates = preG + [h|ctx].[K_h ; W2K]  (K split over 32 chunks)
        {
            float acc[R][4];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.f;
                const float4* sp = reinterpret_cast<const float4*>(sl + (r * 32 + cchunk) * KCP);
#pragma unroll
                for (int i4 = 0; i4 < KC / 4; ++i4) {
                    const float4 sv = sp[i4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        acc[r][g] = fmaf(sv.x, wb[4 * i4 + 0][g], acc[r][g]);
                        acc[r][g] = fmaf(sv.y, wb[4 * i4 + 1][g], acc[r][g]);
                        acc[r][g] = fmaf(sv.z, wb[4 * i4 + 2][g], acc[r][g]);
                        acc[r][g] = fmaf(sv.w, wb[4 * i4 + 3][g], acc[r][g]);
                    }
                }
                row16_allreduce_sum4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
            }
            if (kq == 0 && cact) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    *reinterpret_cast<float4*>(sums + ((cpart_id * HS + cu) * R + r) * 4) =
                        make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
            }
        }
        __syncthreads();
        CHAINF_STAMP()
        if (wave0 && cell) {
            const float4 s0 = *reinterpret_cast<const float4*>(sums + ((0 * HS + cuu) * R + cr) * 4);
            const float4 s1 = *reinterpret_cast<const float4*>(sums + ((1 * HS + cuu) * R + cr) * 4);
            const float* pg = pgl + (cr * HS + cuu) * 4;
            const float gi = fast_sigmoid(pg[0] + s0.x + s1.x);
            const float gj = fast_tanh(pg[1] + s0.y + s1.y);
            const float gf = fast_sigmoid(pg[2] + s0.z + s1.z + 1.0f);
            const float go = fast_sigmoid(pg[3] + s0.w + s1.w);
            c_state = c_state * gf + gi * gj;
            h_state = go * fast_tanh(c_state);
            if (cb_ok) {
                chain_publish(gQ + (size_t)cr * H + cj, ep, c_state, fast);          // q = cell state c (decoder.py:79-80)
                chain_publish(gS + (size_t)cr * KS + cj, ep, h_state, fast);         // h part of the next state
                const size_t rowi = (size_t)i * a.B + cb;
                float* gp = a.gates + rowi * H4 + cj;
                gp[0] = gi; gp[H] = gj; gp[2 * H] = gf; gp[3 * H] = go;
                a.dec_c[rowi * H + cj] = c_state;
                a.dec_h[rowi * H + cj] = h_state;
            }
        }
        // ---- (3) gather q, y slice = q.W_att[:, slice] + b
        if (!wave0 && !wave7) {
            for (int p = tid - 64; p < NQ / 2; p += NPOLL) {
                const int idx = 2 * p, r = idx / H;
                float v0 = 0.f, v1 = 0.f;
                if (r0 + r < a.B) chain_poll2(gQ + idx, ep, v0, v1, a.err);
                *reinterpret_cast<float2*>(ql + r * QP + (idx % H)) = make_float2(v0, v1);
            }
        }
        __syncthreads();
        CHAINF_STAMP()
        {
            float acc = 0.f;
            const float4* qp = reinterpret_cast<const float4*>(ql + yr * QP + (ypart * 16 + kq) * QC);
#pragma unroll
            for (int i4 = 0; i4 < QC / 4; ++i4) {
                const float4 qv = qp[i4];
                acc = fmaf(qv.x, wy[4 * i4 + 0], acc); acc = fmaf(qv.y, wy[4 * i4 + 1], acc);
                acc = fmaf(qv.z, wy[4 * i4 + 2], acc); acc = fmaf(qv.w, wy[4 * i4 + 3], acc);
            }
            acc = row16_allreduce_sum(acc);
            if (kq == 0 && yact) ysum[(ypart * AS + ya) * R + yr] = acc;
        }
        __syncthreads();
        CHAINF_STAMP()
        if (wave0 && tid < R * AS) {
            const int r = tid / AS, aa = tid % AS, acol = mem * AS + aa;
            const float yv = batt + ysum[(0 * AS + aa) * R + r] + ysum[(1 * AS + aa) * R + r];
            if (r0 + r < a.B) {
                chain_publish(gY + (size_t)r * A + acol, ep, yv, fast);
                a.y[((size_t)i * a.B + r0 + r) * A + acol] = yv;
            }
        }
        // ---- (4) gather y, scores on this workgroup's position slice
        if (!wave0 && !wave7) {
            for (int p = tid - 64; p < NY / 2; p += NPOLL) {
                const int idx = 2 * p, r = idx / A;
                float v0 = 0.f, v1 = 0.f;
                if (r0 + r < a.B) chain_poll2(gY + idx, ep, v0, v1, a.err);
                *reinterpret_cast<float2*>(yl + idx) = make_float2(v0, v1);
            }
        }
        __syncthreads();
        CHAINF_STAMP()
        {
            // DPP row (+ 32 per pass) -> (tl = pair % MAXTS, r = pair / MAXTS); lane kq -> A/16 consecutive a (float4 steps)
#pragma unroll
            for (int pass = 0; pass < PASSES; ++pass) {
                const int pair = row + 32 * pass;
                const int tl = pair % MAXTS, r = pair / MAXTS;
                float sc = 0.f;
                if (tl < TS) {
                    constexpr int AL = A / 16;
                    const float* hrow = hfl + (r * MAXTS + tl) * A;
                    const float* yrow = yl + r * A;
                    if (AL % 4 == 0) {       // float4 chunks 64 columns apart: conflict-free across the 16 lanes of the DPP row
#pragma unroll
                        for (int c = 0; c < AL / 4; ++c) {
                            const int a0 = c * 64 + kq * 4;
                            const float4 h4 = *reinterpret_cast<const float4*>(hrow + a0);
                            const float4 y4 = *reinterpret_cast<const float4*>(yrow + a0);
                            const float4 v4 = *reinterpret_cast<const float4*>(vl + a0);
                            sc = fmaf(v4.x, fast_tanh(h4.x + y4.x), sc); sc = fmaf(v4.y, fast_tanh(h4.y + y4.y), sc);
                            sc = fmaf(v4.z, fast_tanh(h4.z + y4.z), sc); sc = fmaf(v4.w, fast_tanh(h4.w + y4.w), sc);
                        }
                    } else {
#pragma unroll
                        for (int q4 = 0; q4 < AL; ++q4) sc = fmaf(vl[kq * AL + q4], fast_tanh(hrow[kq * AL + q4] + yrow[kq * AL + q4]), sc);
                    }
                }
                sc = row16_allreduce_sum(sc);
                if (kq == 0) eout[pair] = sc;
            }
        }
        __syncthreads();
        CHAINF_STAMP()
        if (wave0 && tid < R * MAXTS) {
            const int tl = tid % MAXTS, r = tid / MAXTS;
            if (tl < TS && r0 + r < a.B)
                chain_publish(gE + (size_t)r * G * MAXTS + mem * MAXTS + tl, ep, eout[tid], fast);
        }
        // ---- (5) gather all scores, softmax over tau < len (replicated), context slice
        if (!wave0 && !wave7) {
            const int pairs = (TS + 1) / 2;                       // per (r, source workgroup)
            for (int p = tid - 64; p < R * G * pairs; p += NPOLL) {
                const int r = p / (G * pairs), rem = p % (G * pairs), m = rem / pairs, tp = rem % pairs;
                const int off = r * G * MAXTS + m * MAXTS + 2 * tp;
                float v0 = 0.f, v1 = 0.f;
                if (r0 + r < a.B) {
                    if (2 * tp + 1 < TS) chain_poll2(gE + off, ep, v0, v1, a.err);
                    else {   // odd tail: a single granule
                        long long t0w = 0;
                        ASR_RACE_HUNT_DELAY();
                        for (uint32_t spins = 0;; ++spins) {
                            const u64 x = __hip_atomic_load(gE + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((uint32_t)(x >> 32) == ep) { v0 = __uint_as_float((uint32_t)x); break; }
                            ASR_POLL_BACKOFF();
        if ((spins & 1023) == 1023) {
                                const long long now = wall_clock64();
                                if (t0w == 0) t0w = now; else if (now - t0w > 200000000LL) { *a.err = 51; break; }
                            }
                        }
                    }
                }
                el[off] = v0;
                if (2 * tp + 1 < MAXTS) el[off + 1] = v1;
            }
        }
        __syncthreads();
        CHAINF_STAMP()
        if (wave < R) {      // wave r: softmax of row r over tau < len; slots (m, tl) <-> tau = m*TS + tl, no division
            const int r = wave, L = blenf(r);
            float* er = el + r * G * MAXTS;
            float ev[G * MAXTS / 64];
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < G * MAXTS / 64; ++j) {
                const int sl = lane + 64 * j, tl = sl % MAXTS, tau = (sl / MAXTS) * TS + tl;
                const bool ok = tl < TS && tau < L;
                ev[j] = ok ? er[sl] : -INFINITY;
                m = fmaxf(m, ev[j]);
            }
            m = wave_allreduce_max(m);
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < G * MAXTS / 64; ++j) { ev[j] = ev[j] > -INFINITY ? __expf(ev[j] - m) : 0.f; sum += ev[j]; }
            sum = wave_allreduce_sum(sum);
            const float inv = L > 0 ? 1.0f / sum : 0.f;
#pragma unroll
            for (int j = 0; j < G * MAXTS / 64; ++j) er[lane + 64 * j] = ev[j] * inv;      // zero past the length and in unused slots
        }
        __syncthreads();
        CHAINF_STAMP()
        {
            const int dd = tid % DS, r = (tid / DS) % R, tp = tid / (DS * R);
            float cs = 0.f;
            if (tp < NTP) {
                const int L = blenf(r);
                float c1 = 0.f, c2 = 0.f, c3 = 0.f;
                for (int m = tp; m < G; m += NTP) {
                    const float* ap = el + r * G * MAXTS + m * MAXTS;
                    const float* xp = encl + ((size_t)r * Te + m * TS) * DS + dd;
                    const int nt = min(TS, L - m * TS);
                    // four independent chains, loads ahead of the FMAs (one serial "load, load, fma" chain over up to 32 positions before)
                    int tl = 0;
                    for (; tl + 4 <= nt; tl += 4) {
                        const float4 a4 = *reinterpret_cast<const float4*>(ap + tl);
                        const float x0 = xp[tl * DS], x1 = xp[(tl + 1) * DS], x2 = xp[(tl + 2) * DS], x3 = xp[(tl + 3) * DS];
                        cs = fmaf(a4.x, x0, cs); c1 = fmaf(a4.y, x1, c1); c2 = fmaf(a4.z, x2, c2); c3 = fmaf(a4.w, x3, c3);
                    }
                    for (; tl < nt; ++tl) cs = fmaf(ap[tl], xp[tl * DS], cs);
                }
                cpart[(tp * R + r) * DS + dd] = (cs + c1) + (c2 + c3);
            }
        }
        __syncthreads();
        CHAINF_STAMP()
        if (wave0 && tid < R * DS) {
            const int r = tid / DS, dd = tid % DS;
            float cs = 0.f;
#pragma unroll
            for (int tp = 0; tp < NTP; ++tp) cs += cpart[(tp * R + r) * DS + dd];
            if (r0 + r < a.B) {
                if (s + 1 < nsteps) chain_publish(gS + (size_t)r * KS + H + mem * DS + dd, ep, cs, fast);
                a.ctx[((size_t)i * a.B + r0 + r) * D + mem * DS + dd] = cs;
            }
        }
        if (wave0 && tid < R * MAXTS) {       // alpha of this step -> global: every workgroup stores its own position slice
            const int r = tid / MAXTS, tl = tid % MAXTS, tau = tau0 + tl;
            if (tl < TS && tau < Te && r0 + r < a.B)
                a.alpha[((size_t)i * a.B + r0 + r) * Te + tau] = el[r * G * MAXTS + mem * MAXTS + tl];
        }
        CHAINF_STAMP()
        // (LDS buffers are rewritten only after later barriers of the next step)
    }
    if (STAMP && a.dbg && blockIdx.x == 0 && threadIdx.x == 0) { for (int i = 0; i < 16; ++i) atomicAdd(a.dbg + 16 + i, (unsigned long long)stamp[i]); }
#undef CHAINF_STAMP
}

}  // namespace asr

extern "C" int asr_decoder_chain_rows(int Te) { return Te <= 256 ? 2 : 1; }

extern "C" int asr_decoder_chain_supported(int B, int Te, int D, int A, int H) {
    if (getenv("ASR_DEC_CHAIN") && atoi(getenv("ASR_DEC_CHAIN")) == 0) return 0;
    if (Te > 512 || Te <= 0 || B <= 0) return 0;
    if (asr::resident_wg_budget() < 16) return 0;          // one 16-workgroup group must be co-resident
    return (H == 256 && D == 512 && A == 128) || (H == 64 && D == 128 && A == 16);
}

// LDS of one workgroup of the forward kernel (floats), for R utterances per group and MAXTS positions per workgroup
static size_t chain_fwd_lds_floats(int H, int D, int A, int R, int MAXTS, int Te) {
    const int G = 16, KSP = (H + D + 127) / 128 * 128, KCP = KSP / 32 + 4, QP = (H + 127) / 128 * 128;
    return 4 + (size_t)R * 32 * KCP + 2 * (H / G) * R * 4 + R * QP + 2 * (A / G) * R + 4 + R * A + (size_t)R * G * MAXTS + R * MAXTS +
           8 * R * (D / G) + R * (H / G) * 4 + A + (size_t)R * MAXTS * A + (size_t)R * Te * (D / G);
}
static const size_t kChainLdsMax = 160 * 1024 - 64;
// The forward chain's decomposition for this Te: two utterances per group wherever their slices fit the LDS (16 positions per
// workgroup up to 256 positions, 32 -- two score passes -- beyond), else one.  (asr_decoder_chain_rows is the BACKWARD
// chain's: its LDS also holds dhf, so it drops to one utterance per group above 256 positions.)
static void chain_fwd_mode(int Te, int H, int D, int A, int* R, int* MAXTS) {
    if (Te <= 256) { *R = 2; *MAXTS = 16; return; }
    const char* e = getenv("ASR_CHAIN_FWD_R2");          // =0: one utterance per group above 256 positions, as before round 3
    const bool two = !(e && e[0] == '0');
    if (two && chain_fwd_lds_floats(H, D, A, 2, 32, Te) * sizeof(float) <= kChainLdsMax) { *R = 2; *MAXTS = 32; return; }
    *R = 1; *MAXTS = 32;
}
// granule area of one call: [groups][2 parities][S | Q | Y | E] + [groups][16] XCC slots, for R rows per group
static size_t chain_npar(int R, int MAXTS, int D, int A, int H) { return (size_t)R * (H + D) + (size_t)R * H + (size_t)R * A + (size_t)R * 16 * MAXTS; }
static size_t chain_gran_bytes_r(int B, int D, int A, int H, int R, int MAXTS) {
    const size_t groups = ((size_t)B + R - 1) / R;
    return (groups * 2 * chain_npar(R, MAXTS, D, A, H) * sizeof(u64) + groups * 16 * sizeof(u64) + 255) / 256 * 256;
}
static size_t chain_gran_bytes(int B, int D, int A, int H) {      // the workspace serves every decomposition
    return std::max(std::max(chain_gran_bytes_r(B, D, A, H, 1, 32), chain_gran_bytes_r(B, D, A, H, 2, 16)), chain_gran_bytes_r(B, D, A, H, 2, 32));
}
static size_t chain_relayout_bytes(int D, int H) {
    const size_t KC = ((size_t)(H + D) + 127) / 128 * 128 / 32;
    return 16 * KC * 512 * sizeof(float4);
}
extern "C" size_t asr_decoder_chain_ws_bytes(int B, int D, int A, int H) {
    return chain_gran_bytes(B, D, A, H) + chain_relayout_bytes(D, H);      // granules + XCC slots | re-laid weights
}

template <int H, int D, int A, int R, int MAXTS>
static int chain_launch(hipStream_t s, asr::ChainArgs& a, int Te) {
    constexpr int G = 16;
    const int groups = a.ng;
    const int grid_groups = (((groups + 7) & ~7) * G <= asr::resident_wg_budget()) ? ((groups + 7) & ~7) : groups;
    const size_t lds = sizeof(float) * chain_fwd_lds_floats(H, D, A, R, MAXTS, Te);
    if (lds > kChainLdsMax) return ASR_EUNSUPPORTED;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asr::decoder_chain_fwd_kernel<H, D, A, R, false, MAXTS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (H == 256 && R == 2 && MAXTS == 16 && a.dbg) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asr::decoder_chain_fwd_kernel<256, 512, 128, 2, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((asr::decoder_chain_fwd_kernel<256, 512, 128, 2, true>), dim3(grid_groups * G), dim3(512), lds, s, a);
        return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
    }
    hipLaunchKernelGGL((asr::decoder_chain_fwd_kernel<H, D, A, R, false, MAXTS>), dim3(grid_groups * G), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

// Steps [t0,t1) of the decoder chain; 16 groups per launch (32 utterances at R = 2, 16 at R = 1), larger batches run as
// consecutive launches over group ranges (each range has its own granule area, so no re-zeroing in between).
// gates holds preG for those steps on entry.  ws: asr_decoder_chain_ws_bytes().
int asr_decoder_chain_fwd(void* stream, float* gates, const float* wh, const float* wc, const float* w_att,
                          const float* b_att, const float* v, const float* hf, const float* enc, const int* enc_len,
                          float* dec_c, float* dec_h, float* alpha, float* ctx, float* y, void* ws, int* err,
                          int B, int Te, int D, int A, int H, int t0, int t1) {
    using namespace asr;
    if (!asr_decoder_chain_supported(B, Te, D, A, H) || t1 <= t0) return ASR_EUNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // NOTE: all batch rows of a launch share the [T][B][.] row stride B, so chunking is by group range only
    int R, MAXTS;
    chain_fwd_mode(Te, H, D, A, &R, &MAXTS);
    const size_t bytes = chain_gran_bytes(B, D, A, H);
    float4* wrl = reinterpret_cast<float4*>(static_cast<char*>(ws) + bytes);
    if (t0 == 0) {                                                                          // once per sequence
        if (hipMemsetAsync(ws, 0, bytes, s) != hipSuccess) return ASR_ELAUNCH;
        const int KC = ((H + D + 127) / 128 * 128) / 32;
        if (H == 256) hipLaunchKernelGGL((asr::chain_relayout_kernel<256, 512>), dim3(16 * KC), dim3(512), 0, s, wh, wc, wrl);
        else hipLaunchKernelGGL((asr::chain_relayout_kernel<64, 128>), dim3(16 * KC), dim3(512), 0, s, wh, wc, wrl);
    }
    ChainArgs a;
    a.gates = gates; a.wh = wh; a.wc = wc; a.w_att = w_att; a.b_att = b_att; a.v = v; a.hf = hf; a.enc = enc;
    a.enc_len = enc_len; a.dec_c = dec_c; a.dec_h = dec_h; a.alpha = alpha; a.ctx = ctx; a.y = y;
    a.gx = static_cast<u64*>(ws);
    a.wrl = wrl;
    const size_t groups = ((size_t)B + R - 1) / R;
    a.xcc_slots = a.gx + groups * 2 * chain_npar(R, MAXTS, D, A, H);
    a.err = err; a.B = B; a.Te = Te; a.t0 = t0; a.t1 = t1;
    a.dbg = getenv("ASR_CHAIN_STAMP") ? asr::g_lstm_dbg : nullptr;
    const int gpl = std::min(16, asr::resident_wg_budget() / 16);     // groups per launch: all of them co-resident
    for (int g0 = 0; g0 < (int)groups; g0 += gpl) {
        a.g0 = g0; a.ng = std::min<int>(gpl, (int)groups - g0);
        int rc;
        if (H == 256) rc = R == 1 ? chain_launch<256, 512, 128, 1, 32>(s, a, Te) : MAXTS == 16 ? chain_launch<256, 512, 128, 2, 16>(s, a, Te) : chain_launch<256, 512, 128, 2, 32>(s, a, Te);
        else rc = R == 1 ? chain_launch<64, 128, 16, 1, 32>(s, a, Te) : MAXTS == 16 ? chain_launch<64, 128, 16, 2, 16>(s, a, Te) : chain_launch<64, 128, 16, 2, 32>(s, a, Te);
        if (rc) return rc;
    }
    return ASR_OK;
}
