// Persistent (Bi)LSTM layer, backward through time -- the gradient of csrc/lstm.hip.
//
// Reference: tf.gradients through bidirectional_dynamic_rnn/BasicLSTMCell
// (seq2seq_model.py:148 over encoder.py:55-91).  With saved activations i,j,f,o (sigmoid /
// tanh already applied), c_t, and c_{t-1}:
//   tc = tanh(c_t); do = dh*tc; dc_tot = dc_carry + dh*o*(1-tc^2)
//   di = dc_tot*j; dj = dc_tot*i; df = dc_tot*c_{t-1}; dc_carry' = dc_tot*f
//   dG = [di*i(1-i), dj*(1-j^2), df*f(1-f), do*o(1-o)]           (pre-activation grads)
//   dh_{t-1} (recurrent part) = dG . K_h^T
// dG overwrites the saved gates in place ([B,T,ND,4H]); the input/weight gradients are then
// three MFMA GEMMs per direction (dX = dG.K_x^T, dK_x = X^T.dG, dK_h = Hprev^T.dG).
//
// Same decomposition as the forward: groups of R batch rows x G = H/32 workgroups, each
// owning 32 hidden units (all 4 gates) with its K_h column slice in registers.  A workgroup
// can only form the PARTIAL of dh_{t-1} over its own 128 gate columns, for all H units, so
// the per-step exchange is a reduce-scatter instead of an all-gather: each workgroup
// publishes R*H partial values as {tag, value} granules addressed to the owner of each unit
// and every cell thread sums the G partials of its (row, unit) in fixed order (bitwise
// reproducible).  Same granule count per step as the forward (R*H).
// Thread map: lane = 16*kgl + nc; the DPP row's 16 lanes hold the 16 column chunks (8 gate
// columns = 2 units x 4 gates) and each lane KG = H/32 output rows k, so the reduction over
// columns is again 4 DPP butterflies.
#include "common.h"
#include <hip/hip_ext.h>
#include "p3.h"
#include "granule.h"
#include <cstdlib>
namespace asr { extern unsigned long long* g_lstm_dbg; }

namespace asr {

struct LstmBwdArgs {
    float* __restrict__ gates;         // [B][T][ND][4H]  out: dG (zero past len)
    const float* __restrict__ act;     // [B][T][ND][H][8] forward records {i,j,f,o | c, c_prev, -, -}
    const float* __restrict__ dout;    // [B][Tout][ND*H] gradient w.r.t. the layer output
    const float* kh[2];
    const int* len;
    u64* hx;               // [groups][2][G dst][G src][R][32] granules
    u64* xcc_slots;        // [groups][16] XCC-ID agreement slots (zeroed with hx)
    int* err;
    int B, T, Tout, ND, boff;
    float keep; uint32_t seed;
    // addressing as in LstmRecArgs (csrc/lstm.hip): row of (b,t) in gates/act = b*sb + t*st; in dout =
    // b*osb + t*ost with leading dimension ldo; dropout counter = (boff+b)*dsb + t*dst
    int sb, st, osb, ost, ldo, dsb, dst;
    unsigned long long* dbg;   // STAMP build only
    float* db_part;            // [B][ND][4H] per-utterance sums of dG over time (bias gradient partials) or nullptr
    // lstm_rec_bwd4_kernel: dG ALSO (dg_f32 = 0: ONLY) as bf16 planes for the GEMMs of csrc/gemm_p3.hip, or nullptr: the P3 image
    // of dG [B*T][ND*4H] with the columns of a direction UNIT-major (column dir*4H + 4*unit + gate: a lane's four gates are 8
    // contiguous bytes of a plane; the consumers permute their weights / output columns instead)
    char* dg_p3; int p3_np; int dg_f32;
    const float* act_c;        // lstm_rec_bwd4_kernel: c plane of the 20-byte split records (csrc/lstm.hip LstmRecArgs::act_c)
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// Two adjacent granules with one 16-byte sc1 load (each 8-byte half is written by one store).
__device__ __forceinline__ bool poll_pair(const u64* g, uint32_t epoch, float& v0, float& v1, int* err) {
    long long t0 = 0;
    const u32x4* p = reinterpret_cast<const u32x4*>(g);
    ASR_RACE_HUNT_DELAY();
    for (uint32_t spins = 0;; ++spins) {
        u32x4 x;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(p) : "memory");
        if (x.y == epoch && x.w == epoch) { v0 = __uint_as_float(x.x); v1 = __uint_as_float(x.z); return true; }
        ASR_POLL_BACKOFF();
        if ((spins & 1023) == 1023) {
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > 200000000LL) { *err = 13; v0 = v1 = 0.f; return false; }
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { v0 = v1 = 0.f; return false; }
        }
    }
}

// Roles (wave-uniform, via readfirstlane => scalar branches): the first NCW threads are the cell
// waves -- they own the pointwise backward and every bookkeeping store, and never poll; the
// remaining waves gather the G partials of every (row, unit) with ONE 16-byte load per thread and
// hand them over through LDS, where the cell threads sum them in fixed order (reproducible).
template <int H, int R>
__global__ __launch_bounds__(512) void lstm_rec_bwd_kernel(LstmBwdArgs a) {
    constexpr int HS = 32, NT = 512;
    constexpr int G = H / HS;          // workgroups per group (<= 16)
    constexpr int KG = H / 32;         // output rows k per lane
    constexpr int CS = 12;             // padded LDS chunk stride (8 values + 4)
    constexpr int NCELL = R * HS;
    constexpr int NCW = (NCELL + 63) / 64 * 64;
    constexpr int NPOLL = NT - NCW;
    static_assert(NCW < NT, "need at least one polling wave");
    __shared__ __attribute__((aligned(16))) float dgl[R * 16 * CS];
    __shared__ __attribute__((aligned(16))) float part[G * R * HS];
    __shared__ __attribute__((aligned(16))) float pub[R * H];

    // latency-critical serial chain: win issue arbitration against co-resident throughput kernels
    // (the weight-gradient GEMMs of the layer above run concurrently on the side stream)
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nc = lane & 15, kg = wave * 4 + (lane >> 4);
    const int NG = (a.B + R - 1) / R;
    const int ngroups = a.ND * NG;
    int grp, mem;
    if (((gridDim.x / G) & 7) == 0) { mem = (blockIdx.x >> 3) % G; grp = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * G)); }   // (grid padded to whole
    else { grp = blockIdx.x / G; mem = blockIdx.x % G; }                                                                      //  octets of groups: lstm.hip)
    if (grp >= ngroups) return;
    const int dir = grp / NG, bg = grp % NG;
    const int r0 = bg * R;
    const int H4 = 4 * H;
    const int j0 = mem * HS;

    // K_h slice -> registers: w[i][jn] = K_h[kg*KG + i][col(8*nc + jn)], local column n = 4*unit + gate
    float w[KG][8];
    {
        const float* kh = a.kh[dir];
#pragma unroll
        for (int i = 0; i < KG; ++i)
#pragma unroll
            for (int jn = 0; jn < 8; ++jn) {
                const int n = 8 * nc + jn;
                w[i][jn] = kh[(size_t)(kg * KG + i) * H4 + (n & 3) * H + j0 + (n >> 2)];
            }
    }
    int S = 0;
    for (int r = 0; r < R; ++r) S = max(S, (r0 + r < a.B) ? min(a.len[r0 + r], a.T) : 0);
    const bool cell = tid < NCELL;
    const bool cell_wave = __builtin_amdgcn_readfirstlane(tid) < NCW;
    const int cr = min(tid / HS, R - 1), cu = tid % HS;
    const int cb = r0 + cr;
    const int cb_safe = min(cb, a.B - 1);
    const int clen = (cell && cb < a.B) ? min(a.len[cb], a.T) : 0;
    const int cj = j0 + cu;
    float dc = 0.f;
    u64* hxg = a.hx + (size_t)grp * 2 * G * G * R * HS;
    const bool fast = group_shares_xcd(a.xcc_slots + (size_t)grp * 16, G, mem, tid, a.err, nullptr, 0, 5);

    // software-pipelined operands of the cell (loop-carried registers, no in-loop init)
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
    float dout_v = 0.f;
    auto prefetch = [&](int s) {
        const int t = dir ? s : (clen - 1 - s);
        const int ts = min(max(t, 0), a.T - 1);
        const float4* rp = reinterpret_cast<const float4*>(a.act + ((((size_t)cb_safe * a.sb + (size_t)ts * a.st) * a.ND + dir) * H + cj) * 8);
        ra = rp[0]; rb = rp[1];
        dout_v = a.dout[((size_t)cb_safe * a.osb + (size_t)ts * a.ost) * a.ldo + dir * H + cj];
    };
    if (cell_wave) prefetch(0);

    for (int s = 0; s < S; ++s) {
        const bool live = cell && s < clen;
        const int t = dir ? s : (clen - 1 - s);          // reverse of the forward walk
        if (s > 0) {
            if (!cell_wave) {
                // partials addressed to this workgroup: [src m][r][u], pairs over u
                const u64* src = hxg + ((size_t)((s - 1) & 1) * G + mem) * G * R * HS;
                for (int pidx = tid - NCW; pidx < G * R * HS / 2; pidx += NPOLL) {
                    const int idx = 2 * pidx;
                    const int r = (idx / HS) % R;
                    float v0 = 0.f, v1 = 0.f;
                    if (r0 + r < a.B) poll_pair(src + idx, (uint32_t)s, v0, v1, a.err);
                    *reinterpret_cast<float2*>(part + idx) = make_float2(v0, v1);
                }
            }
            __syncthreads();
        }
        float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cell_wave && cell) {
            float dh = dout_v;
            if (a.keep < 1.0f)
                dh *= keep_scale(a.seed, (uint32_t)((a.boff + cb) * a.dsb + t * a.dst), (uint32_t)(dir * H + cj), a.keep);
            if (s > 0) {
                float rec = 0.f;
#pragma unroll
                for (int m = 0; m < G; ++m) rec += part[(m * R + cr) * HS + cu];
                dh += rec;
            }
            if (live) {
                const float gi = ra.x, gj = ra.y, gf = ra.z, go = ra.w, cc = rb.x, cp = rb.y;
                const float tc = fast_tanh(cc);
                const float dct = dc + dh * go * (1.f - tc * tc);
                dg.x = dct * gj * gi * (1.f - gi);
                dg.y = dct * gi * (1.f - gj * gj);
                dg.z = dct * cp * gf * (1.f - gf);
                dg.w = dh * tc * go * (1.f - go);
                dc = dct * gf;
            }
            // local column n = 4*cu + gate -> chunk cu/2, offset 4*(cu&1)
            *reinterpret_cast<float4*>(dgl + (cr * 16 + (cu >> 1)) * CS + 4 * (cu & 1)) = dg;
        }
        __syncthreads();
        if (s + 1 < S) {
            // partial dh_{prev}[r][k] over this workgroup's 128 gate columns
            float acc[R][KG];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float4 d0 = *reinterpret_cast<const float4*>(dgl + (r * 16 + nc) * CS);
                const float4 d1 = *reinterpret_cast<const float4*>(dgl + (r * 16 + nc) * CS + 4);
#pragma unroll
                for (int i = 0; i < KG; ++i) {
                    float x = d0.x * w[i][0];
                    x = fmaf(d0.y, w[i][1], x); x = fmaf(d0.z, w[i][2], x); x = fmaf(d0.w, w[i][3], x);
                    x = fmaf(d1.x, w[i][4], x); x = fmaf(d1.y, w[i][5], x);
                    x = fmaf(d1.z, w[i][6], x); x = fmaf(d1.w, w[i][7], x);
                    acc[r][i] = row16_allreduce_sum(x);
                }
            }
            // hand the R*H partial sums to the cell waves through LDS, laid out like the granule
            // block of each destination ([dst m'][r][u']); only the cell waves (which never poll)
            // issue stores, so a polling wave's vmcnt(0) never waits on a write-through ack
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int i = 0; i < KG; ++i) {
                    if (((r * KG + i) & 15) == nc) {
                        const int k = kg * KG + i;
                        pub[((k / HS) * R + r) * HS + (k % HS)] = acc[r][i];
                    }
                }
        }
        __syncthreads();
        if (cell_wave && s + 1 < S) {
            u64* dstb = hxg + (size_t)(s & 1) * G * G * R * HS;
            for (int idx = tid; idx < R * H; idx += NCW) {
                const int md = idx / (R * HS), rem = idx % (R * HS);
                if (r0 + rem / HS < a.B) {
                    u64* dst = dstb + ((size_t)md * G + mem) * R * HS + rem;
                    const u64 gv = ((u64)(uint32_t)(s + 1) << 32) | __float_as_uint(pub[idx]);
                    ASR_RACE_HUNT_DELAY();
                    if (fast) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst), "v"(gv) : "memory");
                    else __hip_atomic_store(dst, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (cell_wave && cell) {       // bookkeeping, off the critical path
            if (live) {
                float* gp = a.gates + (((size_t)cb * a.sb + (size_t)t * a.st) * a.ND + dir) * H4 + cj;
                gp[0] = dg.x; gp[H] = dg.y; gp[2 * H] = dg.z; gp[3 * H] = dg.w;
            }
            if (s + 1 < S) prefetch(s + 1);
        }
        // (dgl / part are rewritten only after the next step's first barrier, which every wave
        //  reaches after finishing this step's reads)
    }
    // dG = 0 past each row's length (the weight/input GEMMs read every row)
    for (int r = 0; r < R; ++r) {
        if (r0 + r >= a.B) break;
        const int l = min(a.len[r0 + r], a.T);
        const int nz = a.T - l;
        for (int idx = tid; idx < nz * 4 * HS; idx += NT) {
            const int t = l + idx / (4 * HS), q = idx % (4 * HS);
            a.gates[(((size_t)(r0 + r) * a.sb + (size_t)t * a.st) * a.ND + dir) * H4 + (q / HS) * H + j0 + (q % HS)] = 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// All-gather formulation (default): exchange dG itself instead of partial sums of dh.
//   step s:  gather dG of step s-1 (R x 4H values, published by their owners) -> LDS
//            dh_rec[k] = sum_n dG[n] K_h[k][n] for this workgroup's OWN 32 units k, contraction over ALL 4H
//            columns (K_h rows of the own units in registers: 4 units x 4H/64 columns per lane)
//            pointwise backward for the own units -> dG_s (4 values per unit), published first
// Same structure as the forward kernel (one exchange, two barriers per step); the reduce-scatter above needs
// a third barrier and 4x the publishing stores.  Exchange volume per step: R*4H granules per group (4x the
// forward's), polled by every workgroup of the group with all of a thread's loads in flight.
// STAMP: diagnostic instantiation (ASR_LSTM_STAMP=1 + asr_debug_set_buffer): s_memtime totals of thread 0 (a cell wave) and
// thread 511 (a polling wave) of workgroup 0: [poll | barrier 1 | matvec | barrier 2 | cell]; never used for timing claims.
// MF (bf16 mode of the library, BASELINE config 3): the contraction dG_{s-1} . K_h^T for the own 32 units runs on the bf16
// matrix pipe -- K_h rows rounded to bf16 once per launch into MFMA A-fragments, dG rounded to bf16 by the pollers on its way
// into LDS, fp32 accumulation: 2 unit tiles x 4 position ranges = one (tile, range) per wave, 8 v_mfma_f32_16x16x32_bf16 per
// step; the four range partials meet in `sums` exactly where the VALU path's four DPP-row partials do.
typedef __bf16 qbf16x8 __attribute__((ext_vector_type(8)));
template <int H, int R, bool STAMP = false, bool MF = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 8))) void lstm_rec_bwd_ag_kernel(LstmBwdArgs a) {
    unsigned int stamp[5] = {0, 0, 0, 0, 0};
    unsigned long long tlast = STAMP ? __builtin_amdgcn_s_memtime() : 0;
#define BPTT_STAMP(i) if (STAMP) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); stamp[i] += (unsigned int)(t__ - tlast); tlast = t__; }
    constexpr int HS = 32, NT = 512;
    constexpr int G = H / HS;
    constexpr int N = 4 * H;           // dG columns per row; position p = 4*unit + gate
    constexpr int PC = N / 64;         // positions per lane in the matvec (64 chunks = one wave)
    constexpr int CSB = PC + 4;        // padded LDS chunk stride
    constexpr int NCELL = R * HS;
    constexpr int NCW = (NCELL + 63) / 64 * 64;
    constexpr int NPOLL = NT - NCW - 64;           // the last wave is the LOADER: it never polls
    // Exchange format: one 16-byte quad per unit = its four dG values, each carrying a 1-bit tag in the LOWEST MANTISSA BIT
    // (the value is truncated to 23 mantissa bits; every workgroup, the owner included, consumes the truncated value).  A slot of
    // parity buffer s & 1 is rewritten every second step, so one bit -- ((s >> 1) & 1) ^ 1, i.e. 1 for the first write after
    // the host's memset to zero -- tells this step's write from the previous one; a torn read shows mixed bits and is retried.
    // Half the bytes and half the loads of {tag32, value32} granules: 32 workgroups per XCD each read every quad of their group.
    constexpr int NQUAD = R * N / 4;
    constexpr int NPP = (NQUAD + NPOLL - 1) / NPOLL;   // quads per polling thread
    static_assert(NPP >= 1 && NPP <= 3, "quads per polling thread");
    static_assert(NCW + 64 < NT && PC % 4 == 0 && NCELL <= 64, "mapping");
    __shared__ __attribute__((aligned(16))) float dgl[MF ? 4 : R * 64 * CSB];
    __shared__ __attribute__((aligned(16))) float sums[R * HS * 4];
    __shared__ __attribute__((aligned(16))) unsigned short dgb[MF ? (R + 1) * N : 8];     // MF: dG as bf16, row R = zeros
    static_assert(!MF || (H == 256 && R <= 16), "MF: 2 unit tiles x 4 ranges of 256 positions over the 8 waves");
    // operands of the cell, by step parity: {dout*mask, A, Ki, Kj | Kf, Ko, f, -} per cell thread, filled one step ahead by the
    // loader wave -- the cell waves issue NO loads, so they never wait on vmcnt (which would also drain their own
    // publishing / bookkeeping stores: 0.3 us of a 1.8 us step alone, 0.6 us with the weight-gradient GEMMs co-running)
    __shared__ __attribute__((aligned(16))) float opnd[2][NCELL][8];

    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NG = (a.B + R - 1) / R;
    const int ngroups = a.ND * NG;
    int grp, mem;
    if (((gridDim.x / G) & 7) == 0) { mem = (blockIdx.x >> 3) % G; grp = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * G)); }   // (grid padded to whole
    else { grp = blockIdx.x / G; mem = blockIdx.x % G; }                                                                      //  octets of groups: lstm.hip)
    if (grp >= ngroups) return;
    const int dir = grp / NG, bg = grp % NG;
    const int r0 = bg * R;
    const int H4 = 4 * H;
    const int j0 = mem * HS;

    // K_h rows of the own units (wave w: units 4w..4w+3) over this lane's PC positions -> registers
    float w[MF ? 1 : 4][MF ? 1 : PC];
    qbf16x8 wa[MF ? 8 : 1];      // MF: wave w = (unit tile w & 1, position range w >> 1); row m = lane & 15, k = 32 ks + 8 (lane >> 4) + jj
    if constexpr (MF) {
        const float* kh = a.kh[dir];
        const int unit = j0 + (wave & 1) * 16 + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int pos = (wave >> 1) * 256 + ks * 32 + 8 * (lane >> 4) + jj;
                wa[ks][jj] = (__bf16)kh[(size_t)unit * H4 + (pos & 3) * H + (pos >> 2)];
            }
        for (int idx = tid; idx < (R + 1) * N; idx += NT) dgb[idx] = 0;
    } else {
        const float* kh = a.kh[dir];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < PC; ++q) {
                const int pos = lane * PC + q;
                w[i][q] = kh[(size_t)(j0 + 4 * wave + i) * H4 + (pos & 3) * H + (pos >> 2)];
            }
    }
    int S = 0;
    for (int r = 0; r < R; ++r) S = max(S, (r0 + r < a.B) ? min(a.len[r0 + r], a.T) : 0);
    const bool cell = tid < NCELL;
    const bool cell_wave = __builtin_amdgcn_readfirstlane(tid) < NCW;
    const int cr = min(tid / HS, R - 1), cu = tid % HS;
    const int cb = r0 + cr;
    const int cb_safe = min(cb, a.B - 1);
    const int clen = (cell && cb < a.B) ? min(a.len[cb], a.T) : 0;
    const int cj = j0 + cu;
    float dc = 0.f;
    float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t* hxg = reinterpret_cast<uint32_t*>(a.hx) + (size_t)grp * 2 * R * N;      // [2 parities][R][N] tagged floats
    const bool fast = group_shares_xcd(a.xcc_slots + (size_t)grp * 16, G, mem, tid, a.err, nullptr, 0, R == 1 ? 16 : 6);

    // loader wave (last wave): lane l serves cell thread l
    const bool loader_wave = __builtin_amdgcn_readfirstlane(tid) >= NT - 64;
    const int ll = tid - (NT - 64);
    const bool lact = loader_wave && ll < NCELL;
    const int lcr = min(max(ll, 0) / HS, R - 1), lcu = max(ll, 0) % HS;
    const int lcb = min(r0 + lcr, a.B - 1);
    const int lclen = (lact && r0 + lcr < a.B) ? min(a.len[r0 + lcr], a.T) : 0;
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
    float dout_v = 0.f;
    auto prefetch = [&](int s) {
        const int t = dir ? s : (lclen - 1 - s);
        const int ts = min(max(t, 0), a.T - 1);
        const float4* rp = reinterpret_cast<const float4*>(a.act + ((((size_t)lcb * a.sb + (size_t)ts * a.st) * a.ND + dir) * H + j0 + lcu) * 8);
        ra = rp[0]; rb = rp[1];
        dout_v = a.dout[((size_t)lcb * a.osb + (size_t)ts * a.ost) * a.ldo + dir * H + j0 + lcu];
    };
    // Everything of the pointwise backward that does not depend on dh / dc is formed HERE, off the chain (the cell is a
    // dependent sequence of ~100 VALU ops on one wave otherwise: tanh, the dropout hash, the gate derivatives):
    //   dct = dc + dh*A;  dG = {dct*Ki, dct*Kj, dct*Kf, dh*Ko};  dc' = dct*f   with dh = dout*mask + recurrent part
    auto hand_over = [&](int s) {
        const int t = dir ? s : (lclen - 1 - s);
        const float gi = ra.x, gj = ra.y, gf = ra.z, go = ra.w, cc = rb.x, cp = rb.y;
        const float tc = fast_tanh(cc);
        float dm = dout_v;
        if (a.keep < 1.0f)
            dm *= keep_scale(a.seed, (uint32_t)((a.boff + r0 + lcr) * a.dsb + t * a.dst), (uint32_t)(dir * H + j0 + lcu), a.keep);
        float4* o = reinterpret_cast<float4*>(&opnd[s & 1][ll][0]);
        o[0] = make_float4(dm, go * (1.f - tc * tc), gj * gi * (1.f - gi), gi * (1.f - gj * gj));
        o[1] = make_float4(cp * gf * (1.f - gf), tc * go * (1.f - go), gf, 0.f);
    };
    if (lact) { prefetch(0); hand_over(0); if (S > 1) prefetch(1); }
    __syncthreads();

    for (int s = 0; s < S; ++s) {
        const bool live = cell && s < clen;
        const int t = dir ? s : (clen - 1 - s);
        if (s > 0) {
            if (lact) { hand_over(s); if (s + 1 < S) prefetch(s + 1); }
            if (!cell_wave && !loader_wave) {
                // all of this thread's quads in flight, re-polled together until every tag bit matches
                typedef unsigned int u32x4q __attribute__((ext_vector_type(4)));
                const uint32_t* src = hxg + (size_t)((s - 1) & 1) * R * N;
                const uint32_t want = ((((uint32_t)(s - 1)) >> 1) & 1u) ^ 1u;       // tag bit of the step that published
                bool need[NPP];
                const u32x4q* qp[NPP];
#pragma unroll
                for (int j = 0; j < NPP; ++j) {
                    const int qidx = tid - NCW + NPOLL * j;
                    need[j] = qidx < NQUAD && r0 + (4 * qidx) / N < a.B;
                    qp[j] = reinterpret_cast<const u32x4q*>(src) + min(qidx, NQUAD - 1);
                    if (qidx < NQUAD && !need[j]) {
                        const int idx = 4 * qidx, r = idx / N, pos = idx % N;
                        if constexpr (MF) *reinterpret_cast<uint2*>(dgb + r * N + pos) = make_uint2(0u, 0u);
                        else *reinterpret_cast<float4*>(dgl + (r * 64 + pos / PC) * CSB + (pos % PC)) = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
                long long t0w = 0;
                ASR_RACE_HUNT_DELAY();
                for (uint32_t spins = 0;; ++spins) {
                    u32x4q x[NPP];
                    if constexpr (NPP == 1) {
                        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x[0]) : "v"(qp[0]) : "memory");
                    } else if constexpr (NPP == 2) {
                        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                                     : "=&v"(x[0]), "=&v"(x[1]) : "v"(qp[0]), "v"(qp[1]) : "memory");
                    } else {
                        asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\t"
                                     "global_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                                     : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]) : "v"(qp[0]), "v"(qp[1]), "v"(qp[2]) : "memory");
                    }
                    bool pending = false;
#pragma unroll
                    for (int j = 0; j < NPP; ++j) {
                        if (!need[j]) continue;
                        const uint32_t bits = (x[j].x & 1u) + (x[j].y & 1u) + (x[j].z & 1u) + (x[j].w & 1u);
                        if (bits == 4u * want) {
                            const int idx = 4 * (tid - NCW + NPOLL * j), r = idx / N, pos = idx % N;
                            if constexpr (MF) {
                                union { __bf16 b[4]; uint2 u; } pk;
                                pk.b[0] = (__bf16)__uint_as_float(x[j].x & ~1u); pk.b[1] = (__bf16)__uint_as_float(x[j].y & ~1u);
                                pk.b[2] = (__bf16)__uint_as_float(x[j].z & ~1u); pk.b[3] = (__bf16)__uint_as_float(x[j].w & ~1u);
                                *reinterpret_cast<uint2*>(dgb + r * N + pos) = pk.u;
                            } else {
                                *reinterpret_cast<float4*>(dgl + (r * 64 + pos / PC) * CSB + (pos % PC)) =
                                    make_float4(__uint_as_float(x[j].x & ~1u), __uint_as_float(x[j].y & ~1u),
                                                __uint_as_float(x[j].z & ~1u), __uint_as_float(x[j].w & ~1u));
                            }
                            need[j] = false;
                        } else pending = true;
                    }
                    if (!pending) break;
                    ASR_POLL_BACKOFF();
                    if ((spins & 1023) == 1023) {
                        const long long now = wall_clock64();
                        if (t0w == 0) t0w = now;
                        else if (now - t0w > 200000000LL) { *a.err = 55; break; }
                        if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    }
                }
            }
            BPTT_STAMP(0)
            __syncthreads();
            BPTT_STAMP(1)
            if constexpr (MF) {
                const int n = lane & 15;
                const unsigned short* drow = dgb + (n < R ? n : R) * N + (wave >> 1) * 256 + 8 * (lane >> 4);
                f32x4 dacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const qbf16x8 bfrag = *reinterpret_cast<const qbf16x8*>(drow + ks * 32);
                    dacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks], bfrag, dacc, 0, 0, 0);
                }
                // D: col = lane & 15 = batch row, rows (lane >> 4) * 4 + reg = units of this wave's tile; partial of range wave >> 1
                if (n < R) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        sums[(n * HS + (wave & 1) * 16 + (lane >> 4) * 4 + e) * 4 + (wave >> 1)] = dacc[e];
                }
            } else {
            // dh_rec for the own units: 4 units x R rows per lane, contraction over this lane's PC positions
            float acc[R][4];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.f;
                const float4* dp = reinterpret_cast<const float4*>(dgl + (r * 64 + lane) * CSB);
#pragma unroll
                for (int q4 = 0; q4 < PC / 4; ++q4) {
                    const float4 dv = dp[q4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[r][i] = fmaf(dv.x, w[i][4 * q4 + 0], acc[r][i]);
                        acc[r][i] = fmaf(dv.y, w[i][4 * q4 + 1], acc[r][i]);
                        acc[r][i] = fmaf(dv.z, w[i][4 * q4 + 2], acc[r][i]);
                        acc[r][i] = fmaf(dv.w, w[i][4 * q4 + 3], acc[r][i]);
                    }
                }
            }
            // 16 lanes of a DPP row by butterflies; the 4 rows of the wave meet in LDS (fixed order in the cell)
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[r][i] = row16_allreduce_sum(acc[r][i]);
            if ((lane & 15) == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) sums[(r * HS + 4 * wave + i) * 4 + (lane >> 4)] = acc[r][i];
            }
            }   // !MF
            BPTT_STAMP(2)
            __syncthreads();
            BPTT_STAMP(3)
        }
        if (cell_wave && cell) {
            const float4 oa = *reinterpret_cast<const float4*>(&opnd[s & 1][tid][0]);   // {dout*mask, A, Ki, Kj}
            const float4 ob = *reinterpret_cast<const float4*>(&opnd[s & 1][tid][4]);   // {Kf, Ko, f, -}
            float dh = oa.x;
            if (s > 0) {
                const float4 v = *reinterpret_cast<const float4*>(sums + (cr * HS + cu) * 4);
                dh += (v.x + v.y) + (v.z + v.w);
            }
            float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) {
                const float dct = fmaf(dh, oa.y, dc);
                dg = make_float4(dct * oa.z, dct * oa.w, dct * ob.x, dh * ob.y);
                dc = dct * ob.z;
                dbs.x += dg.x; dbs.y += dg.y; dbs.z += dg.z; dbs.w += dg.w;      // bias gradient: sum of dG over time
            }
            // publish dG_s of this unit FIRST (zeros for rows past their length): one tagged quad
            if (cb < a.B && s + 1 < S) {
                uint32_t* dst = hxg + ((size_t)(s & 1) * R + cr) * N + 4 * cj;
                const uint32_t tb = ((((uint32_t)s) >> 1) & 1u) ^ 1u;
                typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
                const u32x4s g0 = {(__float_as_uint(dg.x) & ~1u) | tb, (__float_as_uint(dg.y) & ~1u) | tb,
                                   (__float_as_uint(dg.z) & ~1u) | tb, (__float_as_uint(dg.w) & ~1u) | tb};
                ASR_RACE_HUNT_DELAY();
                if (fast) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(g0) : "memory");
                else {
                    __hip_atomic_store(dst + 0, g0.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 1, g0.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 2, g0.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 3, g0.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (live) {       // bookkeeping, off the critical path
                float* gp = a.gates + (((size_t)cb * a.sb + (size_t)t * a.st) * a.ND + dir) * H4 + cj;
                gp[0] = dg.x; gp[H] = dg.y; gp[2 * H] = dg.z; gp[3 * H] = dg.w;
            }
        }
        BPTT_STAMP(4)
    }
    if (a.db_part && cell && cb < a.B) {        // one row per utterance and direction: summed over the batch by a tiny colsum
        float* dp = a.db_part + ((size_t)(a.boff + cb) * a.ND + dir) * H4 + cj;
        dp[0] = dbs.x; dp[H] = dbs.y; dp[2 * H] = dbs.z; dp[3 * H] = dbs.w;
    }
    if (STAMP && a.dbg && blockIdx.x == 0 && (tid == 0 || tid == NT - 1)) {
        for (int i = 0; i < 5; ++i) atomicAdd(a.dbg + 32 + (tid == 0 ? 0 : 8) + i, (unsigned long long)stamp[i]);
        if (tid == 0) atomicAdd(a.dbg + 32 + 7, (unsigned long long)S);
    }
#undef BPTT_STAMP
    // dG = 0 past each row's length (the weight/input GEMMs read every row)
    for (int r = 0; r < R; ++r) {
        if (r0 + r >= a.B) break;
        const int l = min(a.len[r0 + r], a.T);
        const int nz = a.T - l;
        for (int idx = tid; idx < nz * 4 * HS; idx += NT) {
            const int t = l + idx / (4 * HS), q = idx % (4 * HS);
            a.gates[(((size_t)(r0 + r) * a.sb + (size_t)t * a.st) * a.ND + dir) * H4 + (q / HS) * H + j0 + (q % HS)] = 0.f;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Version 2 of the all-gather BPTT (round 3): one SOURCE SLICE PER WAVE, as csrc/lstm.hip's version 2 of the forward.
//
// Wave w (1 .. G-1) of a workgroup polls exactly the R x 32 tagged quads that the cell wave of source workgroup
// (mem + w) % G published with one store instruction (one 16-byte sc1 load per lane), stages them in a wave-private LDS
// region (no barrier) and contracts them at once with the K_h rows of the workgroup's own 32 units over that source's 128
// gate positions; wave 0 is the cell wave and contracts the own slice for the NEXT step right after publishing; the last
// wave is the loader of the first version (activation records one step ahead, operands of the pointwise backward handed
// over through LDS).  ONE barrier per step; partial sums double-buffered by step parity, summed by the cell thread in fixed
// order.  Lane map of the contraction: lane = 8*pq + ug -- units 4ug .. 4ug+3, positions 16pq .. 16pq+15 of the slice: 64 weight
// registers and 16 x R dG values from LDS per lane (dG is 4H wide: with fewer positions per lane the LDS return path
// binds), then a reduce-scatter over the 8 position groups inside the wave: one DPP row rotation, v_permlane16_swap and
// v_permlane32_swap, after which every lane holds ONE finished (row, unit) value.  Products on v_pk_fma_f32 (two rows x one
// weight, or with one row two units x one dG value).  R in {1, 2}.
// ---------------------------------------------------------------------------------------------------------------
template <int H, int R, bool STAMP = false>
__global__ __launch_bounds__(2 * H + 64) void lstm_rec_bwd2_kernel(LstmBwdArgs a) {
    unsigned int stamp[5] = {0, 0, 0, 0, 0};
    unsigned long long tlast = STAMP ? __builtin_amdgcn_s_memtime() : 0;
#define BPTT_STAMP(i) if (STAMP) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); stamp[i] += (unsigned int)(t__ - tlast); tlast = t__; }
    // STAMP: every wave's busy time between two step barriers (exit of one to arrival at the next) -> dbg[48 + wave]: the wave
    // with the largest sum is the one the others wait for
    unsigned long long busy = 0, texit = 0;
#define BPTT_BAR() { if (STAMP) busy += __builtin_amdgcn_s_memtime() - texit; __syncthreads(); if (STAMP) texit = __builtin_amdgcn_s_memtime(); }
    constexpr int HS = 32;
    constexpr int NS = H / 32;         // source slices = contraction waves (wave 0: cell + own slice)
    constexpr int G = NS;
    constexpr int NT = NS * 64 + 64;   // + the loader wave
    constexpr int N = 4 * H;
    constexpr int NCELL = R * HS;
    static_assert(R == 1 || R == 2, "the cell is one wave");
    // staging of one slice's dG: position group pq (16 positions) at pq*DST; R = 2: (dG[0][p], dG[1][p]) pairs; padded so
    // that the 8 groups' float4 reads fall into different banks
    constexpr int DST = R == 2 ? 36 : 20;
    __shared__ __attribute__((aligned(16))) float dgs[NS][8 * DST];
    __shared__ __attribute__((aligned(16))) float part[2][NS][64];
    __shared__ __attribute__((aligned(16))) float opnd[2][NCELL][8];

    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NG = (a.B + R - 1) / R;
    const int ngroups = a.ND * NG;
    int grp, mem;
    if (((gridDim.x / G) & 7) == 0) { mem = (blockIdx.x >> 3) % G; grp = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * G)); }
    else { grp = blockIdx.x / G; mem = blockIdx.x % G; }
    if (grp >= ngroups) return;
    const int dir = grp / NG, bg = grp % NG;
    const int r0 = bg * R;
    const int H4 = 4 * H;
    const int j0 = mem * HS;
    const bool cell_wave = wave == 0;
    const bool loader_wave = wave == NS;
    const int src_wg = (mem + wave) % G;           // contraction waves: whose dG this wave consumes (wave 0: the own)
    const int pq = lane >> 3, ug = lane & 7;

    // K_h rows of units j0 + 4ug + i over positions 16pq + q of the slice (position = 4*unit + gate)
    f32x2 wp[4][8];            // R = 2: wp[i][q/2] = (w[i][q], w[i][q+1]);  R = 1: wp[i/2 + 2*(q&1)][q/2] = (w[i][q], w[i+1][q]), see below
    if (!loader_wave) {
        const float* kh = a.kh[dir];
        auto wv = [&](int i, int q) {
            return kh[(size_t)(j0 + 4 * ug + i) * H4 + (q & 3) * H + src_wg * 32 + pq * 4 + (q >> 2)];
        };
        if constexpr (R == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q2 = 0; q2 < 8; ++q2) wp[i][q2] = f32x2{wv(i, 2 * q2), wv(i, 2 * q2 + 1)};
        } else {
            // pairs over units: entry [2*(q & 1) + ip][q >> 1] = (w[2ip][q], w[2ip+1][q])
#pragma unroll
            for (int q = 0; q < 16; ++q)
#pragma unroll
                for (int ip = 0; ip < 2; ++ip) wp[2 * (q & 1) + ip][q >> 1] = f32x2{wv(2 * ip, q), wv(2 * ip + 1, q)};
        }
    }
    int S = 0;
    for (int r = 0; r < R; ++r) S = max(S, (r0 + r < a.B) ? min(a.len[r0 + r], a.T) : 0);
    const bool cell = tid < NCELL;
    const int cr = min(tid / HS, R - 1), cu = tid % HS;
    const int cb = r0 + cr;
    const int clen = (cell && cb < a.B) ? min(a.len[cb], a.T) : 0;
    const int cj = j0 + cu;
    float dc = 0.f;
    float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t* hxg = reinterpret_cast<uint32_t*>(a.hx) + (size_t)grp * 2 * R * N;      // [2 parities][R][N] tagged floats
    const bool fast = group_shares_xcd(a.xcc_slots + (size_t)grp * 16, G, mem, tid, a.err, nullptr, 0, R == 1 ? 17 : 7);

    // loader wave: lane l serves cell thread l
    const int ll = lane;
    const bool lact = loader_wave && ll < NCELL;
    const int lcr = min(ll / HS, R - 1), lcu = ll % HS;
    const int lcb = min(r0 + lcr, a.B - 1);
    const int lclen = (lact && r0 + lcr < a.B) ? min(a.len[r0 + lcr], a.T) : 0;
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
    float dout_v = 0.f;
    auto prefetch = [&](int s) {
        const int t = dir ? s : (lclen - 1 - s);
        const int ts = min(max(t, 0), a.T - 1);
        const float4* rp = reinterpret_cast<const float4*>(a.act + ((((size_t)lcb * a.sb + (size_t)ts * a.st) * a.ND + dir) * H + j0 + lcu) * 8);
        ra = rp[0]; rb = rp[1];
        dout_v = a.dout[((size_t)lcb * a.osb + (size_t)ts * a.ost) * a.ldo + dir * H + j0 + lcu];
    };
    auto hand_over = [&](int s) {
        const int t = dir ? s : (lclen - 1 - s);
        const float gi = ra.x, gj = ra.y, gf = ra.z, go = ra.w, cc = rb.x, cp = rb.y;
        const float tc = fast_tanh(cc);
        float dm = dout_v;
        if (a.keep < 1.0f)
            dm *= keep_scale(a.seed, (uint32_t)((a.boff + r0 + lcr) * a.dsb + t * a.dst), (uint32_t)(dir * H + j0 + lcu), a.keep);
        float4* o = reinterpret_cast<float4*>(&opnd[s & 1][ll][0]);
        o[0] = make_float4(dm, go * (1.f - tc * tc), gj * gi * (1.f - gi), gi * (1.f - gj * gj));
        o[1] = make_float4(cp * gf * (1.f - gf), tc * go * (1.f - go), gf, 0.f);
    };
    if (lact) { prefetch(0); hand_over(0); if (S > 1) prefetch(1); }
    __syncthreads();
    if (STAMP) texit = __builtin_amdgcn_s_memtime();

    // stage the quad (4 gates) of slice unit su, row r
    auto stage = [&](int r, int su, float4 v) {
        float* d = &dgs[wave][(su >> 2) * DST];
        const int q = (su & 3) * 4;
        if constexpr (R == 2) { d[2 * q + r] = v.x; d[2 * q + 2 + r] = v.y; d[2 * q + 4 + r] = v.z; d[2 * q + 6 + r] = v.w; }
        else *reinterpret_cast<float4*>(d + q) = v;
    };
    // this wave's slice (already staged) against its weights -> partials of the own units' dh, one (row, unit) per lane
    auto slice_partial = [&](int par) {
        const f32x4* dq = reinterpret_cast<const f32x4*>(&dgs[wave][pq * DST]);
        if constexpr (R == 2) {
            f32x4 dall[8];
#pragma unroll
            for (int q2 = 0; q2 < 8; ++q2) dall[q2] = dq[q2];
            f32x2 acc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = f32x2{0.f, 0.f};
#pragma unroll
            for (int q2 = 0; q2 < 8; ++q2) {
                const f32x4 dv = dall[q2];
                const f32x2 d0 = __builtin_shufflevector(dv, dv, 0, 1), d1 = __builtin_shufflevector(dv, dv, 2, 3);
#pragma unroll
                for (int i = 0; i < 4; ++i) { pk_fma_blo(acc[i][0], d0, wp[i][q2]); pk_fma_bhi(acc[i][1], d1, wp[i][q2]); }
            }
            float z[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2 t2 = acc[i][0] + acc[i][1];
                float x = t2.x, y = t2.y;                       // rows 0, 1 of unit i over this lane's 16 positions
                x += dpp_mov<0x128>(x); y += dpp_mov<0x128>(y); // + the lane 8 further in the row of 16 (row_ror:8)
                z[i] = swap16_add(x, y);                        // even rows of 16: row 0 of the group; odd rows: row 1
            }
            const float q0 = swap32_add(z[0], z[1]);            // lower half-wave: unit 0, upper: unit 1
            const float q1 = swap32_add(z[2], z[3]);            //                  unit 2,        unit 3
            const int b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = lane >> 5;
            part[par][wave][b4 * 32 + 4 * ug + 2 * b3 + b5] = b3 ? q1 : q0;
        } else {
            f32x4 dall[4];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) dall[q4] = dq[q4];
            f32x2 acc[2][2];       // [unit pair][chain]
            acc[0][0] = acc[0][1] = acc[1][0] = acc[1][1] = f32x2{0.f, 0.f};
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 dv = dall[q4];
                const f32x2 d01 = __builtin_shufflevector(dv, dv, 0, 1), d23 = __builtin_shufflevector(dv, dv, 2, 3);
#pragma unroll
                for (int ip = 0; ip < 2; ++ip) {        // position q = 4q4 + e: weight entry [2*(q & 1) + ip][q >> 1]
                    pk_fma_alo(acc[ip][0], d01, wp[ip][2 * q4]);
                    pk_fma_ahi(acc[ip][1], d01, wp[2 + ip][2 * q4]);
                    pk_fma_alo(acc[ip][0], d23, wp[ip][2 * q4 + 1]);
                    pk_fma_ahi(acc[ip][1], d23, wp[2 + ip][2 * q4 + 1]);
                }
            }
            float v[4];
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                const f32x2 t2 = acc[ip][0] + acc[ip][1];
                v[2 * ip] = t2.x + dpp_mov<0x128>(t2.x);
                v[2 * ip + 1] = t2.y + dpp_mov<0x128>(t2.y);
            }
            const float z0 = swap16_add(v[0], v[1]);            // even rows: unit 0, odd rows: unit 1
            const float z1 = swap16_add(v[2], v[3]);            //            unit 2,           unit 3
            const float qq = swap32_add(z0, z1);                // lower half-wave keeps z0, upper z1
            const int b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = lane >> 5;
            if (!b3) part[par][wave][4 * ug + 2 * b5 + b4] = qq;
        }
    };

    for (int s = 0; s < S; ++s) {
        const bool live = cell && s < clen;
        const int t = dir ? s : (clen - 1 - s);
        const int par = s & 1;
        if (s > 0) {
            if (loader_wave) {
                if (lact) { hand_over(s); if (s + 1 < S) prefetch(s + 1); }
            } else if (!cell_wave) {
                BPTT_STAMP(0)
                if (lane < NCELL) {
                    const int r = lane >> 5, su = lane & 31;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (r0 + r < a.B)
                        tagged_poll4(hxg + ((size_t)((s - 1) & 1) * R + r) * N + 4 * (src_wg * 32 + su), tag_bit(s - 1), v, a.err);
                    stage(r, su, v);
                }
                BPTT_STAMP(1)
                __builtin_amdgcn_wave_barrier();
                slice_partial(par);
                BPTT_STAMP(2)
            }
            if (cell_wave) { BPTT_STAMP(0) }
            BPTT_BAR()
            if (cell_wave) { BPTT_STAMP(1) } else if (!loader_wave) { BPTT_STAMP(3) }
        }
        if (cell_wave) {
            float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cell) {
                const float4 oa = *reinterpret_cast<const float4*>(&opnd[par][tid][0]);   // {dout*mask, A, Ki, Kj}
                const float4 ob = *reinterpret_cast<const float4*>(&opnd[par][tid][4]);   // {Kf, Ko, f, -}
                float dh = oa.x;
                if (s > 0) {
                    float rec = 0.f;
#pragma unroll
                    for (int ww = 0; ww < NS; ++ww) rec += part[par][ww][tid];
                    dh += rec;
                }
                if (live) {
                    const float dct = fmaf(dh, oa.y, dc);
                    dg = make_float4(dct * oa.z, dct * oa.w, dct * ob.x, dh * ob.y);
                    dc = dct * ob.z;
                    dbs.x += dg.x; dbs.y += dg.y; dbs.z += dg.z; dbs.w += dg.w;      // bias gradient: sum of dG over time
                }
                // the value everyone (the owner included) consumes is the tagged, truncated one
                const uint32_t tb = tag_bit(s);
                typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
                const u32x4s g0 = {(__float_as_uint(dg.x) & ~1u) | tb, (__float_as_uint(dg.y) & ~1u) | tb,
                                   (__float_as_uint(dg.z) & ~1u) | tb, (__float_as_uint(dg.w) & ~1u) | tb};
                if (cb < a.B && s + 1 < S) {       // publish dG_s of this unit FIRST: one tagged quad
                    uint32_t* dst = hxg + ((size_t)par * R + cr) * N + 4 * cj;
                    ASR_RACE_HUNT_DELAY();
                    if (fast) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(g0) : "memory");
                    else {
                        __hip_atomic_store(dst + 0, g0.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dst + 1, g0.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dst + 2, g0.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dst + 3, g0.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                BPTT_STAMP(2)
                if (live) {       // bookkeeping, off the critical path
                    float* gp = a.gates + (((size_t)cb * a.sb + (size_t)t * a.st) * a.ND + dir) * H4 + cj;
                    gp[0] = dg.x; gp[H] = dg.y; gp[2 * H] = dg.z; gp[3 * H] = dg.w;
                }
                if (s + 1 < S)    // own slice of the next step's contraction: the truncated values the peers will read
                    stage(cr, cu, make_float4(__uint_as_float(g0.x & ~1u), __uint_as_float(g0.y & ~1u),
                                              __uint_as_float(g0.z & ~1u), __uint_as_float(g0.w & ~1u)));
            }
            if (s + 1 < S) {
                BPTT_STAMP(3)
                __builtin_amdgcn_wave_barrier();
                slice_partial(par ^ 1);
                BPTT_STAMP(4)
            }
        }
    }
    if (a.db_part && cell && cb < a.B) {        // one row per utterance and direction: summed over the batch by a tiny colsum
        float* dp = a.db_part + ((size_t)(a.boff + cb) * a.ND + dir) * H4 + cj;
        dp[0] = dbs.x; dp[H] = dbs.y; dp[2 * H] = dbs.z; dp[3 * H] = dbs.w;
    }
    if (STAMP && a.dbg && blockIdx.x == 0 && (tid == 0 || tid == 64)) {
        for (int i = 0; i < 5; ++i) atomicAdd(a.dbg + 32 + (tid == 0 ? 0 : 8) + i, (unsigned long long)stamp[i]);
        if (tid == 0) atomicAdd(a.dbg + 32 + 7, (unsigned long long)S);
    }
    if (STAMP && a.dbg && blockIdx.x == 0 && lane == 0) atomicAdd(a.dbg + 48 + wave, busy);
#undef BPTT_STAMP
#undef BPTT_BAR
    // dG = 0 past each row's length (the weight/input GEMMs read every row)
    for (int r = 0; r < R; ++r) {
        if (r0 + r >= a.B) break;
        const int l = min(a.len[r0 + r], a.T);
        const int nz = a.T - l;
        for (int idx = tid; idx < nz * 4 * HS; idx += NT) {
            const int tt = l + idx / (4 * HS), q = idx % (4 * HS);
            a.gates[(((size_t)(r0 + r) * a.sb + (size_t)tt * a.st) * a.ND + dir) * H4 + (q / HS) * H + j0 + (q % HS)] = 0.f;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Groups of FOUR workgroups (H = 256: 64 units each, one batch row per group), as lstm_rec_fwd4_kernel of csrc/lstm.hip.
// Wave w contracts the 32 tagged quads (128 dG positions) that source workgroup (mem + (w >> 1)) % 4 published for half (w & 1)
// of its units with the K_h rows of the workgroup's own 64 units: lane = own unit, 128 weight registers, the dG values
// broadcast from a wave-private LDS row, no cross-lane reduction; the cell wave (wave 0) sums the eight partials.  Eight
// waves only: with a ninth (loader) wave the register file allows 168 per wave, less than the weights + operands; the cell
// wave itself fetches the activation record of the NEXT step and turns it into the pointwise operands in the slack after its
// own contraction (it never polls, so its loads delay nobody's polls), one step ahead as the loader wave did.
// ---------------------------------------------------------------------------------------------------------------
// REC32: the saved activations are the 32-byte records {i,j,f,o | c, c_prev, -, -} of the decoder's LM cell chain (written by the
// one-launch training decoder, csrc/decoder_greedy.hip) instead of the encoder's 20-byte split records: the time-major LM-chain
// BPTT (asr_lstm_rec_bwd_tm) then runs on this kernel too -- 32 groups x 4 = 128 workgroups, half the chip.
// QUAD (round 5; ASR_BPTT_QUAD=0 keeps the mapping of rounds 3-4): the broadcast of the dG slice through LDS was ON the chain.
// With lane = own unit every lane of a wave needs all 128 values of the wave's half slice: 32 ds_read_b128 per lane, 1 KB of
// return data each, 7 polling waves at once -- ~900 cycles of the LDS return path per step (a timing-only build that read 8 of
// the 32 quads ran 1.02 instead of 1.17 us per step; the forward, whose lanes need 32 values, is at 1.0).  QUAD: the four lanes
// of a quad share four own units and split the 128 values: lane (u4, r) contracts the 32 values of source units 8r .. 8r+7 with
// the rows of own units 4 u4 .. 4 u4 + 3 (still 128 weights in registers, 64 packed FMAs), 8 ds_read_b128 per lane, and the
// quad's four partial sums of each own unit meet in three DPP adds (reduce-scatter over quad_perm; fixed order).
template <bool REC32 = false, bool QUAD = false>
__global__ __launch_bounds__(512) void lstm_rec_bwd4_kernel(LstmBwdArgs a) {
    constexpr int H = 256, HS = 64, G = 4, NW = 8, NT = 512, H4 = 4 * H, N = 4 * H;
    constexpr unsigned GSTR = REC32 ? 8u : 4u, CSTR = REC32 ? 8u : 1u;     // floats per unit-step in the gates / c arrays
    // QUAD: the quad of source unit su sits at float 4 * (su + su / 8): the four 8-unit ranges a quad's lanes read start 36
    // floats apart, i.e. in different banks (128 floats apart they would be a 4-way conflict)
    __shared__ __attribute__((aligned(16))) float dgs[NW][QUAD ? 144 : 128];
    __shared__ __attribute__((aligned(16))) float part[2][NW][HS];
    __shared__ __attribute__((aligned(16))) float opnd[2][HS][8];

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
    const int j0 = mem * HS, cj = j0 + lane;
    const bool cell_wave = wave == 0;
    const int sbase = ((mem + (wave >> 1)) % G) * HS + (wave & 1) * 32;      // first source unit of this wave's half slice

    // K_h row of the own unit cj over the wave's 128 positions q = 4*su + gate (column gate*H + sbase + su), as pairs.
    // lane = own unit means every lane reads ANOTHER ROW of K_h (4 KB apart): read straight from memory that is 128 scattered
    // 4-byte loads per lane, 64 lines per wave instruction -- ~15 us of prologue per launch, four launches per step (round 5:
    // the BPTT of 100 steps took 126 us, the forward 95).  Coalesced instead: a wave instruction covers 32 consecutive floats
    // of two rows, the 64 x 32 tile of one gate goes through a wave-private LDS tile (pitch 33: conflict-free both ways) and
    // each lane picks up its own row.
    f32x2 wq[64];
    if constexpr (QUAD) {
        // wq[2 * (8 i + m) + pair]: own unit 4 (lane / 4) + i, source unit 8 (lane % 4) + m.  Straight from memory: a lane's 8
        // source units are 32 contiguous bytes of a K_h row, so one float4 wave instruction touches 16 rows x one 128-byte line --
        // 32 loads per lane, all in flight, no LDS (the lane = own unit layout needed the tile below: four rounds of 32 loads +
        // transpose, ~14 us of prologue more than the forward's at every launch)
        const float* kr = a.kh[dir] + (size_t)(j0 + 4 * (lane >> 2)) * H4 + sbase + 8 * (lane & 3);
        float4 v[4][4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int mq = 0; mq < 2; ++mq) v[i][g][mq] = *reinterpret_cast<const float4*>(kr + (size_t)i * H4 + g * H + 4 * mq);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int mq = 0; mq < 2; ++mq) {
                const int su = 8 * i + 4 * mq;
                wq[2 * su + 0] = f32x2{v[i][0][mq].x, v[i][1][mq].x}; wq[2 * su + 1] = f32x2{v[i][2][mq].x, v[i][3][mq].x};
                wq[2 * su + 2] = f32x2{v[i][0][mq].y, v[i][1][mq].y}; wq[2 * su + 3] = f32x2{v[i][2][mq].y, v[i][3][mq].y};
                wq[2 * su + 4] = f32x2{v[i][0][mq].z, v[i][1][mq].z}; wq[2 * su + 5] = f32x2{v[i][2][mq].z, v[i][3][mq].z};
                wq[2 * su + 6] = f32x2{v[i][0][mq].w, v[i][1][mq].w}; wq[2 * su + 7] = f32x2{v[i][2][mq].w, v[i][3][mq].w};
            }
    } else {
        __shared__ float wt[NW][64 * 33];
        float* tile = &wt[wave][0];
        const float* kr = a.kh[dir] + (size_t)j0 * H4 + sbase + (lane & 31);
        const int rh = lane >> 5;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = kr[(size_t)(2 * i + rh) * H4 + g * H];
#pragma unroll
            for (int i = 0; i < 32; ++i) tile[(2 * i + rh) * 33 + (lane & 31)] = v[i];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int su = 0; su < 32; ++su) {
                const float x = tile[lane * 33 + su];
                if (g == 0) wq[2 * su].x = x; else if (g == 1) wq[2 * su].y = x; else if (g == 2) wq[2 * su + 1].x = x; else wq[2 * su + 1].y = x;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    const int S = min(a.len[cb], a.T);
    float dc = 0.f;
    float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t* hxg = reinterpret_cast<uint32_t*>(a.hx) + (size_t)grp * 2 * N;           // [2 parities][N] tagged floats
    const bool fast = group_shares_xcd(a.xcc_slots + (size_t)grp * 16, G, mem, tid, a.err, nullptr, 0, 17);

    // cell wave: the record of step s (time t) -> registers; then -> the operands of the pointwise backward in LDS
    // 20-byte split records (LstmRecArgs::act_c): gates {i,j,f,o} in one plane, c in another; c_prev of a step IS the c of the
    // step the BPTT takes next (the forward's previous step), 0 behind the first forward step.
    // (32-bit element offsets from per-thread base pointers: 64-bit index arithmetic for every address of every step sat on
    // the cell wave, the wave all others wait for.)
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f);
    float rc_ = 0.f, rcn = 0.f, dout_v = 0.f;
    const float* const act_g = a.act + ((((size_t)cb * a.sb) * a.ND + dir) * H + cj) * GSTR;
    const float* const act_cc = REC32 ? act_g + 4 : a.act_c + (((size_t)cb * a.sb) * a.ND + dir) * H + cj;
    const float* const dout_b = a.dout + ((size_t)cb * a.osb) * a.ldo + dir * H + cj;
    const unsigned tstr = (unsigned)(a.st * a.ND * H), ostr = (unsigned)(a.ost * a.ldo);
    auto time_of = [&](int s) { const int t = dir ? s : (S - 1 - s); return (unsigned)min(max(t, 0), a.T - 1); };
    auto prefetch = [&](int s) {
        const unsigned ts = time_of(s), tn = time_of(min(s + 1, S - 1));
        ra = *reinterpret_cast<const float4*>(act_g + ts * tstr * GSTR);
        // c of this step AND of the next one (= this step's c_prev), both loaded here: carrying the second over to the next step
        // in a register made the compiler copy it at the loop's back edge, i.e. wait for the load it had just issued.  The two
        // are neighbours in time: the second load is the next step's first and hits the same lines.
        rc_ = act_cc[ts * tstr * CSTR];
        rcn = act_cc[tn * tstr * CSTR];
        dout_v = dout_b[ts * ostr];
    };
    auto hand_over = [&](int s) {
        const int t = dir ? s : (S - 1 - s);
        const float gi = ra.x, gj = ra.y, gf = ra.z, go = ra.w;
        const float rcp = s + 1 < S ? rcn : 0.f;
        const float tc = fast_tanh(rc_);
        float dm = dout_v;
        if (a.keep < 1.0f)
            dm *= keep_scale(a.seed, (uint32_t)((a.boff + cb) * a.dsb + t * a.dst), (uint32_t)(dir * H + cj), a.keep);
        float4* o = reinterpret_cast<float4*>(&opnd[s & 1][lane][0]);
        o[0] = make_float4(dm, go * (1.f - tc * tc), gj * gi * (1.f - gi), gi * (1.f - gj * gj));
        o[1] = make_float4(rcp * gf * (1.f - gf), tc * go * (1.f - go), gf, 0.f);
    };
    if (cell_wave && S > 0) { prefetch(0); hand_over(0); if (S > 1) prefetch(1); }
    __syncthreads();

    // this wave's half slice (staged in dgs[wave]) against its weights -> the partial of dh of the own unit
    auto slice_partial = [&](int par) {
        if constexpr (QUAD) {
            const f32x4* dq = reinterpret_cast<const f32x4*>(&dgs[wave][36 * (lane & 3)]);
            f32x4 dv[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) dv[m] = dq[m];
            f32x2 acc[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};      // one per own unit of the quad
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const f32x2 d01 = __builtin_shufflevector(dv[m], dv[m], 0, 1), d23 = __builtin_shufflevector(dv[m], dv[m], 2, 3);
#pragma unroll
                for (int i = 0; i < 4; ++i) asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(d01), "v"(wq[2 * (8 * i + m)]));
#pragma unroll
                for (int i = 0; i < 4; ++i) asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(d23), "v"(wq[2 * (8 * i + m) + 1]));
            }
            const float p0 = acc[0].x + acc[0].y, p1 = acc[1].x + acc[1].y, p2 = acc[2].x + acc[2].y, p3 = acc[3].x + acc[3].y;
            // reduce-scatter over the quad: lane r ends with own unit 4 u4 + r summed over the four lanes
            const bool odd = lane & 1, hi = lane & 2;
            float ka = odd ? p1 : p0, kb = odd ? p3 : p2;                     // kept: indices with bit 0 = r & 1
            const float sa = odd ? p0 : p1, sb = odd ? p2 : p3;               // sent to lane r ^ 1
            ka += dpp_mov<0xB1>(sa); kb += dpp_mov<0xB1>(sb);                 // quad_perm [1,0,3,2]
            float k = hi ? kb : ka;                                           // kept: index (r & 1) + 2 * (r >> 1) = r
            const float sx = hi ? ka : kb;                                    // sent to lane r ^ 2
            k += dpp_mov<0x4E>(sx);                                           // quad_perm [2,3,0,1]
            part[par][wave][lane] = k;
            return;
        }
        const f32x4* dq = reinterpret_cast<const f32x4*>(&dgs[wave][0]);
        f32x2 acc[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 dv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) dv[i] = dq[8 * c + i];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x2 d01 = __builtin_shufflevector(dv[i], dv[i], 0, 1), d23 = __builtin_shufflevector(dv[i], dv[i], 2, 3);
                asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[(2 * i) & 3]) : "v"(d01), "v"(wq[2 * (8 * c + i)]));
                asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[(2 * i + 1) & 3]) : "v"(d23), "v"(wq[2 * (8 * c + i) + 1]));
            }
        }
        const f32x2 t2 = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        part[par][wave][lane] = t2.x + t2.y;
    };

    for (int s = 0; s < S; ++s) {
        const int t = dir ? s : (S - 1 - s);
        const int par = s & 1;
        if (s > 0) {
            if (!cell_wave) {
                if (lane < 32) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    tagged_poll4(hxg + (size_t)((s - 1) & 1) * N + 4 * (sbase + lane), tag_bit(s - 1), v, a.err);
                    *reinterpret_cast<float4*>(&dgs[wave][QUAD ? 4 * (lane + (lane >> 3)) : 4 * lane]) = v;
                }
                __builtin_amdgcn_wave_barrier();
                slice_partial(par);
            }
            __syncthreads();
        }
        if (cell_wave) {
            const float4 oa = *reinterpret_cast<const float4*>(&opnd[par][lane][0]);   // {dout*mask, A, Ki, Kj}
            const float4 ob = *reinterpret_cast<const float4*>(&opnd[par][lane][4]);   // {Kf, Ko, f, -}
            float dh = oa.x;
            if (s > 0) {
                float rec = 0.f;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) rec += part[par][ww][lane];
                dh += rec;
            }
            const float dct = fmaf(dh, oa.y, dc);
            const float4 dg = make_float4(dct * oa.z, dct * oa.w, dct * ob.x, dh * ob.y);
            dc = dct * ob.z;
            dbs.x += dg.x; dbs.y += dg.y; dbs.z += dg.z; dbs.w += dg.w;
            const uint32_t tb = tag_bit(s);
            typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
            const u32x4s g0 = {(__float_as_uint(dg.x) & ~1u) | tb, (__float_as_uint(dg.y) & ~1u) | tb,
                               (__float_as_uint(dg.z) & ~1u) | tb, (__float_as_uint(dg.w) & ~1u) | tb};
            const bool more = s + 1 < S;
            if (more) {       // publish dG_s of this unit FIRST: one tagged quad
                uint32_t* dst = hxg + (size_t)par * N + 4 * cj;
                ASR_RACE_HUNT_DELAY();
                if (fast) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(g0) : "memory");
                else {
                    __hip_atomic_store(dst + 0, g0.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 1, g0.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 2, g0.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 3, g0.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            {       // bookkeeping, off the critical path
                if (a.dg_f32) {
                    float* gp = a.gates + (((size_t)cb * a.sb + (size_t)t * a.st) * a.ND + dir) * H4 + cj;
                    gp[0] = dg.x; gp[H] = dg.y; gp[2 * H] = dg.z; gp[3 * H] = dg.w;
                }
                if (a.dg_p3)
                    p3_store4(a.dg_p3, p3_elem_off((size_t)cb * a.sb + (size_t)t * a.st, dir * H4 + 4 * cj, (a.ND * H4) >> 3, a.p3_np),
                              dg.x, dg.y, dg.z, dg.w, a.p3_np, true);
            }
            if (more) {
                // first half of the own slice for the next step: the truncated values the peers will read
                if (lane < 32)
                    *reinterpret_cast<float4*>(&dgs[0][QUAD ? 4 * (lane + (lane >> 3)) : 4 * lane]) = make_float4(__uint_as_float(g0.x & ~1u), __uint_as_float(g0.y & ~1u),
                                                                                __uint_as_float(g0.z & ~1u), __uint_as_float(g0.w & ~1u));
                __builtin_amdgcn_wave_barrier();
                slice_partial(par ^ 1);
                // the next step's operands (its record was requested one step ago), then the request for the step after it
                hand_over(s + 1);
                if (s + 2 < S) prefetch(s + 2);
            }
        }
    }
    if (a.db_part && cell_wave) {        // one row per utterance and direction: summed over the batch by a tiny colsum
        float* dp = a.db_part + ((size_t)(a.boff + cb) * a.ND + dir) * H4 + cj;
        dp[0] = dbs.x; dp[H] = dbs.y; dp[2 * H] = dbs.z; dp[3 * H] = dbs.w;
    }
    {   // dG = 0 past the row's length (the weight/input GEMMs read every row)
        const int nz = a.T - S;
        if (a.dg_f32)
        for (int idx = tid; idx < nz * 4 * HS; idx += NT) {
            const int tt = S + idx / (4 * HS), q = idx % (4 * HS);
            a.gates[(((size_t)cb * a.sb + (size_t)tt * a.st) * a.ND + dir) * H4 + (q / HS) * H + j0 + (q % HS)] = 0.f;
        }
        if (a.dg_p3) {       // the workgroup's 4 * 64 unit-major columns of a row: HS / 2 chunks = HS / 2 * np contiguous pieces
            const int ppr = HS / 2 * a.p3_np;
            for (int idx = tid; idx < nz * ppr; idx += NT) {
                const int tt = S + idx / ppr;
                char* rowp = a.dg_p3 + p3_elem_off((size_t)cb * a.sb + (size_t)tt * a.st, dir * H4 + 4 * j0, (a.ND * H4) >> 3, a.p3_np);
                reinterpret_cast<uint4*>(rowp)[idx % ppr] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
}

}  // namespace asr
extern "C" int asr_get_gemm_precision(void);
extern "C" int asr_get_lstm_mfma(void);
int asr_lstm_max_wgs();
namespace asr {
template <int H>
static int launch_bwd_h(hipStream_t s, const LstmBwdArgs& a, int R) {
    int grid = a.ND * ((a.B + R - 1) / R) * (H / 32);
    { const int padded = ((a.ND * ((a.B + R - 1) / R) + 7) & ~7) * (H / 32); if (padded <= asr_lstm_max_wgs()) grid = padded; }
    static const bool allgather = [] { const char* e = getenv("ASR_BPTT_AG"); return !(e && e[0] == '0'); }();
    if (allgather && R <= 2) {     // more rows per group: too many granule loads per polling thread -> reduce-scatter kernel
        const bool mf_env = asr_get_lstm_mfma() != 0;      // opt-in since round 3 (csrc/lstm.hip)
        static const bool v2 = [] { const char* e = getenv("ASR_LSTM_V2"); return !(e && e[0] == '0'); }();
        if (H == 256 && mf_env && !a.dbg && asr_get_gemm_precision() == 1) {      // bf16 mode: contraction on the bf16 matrix pipe
            if (R == 1) hipLaunchKernelGGL((lstm_rec_bwd_ag_kernel<256, 1, false, true>), dim3(grid), dim3(512), 0, s, a);
            else hipLaunchKernelGGL((lstm_rec_bwd_ag_kernel<256, 2, false, true>), dim3(grid), dim3(512), 0, s, a);
        }
        else if (H <= 256 && v2 && !a.dbg) {       // version 2: slice per wave, one barrier per step (ASR_LSTM_V2=0 keeps version 1)
            if (R == 1) hipLaunchKernelGGL((lstm_rec_bwd2_kernel<(H <= 256 ? H : 256), 1>), dim3(grid), dim3(2 * H + 64), 0, s, a);
            else hipLaunchKernelGGL((lstm_rec_bwd2_kernel<(H <= 256 ? H : 256), 2>), dim3(grid), dim3(2 * H + 64), 0, s, a);
        }
        else if (H == 256 && R == 2 && v2 && a.dbg) hipLaunchKernelGGL((lstm_rec_bwd2_kernel<256, 2, true>), dim3(grid), dim3(576), 0, s, a);
        else if (R == 1) hipLaunchKernelGGL((lstm_rec_bwd_ag_kernel<H, 1>), dim3(grid), dim3(512), 0, s, a);
        else if (H == 256 && a.dbg) hipLaunchKernelGGL((lstm_rec_bwd_ag_kernel<256, 2, true>), dim3(grid), dim3(512), 0, s, a);
        else hipLaunchKernelGGL((lstm_rec_bwd_ag_kernel<H, 2>), dim3(grid), dim3(512), 0, s, a);
        return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
    }
    switch (R) {
        case 1: hipLaunchKernelGGL((lstm_rec_bwd_kernel<H, 1>), dim3(grid), dim3(512), 0, s, a); break;
        case 2: hipLaunchKernelGGL((lstm_rec_bwd_kernel<H, 2>), dim3(grid), dim3(512), 0, s, a); break;
        case 4:
            if constexpr (H == 512) return ASR_EUNSUPPORTED;       // (spilling instantiations: never selected, see asr_lstm_layer_bwd_p3)
            else hipLaunchKernelGGL((lstm_rec_bwd_kernel<H, 4>), dim3(grid), dim3(512), 0, s, a);
            break;
        case 8:
            if constexpr (H == 512) return ASR_EUNSUPPORTED;
            else hipLaunchKernelGGL((lstm_rec_bwd_kernel<H, 8>), dim3(grid), dim3(512), 0, s, a);
            break;
        default: return ASR_EINVAL;
    }
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

}  // namespace asr

extern "C" int asr_gemm_f32_batched(void* stream, int transA, int transB, int M, int N, int K,
                                    const float* A, int lda, long long strideA, const float* B, int ldb, long long strideB,
                                    float* C, int ldc, long long strideC, const float* bias, int accumulate, int batch);
extern "C" int asr_gemm_f32(void*, int, int, int, int, int, const float*, int, const float*, int,
                            float*, int, const float*, int);
extern "C" int asr_colsum_f32(void* stream, const float* x, int ldx, int M, int N, float* out, int accumulate);
int asr_lstm_pick_rows(int B, int ND, int G);
int asr_lstm_max_wgs();

static size_t lstm_bwd_hx_bytes(int B, int H, int ndir) {
    const size_t G = H / 32 < 4 ? 4 : H / 32;    // reduce-scatter: 2*G*H granules per row; all-gather: 2*4H
    return (size_t)ndir * (size_t)((B + 7) / 8 * 8) * 2 * G * H * sizeof(u64);
}
static size_t lstm_bwd_sync_bytes(int B, int H, int ndir) {
    return lstm_bwd_hx_bytes(B, H, ndir) + (size_t)ndir * (size_t)((B + 7) / 8 * 8) * 16 * sizeof(u64);   // granules + XCC slots
}
extern "C" size_t asr_lstm_bwd_ws_bytes(int B, int H, int ndir) {
    return lstm_bwd_sync_bytes(B, H, ndir) + (size_t)B * ndir * 4 * H * sizeof(float);      // + bias-gradient partials [B][ND][4H]
}

// Backward of asr_lstm_layer_fwd.  act/hprev are the forward's saved tensors; gates (the
// forward's input-projection buffer) is overwritten with dG.  dx [B,T,in] (may be NULL for the first layer) receives the input
// gradient; dkernel_*/dbias_* are ACCUMULATED into (TF layout [in+H,4H] / [4H]).
int asr_colsum_pair_f32(hipStream_t s, const float* x, int ldx, int M, int N, float* out0, float* out1);
struct asr_lstm_p3 {            // include/e2e_asr_hip.h
    int np; const void* x_p3; int x_cols; const void* kxT_p3; void* out_p3; void* hprev_p3;
    void* dg_p3; const void* kxu_p3; const int* colmap;
};
extern "C" int asr_gemm_p3_kk(void* stream, int M, int N, int K, const void* A, int lda8, const void* B, int ldb8, int np,
                              float* C, int ldc, const float* bias, int accumulate, int splits);
bool asr_lstm_g4_selected(int B, int H, int ndir);
extern "C" int asr_lstm_layer_bwd_p3(void* stream, const float* x, int B, int T, int in_dim, int ldx,
                                  const int* len, int H, int ndir,
                                  const float* kernel_fw, const float* kernel_bw,
                                  const float* dout, int Tout, float* gates, const float* act,
                                  const float* hprev, float* dx,
                                  float* dkernel_fw, float* dbias_fw, float* dkernel_bw, float* dbias_bw,
                                  void* hx_ws, size_t hx_bytes, int* err_flag, float keep_prob, unsigned seed,
                                  const float* kx_cat, const asr_lstm_p3* p3);
extern "C" int asr_lstm_layer_bwd(void* stream, const float* x, int B, int T, int in_dim, int ldx,
                                  const int* len, int H, int ndir,
                                  const float* kernel_fw, const float* kernel_bw,
                                  const float* dout, int Tout, float* gates, const float* act,
                                  const float* hprev, float* dx,
                                  float* dkernel_fw, float* dbias_fw, float* dkernel_bw, float* dbias_bw,
                                  void* hx_ws, size_t hx_bytes, int* err_flag, float keep_prob, unsigned seed,
                                  const float* kx_cat) {
    return asr_lstm_layer_bwd_p3(stream, x, B, T, in_dim, ldx, len, H, ndir, kernel_fw, kernel_bw, dout, Tout, gates, act, hprev, dx,
                                 dkernel_fw, dbias_fw, dkernel_bw, dbias_bw, hx_ws, hx_bytes, err_flag, keep_prob, seed, kx_cat, nullptr);
}
// With p3 (asr_lstm_p3_supported shapes): the BPTT writes dG as bf16 planes (p3->dg_p3, unit-major columns) and the products
// that consume it run on plane operands (csrc/gemm_p3.hip): dX = dG . K_x^T with p3->kxu_p3, and every weight gradient of the
// layer in ONE launch from p3->x_p3 / p3->hprev_p3 (the forward wrote them).  `gates` / `hprev` (fp32) are then not touched.
extern "C" int asr_lstm_layer_bwd_p3(void* stream, const float* x, int B, int T, int in_dim, int ldx,
                                  const int* len, int H, int ndir,
                                  const float* kernel_fw, const float* kernel_bw,
                                  const float* dout, int Tout, float* gates, const float* act,
                                  const float* hprev, float* dx,
                                  float* dkernel_fw, float* dbias_fw, float* dkernel_bw, float* dbias_bw,
                                  void* hx_ws, size_t hx_bytes, int* err_flag, float keep_prob, unsigned seed,
                                  const float* kx_cat, const asr_lstm_p3* p3) {
    using namespace asr;
    const bool p3_dg = p3 && p3->dg_p3;
    const bool p3_dx = p3_dg && (!dx || (p3->kxu_p3 && in_dim % 256 == 0 && (B * T) % 128 == 0));
    const bool p3_wg = p3_dg && p3->x_p3 && p3->hprev_p3 && p3->colmap && p3->x_cols % 128 == 0 && p3->x_cols >= in_dim && ndir == 2 &&
                       dkernel_bw && (B * T) % 16 == 0;
    if (p3_dg && !(p3_dx && p3_wg)) return ASR_EUNSUPPORTED;        // (all consumers of dG or none: the fp32 dG is not written)
    if (!x || !len || !kernel_fw || !dout || !gates || !act || (!hprev && !p3_wg) || !dkernel_fw || !dbias_fw || !hx_ws || !err_flag)
        return ASR_EINVAL;
    if (ndir != 1 && ndir != 2) return ASR_EINVAL;
    if (ndir == 2 && (!kernel_bw || !dkernel_bw || !dbias_bw)) return ASR_EINVAL;
    if (H != 64 && H != 128 && H != 256 && H != 512) return ASR_EUNSUPPORTED;
    const bool prezeroed = (hx_bytes & ASR_WS_PREZEROED) != 0;
    hx_bytes &= ~ASR_WS_PREZEROED;
    if (hx_bytes < asr_lstm_bwd_ws_bytes(B, H, ndir)) return ASR_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int H4 = 4 * H, G = H / 32;
    LstmBwdArgs a;
    a.gates = gates; a.act = act; a.dout = dout;
    a.kh[0] = kernel_fw + (size_t)in_dim * H4;
    a.kh[1] = ndir == 2 ? kernel_bw + (size_t)in_dim * H4 : nullptr;
    a.len = len; a.hx = static_cast<u64*>(hx_ws); a.err = err_flag;
    a.xcc_slots = reinterpret_cast<u64*>(static_cast<char*>(hx_ws) + lstm_bwd_hx_bytes(B, H, ndir));
    a.B = B; a.T = T; a.Tout = Tout; a.ND = ndir; a.boff = 0; a.keep = keep_prob; a.seed = seed;
    a.sb = T; a.st = 1; a.osb = Tout; a.ost = 1; a.ldo = ndir * H; a.dsb = Tout; a.dst = 1;
    a.dbg = getenv("ASR_LSTM_STAMP") ? asr::g_lstm_dbg : nullptr;
    a.dg_p3 = p3_dg ? static_cast<char*>(p3->dg_p3) : nullptr; a.p3_np = p3_dg ? p3->np : 0; a.dg_f32 = p3_dg ? 0 : 1;
    a.act_c = nullptr;
    int R = asr_lstm_pick_rows(B, ndir, G);
    // H = 512: the reduce-scatter kernel for four or eight rows per group spills (256 registers + 200-768 bytes of scratch: 9.6 us
    // per step at B = 32); two rows per group through the all-gather kernel in twice the launches are 4.0 us per step of a layer
    if (H == 512 && R > 2) R = 2;
    // the all-gather kernel (R <= 2) sums dG over time per utterance itself: the bias gradient is then a colsum over B rows
    // instead of over B*T rows of dG (0.4 GB of side-stream reads per step at config 2)
    static const bool ag_env = [] { const char* e = getenv("ASR_BPTT_AG"); return !(e && e[0] == '0'); }();
    float* db_part = (ag_env && R <= 2) ? reinterpret_cast<float*>(static_cast<char*>(hx_ws) + lstm_bwd_sync_bytes(B, H, ndir)) : nullptr;
    a.db_part = db_part;
    const int max_groups = asr_lstm_max_wgs() / G / ndir;
    if (max_groups < 1) return ASR_EUNSUPPORTED;      // one group (both directions) cannot be co-resident on this device
    const int rows_per_launch = max_groups * R;
    // groups of four workgroups, one row per group (lstm_rec_bwd4_kernel), when the whole batch is resident at once
    // (ASR_LSTM_G4=0: the eight-workgroup groups of version 2)
    // groups of four workgroups, one row per group (lstm_rec_bwd4_kernel): the same predicate as the forward's (csrc/lstm.hip) --
    // the pair shares the 20-byte split record format
    const bool g4 = asr_lstm_g4_selected(B, H, ndir);
    if (p3_dg && !g4) return ASR_EUNSUPPORTED;            // only the groups-of-four BPTT writes planes
    hipEvent_t e_bptt_done = nullptr;                     // stop event of the (single) BPTT launch, or nullptr
    if (g4) {
        const int rpl4 = asr_lstm_max_wgs() / (4 * ndir);
        if (!a.db_part) a.db_part = reinterpret_cast<float*>(static_cast<char*>(hx_ws) + lstm_bwd_sync_bytes(B, H, ndir));
        db_part = a.db_part;
        for (int b0 = 0; b0 < B; b0 += rpl4) {
            if (!(prezeroed && b0 == 0) && hipMemsetAsync(hx_ws, 0, lstm_bwd_sync_bytes(B, H, ndir), s) != hipSuccess) return ASR_ELAUNCH;
            LstmBwdArgs c = a;
            c.B = (B - b0 < rpl4) ? (B - b0) : rpl4;
            c.boff = b0;
            c.gates = gates + (size_t)b0 * T * ndir * H4;
            if (a.dg_p3) c.dg_p3 = a.dg_p3 + (size_t)b0 * T * (ndir * H4 / 8) * 16 * a.p3_np;
            c.act = act + (size_t)b0 * T * ndir * H * 4;                       // gates plane, then the c plane (csrc/lstm.hip)
            c.act_c = act + (size_t)B * T * ndir * H * 4 + (size_t)b0 * T * ndir * H;
            c.dout = dout + (size_t)b0 * Tout * ndir * H;
            c.len = len + b0;
            const int groups = ndir * c.B;
            const int padded = ((groups + 7) & ~7) * 4;
            const int grid = padded <= asr_lstm_max_wgs() ? padded : groups * 4;
            // start / stop events of the dispatch itself (hipExtLaunchKernelGGL): the timing events of bench.py's roofline leg, and
            // -- single launch -- the event the side stream's weight gradients wait on, without marker packets on this stream
            // (a recorded event costs the stream ~3 us, ~6 with a waiter: scripts/micro/fork_cost.hip; ASR_EXT_EVENTS=0: records)
            static const bool ext_ev = [] { const char* e = getenv("ASR_EXT_EVENTS"); return !(e && e[0] == '0'); }();
            hipEvent_t ev_a = nullptr, ev_b = nullptr;
            if (ext_ev) prof_launch_events(ASR_PROF_LSTM_REC_BWD, &ev_a, &ev_b);
            else prof_begin(ASR_PROF_LSTM_REC_BWD, s);
            if (ext_ev && !ev_b && B <= rpl4) ev_b = next_event();
            static const bool quad = [] { const char* e = getenv("ASR_BPTT_QUAD"); return !(e && e[0] == '0'); }();
            if (quad) hipExtLaunchKernelGGL((asr::lstm_rec_bwd4_kernel<false, true>), dim3(grid), dim3(512), 0, s, ev_a, ev_b, 0, c);
            else hipExtLaunchKernelGGL((asr::lstm_rec_bwd4_kernel<false, false>), dim3(grid), dim3(512), 0, s, ev_a, ev_b, 0, c);
            if (!ext_ev) prof_end(ASR_PROF_LSTM_REC_BWD, s);
            if (hipGetLastError() != hipSuccess) return ASR_ELAUNCH;
            if (ext_ev && B <= rpl4) e_bptt_done = ev_b;
        }
    } else
    for (int b0 = 0; b0 < B; b0 += rows_per_launch) {
        if (!(prezeroed && b0 == 0) && hipMemsetAsync(hx_ws, 0, lstm_bwd_sync_bytes(B, H, ndir), s) != hipSuccess) return ASR_ELAUNCH;
        LstmBwdArgs c = a;
        c.B = (B - b0 < rows_per_launch) ? (B - b0) : rows_per_launch;
        c.boff = b0;
        c.gates = gates + (size_t)b0 * T * ndir * H4;
        c.act = act + (size_t)b0 * T * ndir * H * 8;
        c.dout = dout + (size_t)b0 * Tout * ndir * H;
        c.len = len + b0;
        int rc;
        prof_begin(ASR_PROF_LSTM_REC_BWD, s);
        switch (H) {
            case 64: rc = launch_bwd_h<64>(s, c, R); break;
            case 128: rc = launch_bwd_h<128>(s, c, R); break;
            case 256: rc = launch_bwd_h<256>(s, c, R); break;
            default: rc = launch_bwd_h<512>(s, c, R); break;
        }
        prof_end(ASR_PROF_LSTM_REC_BWD, s);
        if (rc) return rc;
    }
    // bias gradient from the per-utterance partials: B rows, on the caller's stream (the workspace is reused by the next
    // layer's BPTT, so it must not wait in the side stream's queue)
    if (db_part && ndir == 2) {           // both directions in one launch
        int rc;
        if ((rc = asr_colsum_pair_f32(s, db_part, 2 * H4, B, H4, dbias_fw, dbias_bw))) return rc;
    } else
    for (int d = 0; d < ndir && db_part; ++d) {
        int rc;
        if ((rc = asr_colsum_f32(stream, db_part + (size_t)d * H4, ndir * H4, B, H4, d ? dbias_bw : dbias_fw, 1))) return rc;
    }
    // Input gradient dX = dG.K_x^T stays on the caller's stream (the next layer's BPTT needs it).
    // The weight/bias gradients (dK_x = X^T.dG, dK_h = Hprev^T.dG, db = colsum dG) are needed only
    // by the optimizer: they go to the library's side stream and overlap the next layer's BPTT,
    // whose workgroups mostly wait on the exchange and leave the matrix pipes idle.
    // asr_side_join() orders them before the gradients are consumed.
    const int M = B * T;
    if (p3_dg) {
        int rc;
        if (dx && (rc = asr_gemm_p3_kk(stream, M, in_dim, ndir * H4, p3->dg_p3, ndir * H4 / 8, p3->kxu_p3, ndir * H4 / 8, p3->np,
                                       dx, in_dim, nullptr, 0, 1))) return rc;
        static const bool wg_inline_p = [] { const char* e = getenv("ASR_WGRAD_INLINE"); return e && e[0] == '1'; }();
        hipStream_t ssp = wg_inline_p ? s : side_stream();
        // the weight gradients read dG (and x, h_prev of the forward): they wait for the BPTT, not for the dX product above
        if (!wg_inline_p && e_bptt_done) { if (hipStreamWaitEvent(ssp, e_bptt_done, 0) != hipSuccess) return ASR_ELAUNCH; }
        else {
            hipEvent_t e_dgp = next_event();
            if (!wg_inline_p && (hipEventRecord(e_dgp, s) != hipSuccess || hipStreamWaitEvent(ssp, e_dgp, 0) != hipSuccess)) return ASR_ELAUNCH;
        }
        if ((rc = p3_lstm_wgrad(ssp, M, p3->x_cols, in_dim, H, ndir, p3->x_p3, p3->x_cols / 8, p3->hprev_p3, p3->dg_p3, p3->np,
                                dkernel_fw, (long long)(dkernel_bw - dkernel_fw), p3->colmap))) return rc;
        hipEvent_t e_donep = next_event();
        if (hipEventRecord(e_donep, ssp) != hipSuccess) return ASR_ELAUNCH;
        set_pending_join(e_donep);
        return ASR_OK;
    }
    if (dx && ndir == 2 && kx_cat) {
        // dX = [dG_fw | dG_bw] . [K_x,fw | K_x,bw]^T as ONE product with K = 8H (dG rows are contiguous over the two
        // directions; kx_cat is the [in, 8H] array the forward used): no second accumulating pass over dX
        int rc;
        if ((rc = asr_gemm_f32(stream, 0, 1, M, in_dim, 2 * H4, gates, 2 * H4, kx_cat, 2 * H4, dx, in_dim, nullptr, 0))) return rc;
    } else
    for (int d = 0; d < ndir && dx; ++d) {
        int rc;
        if ((rc = asr_gemm_f32(stream, 0, 1, M, in_dim, H4, gates + (size_t)d * H4, ndir * H4, d ? kernel_bw : kernel_fw, H4,
                               dx, in_dim, nullptr, d > 0))) return rc;
    }
    // (ASR_WGRAD_INLINE=1, experiment: the weight gradients on the caller's stream, nothing next to the next layer's BPTT)
    static const bool wg_inline = [] { const char* e = getenv("ASR_WGRAD_INLINE"); return e && e[0] == '1'; }();
    hipStream_t ss = wg_inline ? s : side_stream();
    void* side = static_cast<void*>(ss);
    if (!wg_inline && e_bptt_done) { if (hipStreamWaitEvent(ss, e_bptt_done, 0) != hipSuccess) return ASR_ELAUNCH; }
    else {
        hipEvent_t e_dg = next_event();
        if (!wg_inline && (hipEventRecord(e_dg, s) != hipSuccess || hipStreamWaitEvent(ss, e_dg, 0) != hipSuccess)) return ASR_ELAUNCH;
    }
    // both directions of a product as ONE batched launch when the two gradient buffers sit a vector-aligned stride apart
    // (they do in the flat gradient buffer): X is then streamed once for the two dK_x instead of twice, and a launch has
    // twice the tiles (less split-K, fewer atomics)
    const long long dks = ndir == 2 ? (long long)(dkernel_bw - dkernel_fw) : 0;
    const bool fuse_dirs = ndir == 2 && dks % 4 == 0;
    if (fuse_dirs) {
        int rc;
        if ((rc = asr_gemm_f32_batched(side, 1, 0, in_dim, H4, M, x, ldx, 0, gates, ndir * H4, H4, dkernel_fw, H4, dks, nullptr, 1, 2)))
            return rc;
        if ((rc = asr_gemm_f32_batched(side, 1, 0, H, H4, M, hprev, ndir * H, H, gates, ndir * H4, H4,
                                       dkernel_fw + (size_t)in_dim * H4, H4, dks, nullptr, 1, 2))) return rc;
    }
    for (int d = 0; d < ndir; ++d) {
        const float* dG = gates + (size_t)d * H4;
        const int ldg = ndir * H4;
        float* dK = d ? dkernel_bw : dkernel_fw;
        float* dB = d ? dbias_bw : dbias_fw;
        int rc;
        if (!fuse_dirs) {
            if ((rc = asr_gemm_f32(side, 1, 0, in_dim, H4, M, x, ldx, dG, ldg, dK, H4, nullptr, 1))) return rc;
            if ((rc = asr_gemm_f32(side, 1, 0, H, H4, M, hprev + (size_t)d * H, ndir * H, dG, ldg,
                                   dK + (size_t)in_dim * H4, H4, nullptr, 1))) return rc;
        }
        if (!db_part && (rc = asr_colsum_f32(side, dG, ldg, M, H4, dB, 1))) return rc;
    }
    hipEvent_t e_done = next_event();
    if (hipEventRecord(e_done, ss) != hipSuccess) return ASR_ELAUNCH;
    set_pending_join(e_done);
    return ASR_OK;
}


// Time-major single-direction BPTT over all T steps (the decoder's LM cell chain; see asr_lstm_rec_fwd_tm):
// gates [T][B][4H] receives dG, act [T][B][H][8], dout rows t*B + b with leading dimension ldo.
bool asr_lstm_tm_supported(int B, int H);
int asr_lstm_rec_bwd_tm(hipStream_t s, float* gates, const float* act, const float* dout, int ldo, const float* kh,
                        const int* full_len, void* hx_ws, int* err, int B, int T, int H, float keep, unsigned seed) {
    using namespace asr;
    if (!asr_lstm_tm_supported(B, H)) return ASR_EUNSUPPORTED;
    if (hipMemsetAsync(hx_ws, 0, asr_lstm_bwd_ws_bytes(B, H, 1), s) != hipSuccess) return ASR_ELAUNCH;
    LstmBwdArgs a;
    a.gates = gates; a.act = act; a.dout = dout; a.kh[0] = kh; a.kh[1] = nullptr; a.len = full_len;
    a.hx = static_cast<u64*>(hx_ws); a.err = err;
    a.xcc_slots = reinterpret_cast<u64*>(static_cast<char*>(hx_ws) + lstm_bwd_hx_bytes(B, H, 1));
    a.B = B; a.T = T; a.Tout = T; a.ND = 1; a.boff = 0; a.keep = keep; a.seed = seed;
    a.sb = 1; a.st = B; a.osb = 1; a.ost = B; a.ldo = ldo; a.dsb = 1; a.dst = B;
    a.dbg = nullptr; a.db_part = nullptr;
    a.dg_p3 = nullptr; a.p3_np = 0; a.dg_f32 = 1; a.act_c = nullptr;
    // groups of four workgroups (lstm_rec_bwd4_kernel<REC32>): 4 B workgroups instead of 8 B -- at B = 32 half the chip, and the
    // faster step (1.15 vs 1.3 us).  ASR_LM_G4=0: the eight-workgroup kernel as before round 5.
    static const bool lm_g4 = [] { const char* e = getenv("ASR_LM_G4"); return !(e && e[0] == '0'); }();
    if (lm_g4 && H == 256 && asr_lstm_g4_selected(B, H, 1) && 4 * B <= asr_lstm_max_wgs()) {
        const int padded = ((B + 7) & ~7) * 4;
        const int grid = padded <= asr_lstm_max_wgs() ? padded : B * 4;
        static const bool quad = [] { const char* e = getenv("ASR_BPTT_QUAD"); return !(e && e[0] == '0'); }();
        if (quad) hipLaunchKernelGGL((asr::lstm_rec_bwd4_kernel<true, true>), dim3(grid), dim3(512), 0, s, a);
        else hipLaunchKernelGGL((asr::lstm_rec_bwd4_kernel<true, false>), dim3(grid), dim3(512), 0, s, a);
        return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
    }
    int R = asr_lstm_pick_rows(B, 1, H / 32);
    if (H == 512 && R > 2) R = 2;          // (asr_lstm_tm_supported has checked that the batch fits)
    switch (H) {
        case 64: return launch_bwd_h<64>(s, a, R);
        case 128: return launch_bwd_h<128>(s, a, R);
        case 256: return launch_bwd_h<256>(s, a, R);
        default: return launch_bwd_h<512>(s, a, R);
    }
}
