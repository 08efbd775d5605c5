// Persistent attention decoder, inference graph (isTraining=False): ALL steps of the greedy decode loop
// (attn_decoder.py:76-162 with loop_function = argmax feedback, decoder.py:139-154; eval_model.py:56-118) in ONE launch.
//
// In the inference graph every step's input token is the argmax of the previous step's logits, so nothing can be
// hoisted out of the loop except products with constants: the whole step
//     tok_i -> LM cell -> [InputProjection folded] outer cell -> q = c_i -> y -> e -> alpha -> ctx_i
//           -> AttnProjection -> OutputProjection -> argmax -> tok_{i+1}
// is one dependency chain.  As separate launches that chain is seven kernels and two stream hand-overs per step
// (74 us per step at config 2, of which 49 us inside skinny kernels that re-read their weights from L2); here a
// step is eight granule exchanges inside a persistent kernel whose weights never leave the chip.
//
// Decomposition: groups of R = 4 utterances never talk to each other; a group is G = 32 workgroups -- ONE XCD, so every
// exchange stays in that XCD's L2 -- and 8 groups (32 utterances) fill the 256 CUs.  Each workgroup owns 1/32 of every
// weight matrix, in REGISTERS, cut by output column: 8 LM units (K_h of the LM cell, 256x32), 8 decoder units
// ([WK_P ; K_h ; WK_c], 1024x32, with InputProjection folded as in decoder_chain.hip), 4 attention columns of W_att,
// 8 AttnProjection columns (768x8), 32 vocabulary columns of OutputProjection (256x32); plus, in LDS, the hf rows of
// its <= 8 encoder positions and 16 context columns of the group's enc rows.  The embedding never enters a matvec:
// EK = embedding . K_x(LM) + b (one [V,4lmH] GEMM per call) turns "embed tok, multiply" into a row lookup, and the
// h-part of the LM gates for step i+1 is formed while step i's logits are still being reduced.
//
// Exchanges are all-gathers of tagged 8-byte granules (granule.h); wave 0 is the cell/publisher wave (never polls),
// waves 1-6 poll, wave 7 writes the logits out.  Activations are not saved: this is the inference graph.
#include "common.h"
#include "granule.h"
#include <algorithm>
#include <cstdlib>

namespace asr { extern unsigned long long* g_lstm_dbg; }

namespace asr {

struct GreedyArgs {
    const float* ek;         // [V][4*LMH]  embedding . K_x(LM) + b_lm
    const float* lm_kh;      // [LMH][4*LMH] recurrent rows of the LM cell kernel
    const float* wk;         // [LMH + D][4H] W_inp . K_x (rows: LM output | context)
    const float* bprime;     // [4H] b_inp . K_x + b_dec
    const float* dec_kh;     // [H][4H] recurrent rows of the outer cell kernel
    const float* w_att; const float* b_att; const float* v;       // [H][A], [A], [A]
    const float* ap_w; const float* ap_b;                         // [H+D][H], [H]
    const float* out_w; const float* out_b;                       // [H][V], [V]
    const float* hf;         // [B][Te][A]
    const float* enc;        // [B][Te][D]
    const int* enc_len;      // [B]
    const int* seq_len;      // [B]  rows emit zeros from step seq_len[b] on
    int* tok;                // [T][B] in: row 0 (GO); out: rows 1.. = argmax of the previous step
    float* logits;           // [T][B][V]
    u64* gx;                 // granules [groups][2 parities][NPAR]
    u64* xcc_slots;          // [groups][32]
    int* err;
    int B, Te, T, V, g0, ng;
    unsigned long long* dbg;    // STAMP build only
    // ---- training graph (TRAIN instantiation, attn_decoder.py:126-145): tok[] holds the teacher token of EVERY step; the
    // token of step i+1 is drawn from softmax(logits_i) (Gumbel-max, decoder.py:156-180) where bit i of fbmask is set (the
    // host's scheduled-sampling coins); AttnProjection / OutputProjection run in the loop only at those steps (the logits of
    // all steps come from hoisted GEMMs afterwards); DropoutWrapper on the LM output; activations saved for the backward
    // in the layouts of decoder_chain.hip / lstm.hip (time-major rows i*B + b).
    uint32_t fbmask[8];
    float keep; uint32_t seed;
    float* lm_out; float* lm_hprev; float* lm_act;       // [T][B][LMH] (dropped) output, [T][B][LMH] h_{i-1}, [T][B][LMH][8] records
    float* dec_gates; float* dec_c; float* dec_h;        // [T][B][4H] activated i,j,f,o; [T][B][H]; [T][B][H]
    float* y; float* alpha; float* ctx;                  // [T][B][A], [T][B][Te], [T][B][D]
};

// (value, index) argmax combine with first-max tie-breaking (np.argmax / tf.argmax); NaN never wins
__device__ __forceinline__ void amax_take(float& bv, int& bi, float ov, int oi) {
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
}

// STAMP: diagnostic instantiation (ASR_CHAIN_STAMP=1 + asr_debug_set_buffer): s_memtime totals of wave 0 of workgroup 0 per
// phase (code up to each barrier of a step, in program order) -> dbg[48..]; never used for timing claims.
// (max, lowest index) over the 16 lanes of a DPP row: the butterflies of row16_allreduce_sum on (value, index) pairs -- every
// lane ends with the row's result (four ds_bpermute shuffles per stage cost ~1600 cycles per reduction, twice per step)
template <int CTRL>
__device__ __forceinline__ void amax_dpp(float& bv, int& bi) {
    const float ov = dpp_mov<CTRL>(bv);
    const int oi = __builtin_amdgcn_update_dpp(0, bi, CTRL, 0xF, 0xF, true);
    amax_take(bv, bi, ov, oi);
}
__device__ __forceinline__ void row16_argmax(float& bv, int& bi) {
    amax_dpp<0xB1>(bv, bi); amax_dpp<0x4E>(bv, bi); amax_dpp<0x141>(bv, bi); amax_dpp<0x140>(bv, bi);
}

// MAXTS = encoder positions per workgroup: 8 (Te <= 256: the hf rows of the slice live in LDS) or 16 (Te <= 512, the phone decoder
// of BASELINE config 4 on encoder depth 2: scores in two passes of 8 positions with the hf rows read from L2 every step -- 26 KB
// per workgroup -- because LDS holds the 102 KB of context columns; everything else is the same step).
template <int H, int D, int A, int LMH, bool STAMP = false, bool TRAIN = false, int MAXTS = 8>
__global__ __launch_bounds__(512) void decoder_greedy_kernel(GreedyArgs a) {
    static_assert(MAXTS == 8 || MAXTS == 16, "positions per workgroup");
    unsigned int stamp[24] = {0};
    unsigned long long tlast = 0;
    int sph = 0;
#define GREEDY_STAMP() if (STAMP) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); stamp[sph < 23 ? sph : 23] += (unsigned int)(t__ - tlast); tlast = t__; ++sph; }
    constexpr int R = 4, G = 32, NT = 512, VS = 32;
    constexpr bool HFL = MAXTS == 8;             // hf rows of my positions in LDS
    constexpr int HS = H / G, LS = LMH / G, AS = A / G, DS = D / G, PS = H / G;
    constexpr int KD = LMH + H + D;              // outer cell input [lm_out | h | ctx]
    constexpr int KA = H + D;                    // AttnProjection input [q | ctx]
    constexpr int H4 = 4 * H, L4 = 4 * LMH;
    constexpr int AL = A / 16;
    static_assert(HS == 8 && LS == 8 && PS == 8 && AS == 4 && DS == 16, "thread maps below are cut for 8/8/8/4/16 slices");
    static_assert(LMH == 256 && H == 256 && KD == 1024 && KA == 768 && AL % 4 == 0, "K ranges: parts of 64 / 256 / 192 / 256 values");
    constexpr int NPOLL = NT - 128;              // waves 1..6

    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* lds_flag = reinterpret_cast<int*>(smem);
    float* v_dec = smem + 4;                     // [R][KD]   [lm_out_i | h_{i-1} | ctx_{i-1}]
    float* v_ap = v_dec + R * KD;                // [R][KA]   [q_i | ctx_i]
    float* v_p = v_ap + R * KA;                  // [R][H]    AttnProjection output
    float* yl = v_p + R * H;                     // [R][A]
    float* el = yl + R * A;                      // [R][G*MAXTS] scores / alpha
    float* sums = el + R * G * MAXTS;            // [4 parts][8 units][R][4 gates]  outer cell
    float* lmsum = sums + 4 * 8 * R * 4;         // same, LM cell (h-part, formed one step ahead)
    float* ysum = lmsum + 4 * 8 * R * 4;         // [4][AS][R]
    float* psum = ysum + 4 * AS * R;             // [4][PS][R]
    float* cpart = psum + 4 * PS * R;            // [8][R][DS]
    float* eout = cpart + 8 * R * DS;            // [R * MAXTS]
    float* lg = eout + R * MAXTS;                // [R][VS] logits of my vocabulary slice
    float* mv = lg + R * VS;                     // [R][G] partial maxima, then [R][G] their indices (int)
    int* mi = reinterpret_cast<int*>(mv + R * G);
    float* vl = mv + 2 * R * G;                  // [A]
    float* hfl = vl + A;                         // [R][MAXTS][A]  (HFL)
    const int Te = a.Te, V = a.V;
    const int TS = (Te + G - 1) / G;
    float* v_lmh = hfl + (HFL ? R * MAXTS * A : 0);   // [R][LMH]  TRAIN: undropped h_lm_i (the LM's own recurrence input)
    float* encl = v_lmh + (TRAIN ? R * LMH : 0); // [R][Te][DS]

    const int grp_l = blockIdx.x & 7, mem = blockIdx.x >> 3;       // round-robin dispatch: a group = the 32 workgroups of one XCD
    if (grp_l >= a.ng) return;
    __builtin_amdgcn_s_setprio(3);
    const int grp = a.g0 + grp_l;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane & 15, row = tid >> 4;
    const int u8 = row & 7, part = row >> 3;
    const int r0 = grp * R;
    const bool wave0 = __builtin_amdgcn_readfirstlane(tid) < 64;
    const bool wave7 = __builtin_amdgcn_readfirstlane(tid) >= NT - 64;
    const bool poller = !wave0 && !wave7;
    auto rok = [&](int r) { return r0 + r < a.B; };
    auto browf = [&](int r) { return min(r0 + r, a.B - 1); };
    int blen[R];
#pragma unroll
    for (int r = 0; r < R; ++r) blen[r] = rok(r) ? min(max(a.enc_len[browf(r)], 0), Te) : 0;
    constexpr int NLM = (TRAIN ? 2 : 1) * R * LMH, NQH = 2 * R * H, NY = R * A, NE = R * G * MAXTS, NC = R * D, NP = R * H, NM = 2 * R * G;
    constexpr int NPAR = NLM + NQH + NY + NE + NC + NP + NM;
    u64* gbase = a.gx + (size_t)grp * 2 * NPAR;
    const bool fast = group_shares_xcd(a.xcc_slots + (size_t)grp * G, G, mem, tid, a.err, lds_flag, 0, 3);

    // ---- resident weights (registers), lane kq of a DPP row <-> float4 j of a K part at k = base + (16 j + kq) * 4
    // (gates (i, j) and (f, o) of one K row as register PAIRS: the two matvecs below run on v_pk_fma_f32 -- two fp32 FMAs per
    //  issue slot, bitwise two v_fma_f32 -- with the h / input value broadcast to both halves; round 3, as csrc/lstm.hip)
    f32x2 wlm[4][2];          // LM cell, recurrent part: K = LMH in 4 parts of 64
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = part * 64 + kq * 4 + e;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2)
            wlm[e][g2] = f32x2{a.lm_kh[(size_t)k * L4 + (2 * g2) * LMH + mem * LS + u8], a.lm_kh[(size_t)k * L4 + (2 * g2 + 1) * LMH + mem * LS + u8]};
    }
    f32x2 wdec[16][2];        // outer cell: K = KD in 4 parts of 256
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = part * 256 + (j * 16 + kq) * 4 + e;
            const float* wr = (k < LMH) ? a.wk + (size_t)k * H4 : (k < LMH + H ? a.dec_kh + (size_t)(k - LMH) * H4 : a.wk + (size_t)(k - H) * H4);
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) wdec[j * 4 + e][g2] = f32x2{wr[(2 * g2) * H + mem * HS + u8], wr[(2 * g2 + 1) * H + mem * HS + u8]};
        }
    const int ycol = row & 3, ypart = (row >> 2) & 3;
    const bool yact = row < 16;
    float wy[4];              // y = q . W_att: K = H in 4 parts of 64 (DPP rows 0..15)
#pragma unroll
    for (int e = 0; e < 4; ++e) wy[e] = yact ? a.w_att[(size_t)(ypart * 64 + kq * 4 + e) * A + mem * AS + ycol] : 0.f;
    // AttnProjection (K = KA in 4 parts of 192) and OutputProjection (one vocabulary column per DPP row, K = H) slices.  The
    // inference graph needs them at every step: registers.  The training graph runs the projections at feedback steps only
    // (~10 % of the steps): it re-reads the 28 values from L2 there and leaves the registers to the rest of the step.
    const int vcol = mem * VS + row;
    // (`z` is an opaque zero: inside the step loop it keeps the compiler from hoisting the loads back out of it)
    auto load_wap = [&](float* w, int z) {
        const float* base = a.ap_w + z;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) w[j * 4 + e] = base[(size_t)(part * 192 + (j * 16 + kq) * 4 + e) * H + mem * PS + u8];
    };
    auto load_wout = [&](float* w, int z) {
        const float* base = a.out_w + z;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) w[j * 4 + e] = vcol < V ? base[(size_t)((j * 16 + kq) * 4 + e) * V + vcol] : 0.f;
    };
    float wap_r[12], wout_r[16];
    if (!TRAIN) { load_wap(wap_r, 0); load_wout(wout_r, 0); }
    const float outb = vcol < V ? a.out_b[vcol] : 0.f;

    // ---- resident activations (LDS)
    const int tau0 = mem * TS;
    if (HFL)
    for (int idx = tid; idx < R * MAXTS * A; idx += NT) {
        const int r = idx / (MAXTS * A), rem = idx % (MAXTS * A), tl = rem / A, aa = rem % A, tau = tau0 + tl;
        hfl[idx] = (tl < TS && tau < Te) ? a.hf[((size_t)browf(r) * Te + tau) * A + aa] : 0.f;
    }
    for (int idx = tid; idx < R * Te * DS; idx += NT) {
        const int r = idx / (Te * DS), rem = idx % (Te * DS), tau = rem / DS, dd = rem % DS;
        encl[idx] = a.enc[((size_t)browf(r) * Te + tau) * D + mem * DS + dd];
    }
    for (int idx = tid; idx < A; idx += NT) vl[idx] = a.v[idx];
    for (int idx = tid; idx < R * KD; idx += NT) v_dec[idx] = 0.f;
    for (int idx = tid; idx < R * KA; idx += NT) v_ap[idx] = 0.f;
    for (int idx = tid; idx < 4 * 8 * R * 4; idx += NT) lmsum[idx] = 0.f;
    // cell threads (wave 0): r = tid / 8, unit = tid % 8, for both cells
    const bool cell = tid < R * 8;
    const int cr = cell ? tid >> 3 : 0, cu = tid & 7;
    const bool cb_ok = cell && rok(cr);
    const int cb = browf(cr);
    float c_lm = 0.f, c_dec = 0.f, h_lm_prev = 0.f;
    int tokr = cb_ok ? a.tok[cb] : 0;
    tokr = min(max(tokr, 0), V - 1);
    float bp[4], ybias = 0.f, pbias = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) bp[g] = cell ? a.bprime[g * H + mem * HS + cu] : 0.f;
    if (tid < R * AS) ybias = a.b_att[mem * AS + (tid & 3)];
    if (cell) pbias = a.ap_b[mem * PS + cu];
    // emit lengths of the rows a lane handles in the argmax (wave 0: r = lane / 16) and logits write-out (wave 7: r = lane / 32, + 2)
    const int slen0 = a.seq_len[browf((lane >> 4) & 3)];
    const int slen7a = a.seq_len[browf((lane >> 5) & 1)], slen7b = a.seq_len[browf(2 + ((lane >> 5) & 1))];
    __syncthreads();

    if (STAMP) tlast = __builtin_amdgcn_s_memtime();
    bool lm_ready = false;      // uniform: this step's LM output is already in LDS (gathered during the previous step)
    // MAXTS = 16: the hf rows of this thread's two positions (scores phase: DPP row -> (utterance, position % 8), lane kq -> two
    // float4 chunks) do not change from step to step: 16 registers, loaded once (rounds 4 - 5 re-read them from L2 in every step,
    // 26 KB per workgroup and an L2 round trip in front of the tanh)
    float4 hreg[2][2];
    if constexpr (!HFL) {
        const int r = row / 8;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int tl = (row % 8) + 8 * ps, tau = min(tau0 + min(tl, TS - 1), Te - 1);
            const float* hrow = a.hf + ((size_t)browf(r) * Te + tau) * A;
#pragma unroll
            for (int c = 0; c < 2; ++c) hreg[ps][c] = *reinterpret_cast<const float4*>(hrow + c * 64 + kq * 4);
        }
    }
    const int tid_outer = tid;
    int nfb = 0;                // feedback steps so far (uniform): numbers the exchange of p, which only those steps make
    for (int i = 0; i < a.T; ++i) {
        sph = 0;
        // Everything derived from the thread index is derived AGAIN in every step, from a copy the compiler cannot see through.
        // Hoisted out of the loop, the LDS and global addresses built from these indices were, in the training instantiation, 33
        // registers more than the file holds, and their reloads from scratch sat, each behind a vmcnt(0), right in front of the
        // publishing stores and the gathers' LDS writes of the critical path (256 registers + 132 bytes of scratch -> 220 and
        // none; decoder forward 1.27 -> 1.23 ms, inference 1.36 -> 1.33).  Re-deriving costs a few VALU operations.
        int tz = 0;
        asm volatile("v_mov_b32 %0, 0" : "=v"(tz));
        const int tid = tid_outer + tz, lane = tid & 63;
        const int kq = lane & 15, row = tid >> 4;
        const int u8 = row & 7, part = row >> 3;
        const int ycol = row & 3, ypart = (row >> 2) & 3;
        const bool yact = row < 16;
        const int vcol = mem * VS + row;
        const bool cell = tid < R * 8;
        const int cr = cell ? tid >> 3 : 0, cu = tid & 7;
        const bool cb_ok = cell && rok(cr);
        const int cb = browf(cr);
        const uint32_t ep = (uint32_t)(i + 1);
        const uint32_t tb = tag_bit(i);
        // does step i feed its own prediction forward?  inference graph: always; training graph: at the flagged steps
        const bool fbi = !TRAIN || (i + 1 < a.T && ((a.fbmask[(i >> 5) & 7] >> (i & 31)) & 1u));
        u64* gLM = gbase + (size_t)(i & 1) * NPAR;
        u64* gQH = gLM + NLM; u64* gY = gQH + NQH; u64* gE = gY + NY; u64* gC = gE + NE; u64* gP = gC + NC; u64* gM = gP + NP;
        // every exchange but the (max, index) pairs travels as tagged floats (granule.h): the regions keep their granule-sized
        // slots and use the first half of each
        uint32_t* tLM = reinterpret_cast<uint32_t*>(gLM); uint32_t* tQH = reinterpret_cast<uint32_t*>(gQH);
        uint32_t* tY = reinterpret_cast<uint32_t*>(gY); uint32_t* tE = reinterpret_cast<uint32_t*>(gE);
        uint32_t* tC = reinterpret_cast<uint32_t*>(gC);
        // The 1-bit tag of a tagged float tells a slot's write of this step from the one two steps before ONLY if the slot is
        // written every second step.  p is exchanged at the feedback steps alone (training graph: ~10 % of the steps), so its
        // parity buffer and tag bit follow the COUNT of feedback steps, not the step index: with the step index a first feedback
        // step i with ((i >> 1) & 1) == 1 expected the bit the host's memset left (0), and a poller that came before the
        // publisher took zeros for p -- a wrong draw now and then (found in round 4 by a coin pattern with feedback at steps 2, 3).
        (void)gP;
        uint32_t* tP = reinterpret_cast<uint32_t*>(gbase + (size_t)(nfb & 1) * NPAR + (NLM + NQH + NY + NE + NC));
        const uint32_t tbp = tag_bit(nfb);
        // ---- (1) LM cell of my units: gates = EK[tok] + h_lm_{step-1} . K_h (the matvec ran one step earlier), published into the
        // LM region of `step`'s parity with `step`'s tag bit.  The training graph calls it for step i+1 already DURING step i when
        // the next token is the teacher's (below): the LM recurrence depends on tokens only, not on the attention chain.
        auto lm_cell = [&](int step, float4 s) {
            uint32_t* tL = reinterpret_cast<uint32_t*>(gbase + (size_t)(step & 1) * NPAR);
            const uint32_t tbs = tag_bit(step);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float4 x = *reinterpret_cast<const float4*>(lmsum + ((p * 8 + cu) * R + cr) * 4);
                s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
            }
            const float gi = fast_sigmoid(s.x), gj = fast_tanh(s.y), gf = fast_sigmoid(s.z + 1.0f), go = fast_sigmoid(s.w);
            const float c_old = c_lm;
            c_lm = c_lm * gf + gi * gj;
            const float hl = go * fast_tanh(c_lm);
            if (TRAIN) {
                // DropoutWrapper(output_keep_prob) scales what the decoder sees; the LM's recurrence keeps h itself
                float o = hl;
                if (a.keep < 1.0f) o *= keep_scale(a.seed, (uint32_t)(step * a.B + cb), (uint32_t)(mem * LS + cu), a.keep);
                tagged_publish2(tL + 2 * ((size_t)cr * LMH + mem * LS + cu), tbs, o, hl, fast);
                // bookkeeping stores after the publish (this wave never polls): the record format of csrc/lstm.hip
                // (32-bit element offsets from the uniform base pointers: SGPR base + VGPR offset addressing, no 64-bit
                // running pointers kept in registers across the loop)
                const unsigned ridx = (unsigned)((step * a.B + cb) * LMH + mem * LS + cu);
                a.lm_out[ridx] = __uint_as_float(__float_as_uint(o) & ~1u);          // as every consumer saw it
                a.lm_hprev[ridx] = h_lm_prev;
                float4* rp = reinterpret_cast<float4*>(a.lm_act + ridx * 8u);
                rp[0] = make_float4(gi, gj, gf, go);
                rp[1] = make_float4(c_lm, c_old, 0.f, 0.f);
                h_lm_prev = __uint_as_float(__float_as_uint(hl) & ~1u);
            } else {
                tagged_publish(tL + (size_t)cr * LMH + mem * LS + cu, tbs, hl, fast);
            }
        };
        auto ek_row = [&](int tok) {
            const float* ek = a.ek + (size_t)tok * L4 + mem * LS + cu;
            return make_float4(ek[0], ek[LMH], ek[2 * LMH], ek[3 * LMH]);
        };
        // ---- (2) gather lm_out of `step` into v_dec[:, :LMH] (and, training graph, the plain h into v_lmh)
        auto lm_gather = [&](int step) {
            const uint32_t* tL = reinterpret_cast<const uint32_t*>(gbase + (size_t)(step & 1) * NPAR);
            const uint32_t tbs = tag_bit(step);
            if (TRAIN) {
                // a quad = (dropped, plain) outputs of two adjacent units; NLM / 4 = 512 quads over 384 threads: both of a thread's
                // quads in flight at once (granule.h tagged_poll4_many)
                static_assert(NLM / 4 <= 2 * NPOLL, "two quads per polling thread");
                const int p0 = tid - 64, p1 = p0 + NPOLL;
                const int pp[2] = {p0, p1 < NLM / 4 ? p1 : p0};
                const uint32_t* ptr[2] = {tL + 4 * pp[0], tL + 4 * pp[1]};
                bool need[2] = {rok((2 * pp[0]) / LMH), p1 < NLM / 4 && rok((2 * pp[1]) / LMH)};
                const uint32_t tbb[2] = {tbs, tbs};
                float4 v[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
                tagged_poll4_many<2>(ptr, need, tbb, v, a.err);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j == 1 && p1 >= NLM / 4) break;
                    const int r = (2 * pp[j]) / LMH, k = (2 * pp[j]) % LMH;
                    *reinterpret_cast<float2*>(v_dec + r * KD + k) = make_float2(v[j].x, v[j].z);
                    *reinterpret_cast<float2*>(v_lmh + r * LMH + k) = make_float2(v[j].y, v[j].w);
                }
            } else {
                for (int p = tid - 64; p < NLM / 4; p += NPOLL) {
                    const int idx = 4 * p, r = idx / LMH, k = idx % LMH;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (rok(r)) tagged_poll4(tL + idx, tbs, v, a.err);
                    *reinterpret_cast<float4*>(v_dec + r * KD + k) = v;
                }
            }
        };
        // training graph: where the next token is the teacher's, the LM cell of step i+1 runs during step i (`early`) and its
        // output is gathered together with ctx_i in phase 6, so step i+1 starts directly with the outer cell's matvec
        const bool early = TRAIN && !fbi && i + 1 < a.T;
        float4 ek_next = make_float4(0.f, 0.f, 0.f, 0.f);
        if (early && wave0 && cell && cb_ok) {      // prefetch: teacher token of step i+1 and its EK row (lands during the step)
            int tn = a.tok[(size_t)(i + 1) * a.B + cb];
            tn = min(max(tn, 0), V - 1);
            ek_next = ek_row(tn);
        }
        if (!lm_ready) {
            if (wave0 && cell && cb_ok) lm_cell(i, ek_row(tokr));
            if (poller) lm_gather(i);
            __syncthreads();
        }
        GREEDY_STAMP()
        {
            float acc[R][4];
            f32x2 ap2[R][2];
#pragma unroll
            for (int r = 0; r < R; ++r) ap2[r][0] = ap2[r][1] = f32x2{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x2 xlo[R], xhi[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(v_dec + r * KD + part * 256 + kq * 4 + j * 64);
                    xlo[r] = __builtin_shufflevector(x, x, 0, 1); xhi[r] = __builtin_shufflevector(x, x, 2, 3);
                }
                // eight independent accumulators in rotation
#pragma unroll
                for (int r = 0; r < R; ++r) { pk_fma_alo(ap2[r][0], xlo[r], wdec[j * 4 + 0][0]); pk_fma_alo(ap2[r][1], xlo[r], wdec[j * 4 + 0][1]); }
#pragma unroll
                for (int r = 0; r < R; ++r) { pk_fma_ahi(ap2[r][0], xlo[r], wdec[j * 4 + 1][0]); pk_fma_ahi(ap2[r][1], xlo[r], wdec[j * 4 + 1][1]); }
#pragma unroll
                for (int r = 0; r < R; ++r) { pk_fma_alo(ap2[r][0], xhi[r], wdec[j * 4 + 2][0]); pk_fma_alo(ap2[r][1], xhi[r], wdec[j * 4 + 2][1]); }
#pragma unroll
                for (int r = 0; r < R; ++r) { pk_fma_ahi(ap2[r][0], xhi[r], wdec[j * 4 + 3][0]); pk_fma_ahi(ap2[r][1], xhi[r], wdec[j * 4 + 3][1]); }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc[r][0] = ap2[r][0].x; acc[r][1] = ap2[r][0].y; acc[r][2] = ap2[r][1].x; acc[r][3] = ap2[r][1].y;
                row16_allreduce_sum4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
            }
            if (kq == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    *reinterpret_cast<float4*>(sums + ((part * 8 + u8) * R + r) * 4) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
            }
            if (TRAIN) {
                // LM cell of step i+1, recurrent part: h_lm_i . K_h for my units, from the UNDROPPED h (the inference graph
                // forms it in phase 7, which the training graph runs at feedback steps only).  lmsum was read by this step's
                // phase 1 before the barrier above and is read again only after the barrier below.
                float al[R][4];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(v_lmh + r * LMH + part * 64 + kq * 4);
                    const f32x2 xlo = __builtin_shufflevector(x, x, 0, 1), xhi = __builtin_shufflevector(x, x, 2, 3);
                    f32x2 t0 = f32x2{0.f, 0.f}, t1 = t0;
                    pk_fma_alo(t0, xlo, wlm[0][0]); pk_fma_alo(t1, xlo, wlm[0][1]);
                    pk_fma_ahi(t0, xlo, wlm[1][0]); pk_fma_ahi(t1, xlo, wlm[1][1]);
                    pk_fma_alo(t0, xhi, wlm[2][0]); pk_fma_alo(t1, xhi, wlm[2][1]);
                    pk_fma_ahi(t0, xhi, wlm[3][0]); pk_fma_ahi(t1, xhi, wlm[3][1]);
                    al[r][0] = t0.x; al[r][1] = t0.y; al[r][2] = t1.x; al[r][3] = t1.y;
                    row16_allreduce_sum4(al[r][0], al[r][1], al[r][2], al[r][3]);
                }
                if (kq == 0) {
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        *reinterpret_cast<float4*>(lmsum + ((part * 8 + u8) * R + r) * 4) = make_float4(al[r][0], al[r][1], al[r][2], al[r][3]);
                }
            }
        }
        __syncthreads();
        GREEDY_STAMP()
        if (wave0 && cell) {
            float4 s = make_float4(bp[0], bp[1], bp[2], bp[3]);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float4 x = *reinterpret_cast<const float4*>(sums + ((p * 8 + cu) * R + cr) * 4);
                s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
            }
            const float gi = fast_sigmoid(s.x), gj = fast_tanh(s.y), gf = fast_sigmoid(s.z + 1.0f), go = fast_sigmoid(s.w);
            c_dec = c_dec * gf + gi * gj;
            const float hd = go * fast_tanh(c_dec);
            // q = cell state c (decoder.py:79-80) and h, adjacent granules
            if (cb_ok) tagged_publish2(tQH + 2 * ((size_t)cr * H + mem * HS + cu), tb, c_dec, hd, fast);
            if (TRAIN && cb_ok) {
                const unsigned rowi = (unsigned)(i * a.B + cb);
                const unsigned go_ = rowi * H4 + mem * HS + cu;
                a.dec_gates[go_] = gi; a.dec_gates[go_ + H] = gj; a.dec_gates[go_ + 2 * H] = gf; a.dec_gates[go_ + 3 * H] = go;
                a.dec_c[rowi * H + mem * HS + cu] = c_dec;
                a.dec_h[rowi * H + mem * HS + cu] = hd;
            }
            if (early && cb_ok) lm_cell(i + 1, ek_next);      // lmsum = h_lm_i . K_h was formed before the barrier above
        }
        // ---- (3) gather (q_i, h_i); y slice = q . W_att[:, slice] + b
        if (poller) {
            // a quad = (q, h) of two adjacent units; 512 quads over 384 threads: both of a thread's quads in flight
            static_assert(NQH / 4 <= 2 * NPOLL, "two quads per polling thread");
            const int p0 = tid - 64, p1 = p0 + NPOLL;
            const int pp[2] = {p0, p1 < NQH / 4 ? p1 : p0};
            const uint32_t* ptr[2] = {tQH + 4 * pp[0], tQH + 4 * pp[1]};
            bool need[2] = {rok((2 * pp[0]) / H), p1 < NQH / 4 && rok((2 * pp[1]) / H)};
            const uint32_t tbb[2] = {tb, tb};
            float4 v[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
            tagged_poll4_many<2>(ptr, need, tbb, v, a.err);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j == 1 && p1 >= NQH / 4) break;
                const int r = (2 * pp[j]) / H, k = (2 * pp[j]) % H;
                *reinterpret_cast<float2*>(v_ap + r * KA + k) = make_float2(v[j].x, v[j].z);
                *reinterpret_cast<float2*>(v_dec + r * KD + LMH + k) = make_float2(v[j].y, v[j].w);
            }
        }
        __syncthreads();
        GREEDY_STAMP()
        {
            float acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float4 x = *reinterpret_cast<const float4*>(v_ap + r * KA + ypart * 64 + kq * 4);
                acc[r] = fmaf(x.x, wy[0], fmaf(x.y, wy[1], fmaf(x.z, wy[2], x.w * wy[3])));
            }
            static_assert(R == 4, "row16_allreduce_sum4");
            row16_allreduce_sum4(acc[0], acc[1], acc[2], acc[3]);
            if (kq == 0 && yact) {
#pragma unroll
                for (int r = 0; r < R; ++r) ysum[(ypart * AS + ycol) * R + r] = acc[r];
            }
        }
        __syncthreads();
        GREEDY_STAMP()
        if (wave0 && tid < R * AS) {
            const int r = tid >> 2, col = tid & 3;
            const float yv = ybias + (ysum[(0 * AS + col) * R + r] + ysum[(1 * AS + col) * R + r]) +
                             (ysum[(2 * AS + col) * R + r] + ysum[(3 * AS + col) * R + r]);
            if (rok(r)) {
                tagged_publish(tY + (size_t)r * A + mem * AS + col, tb, yv, fast);
                if (TRAIN) a.y[(unsigned)((i * a.B + r0 + r) * A + mem * AS + col)] = yv;
            }
        }
        // ---- (4) gather y, scores on my position slice
        if (poller) {
            for (int p = tid - 64; p < NY / 4; p += NPOLL) {
                const int idx = 4 * p, r = idx / A;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rok(r)) tagged_poll4(tY + idx, tb, v, a.err);
                *reinterpret_cast<float4*>(yl + idx) = v;
            }
        }
        __syncthreads();
        GREEDY_STAMP()
        if constexpr (HFL) {
            const int tl = row % MAXTS, r = row / MAXTS;     // DPP row -> (utterance, position); lane kq -> A/16 columns
            float sc = 0.f;
            if (tl < TS) {
                const float* hrow = hfl + (r * MAXTS + tl) * A;
                const float* yrow = yl + r * A;
#pragma unroll
                for (int c = 0; c < AL / 4; ++c) {
                    const int a0 = c * 64 + kq * 4;
                    const float4 h4 = *reinterpret_cast<const float4*>(hrow + a0);
                    const float4 y4 = *reinterpret_cast<const float4*>(yrow + a0);
                    const float4 v4 = *reinterpret_cast<const float4*>(vl + a0);
                    sc = fmaf(v4.x, fast_tanh(h4.x + y4.x), sc); sc = fmaf(v4.y, fast_tanh(h4.y + y4.y), sc);
                    sc = fmaf(v4.z, fast_tanh(h4.z + y4.z), sc); sc = fmaf(v4.w, fast_tanh(h4.w + y4.w), sc);
                }
            }
            sc = row16_allreduce_sum(sc);
            if (kq == 0) eout[row] = sc;
        } else {
            // two passes of 8 positions per utterance (DPP row -> (utterance, position % 8)); the hf rows are in registers (hreg)
            static_assert(MAXTS == 8 || AL / 4 == 2, "two 64-column chunks per row");
            const int r = row / 8;
            const float* yrow = yl + r * A;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int tl = (row % 8) + 8 * ps;
                float sc = 0.f;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int a0 = c * 64 + kq * 4;
                    const float4 y4 = *reinterpret_cast<const float4*>(yrow + a0);
                    const float4 v4 = *reinterpret_cast<const float4*>(vl + a0);
                    const float4 x4 = hreg[ps][c];
                    sc = fmaf(v4.x, fast_tanh(x4.x + y4.x), sc); sc = fmaf(v4.y, fast_tanh(x4.y + y4.y), sc);
                    sc = fmaf(v4.z, fast_tanh(x4.z + y4.z), sc); sc = fmaf(v4.w, fast_tanh(x4.w + y4.w), sc);
                }
                sc = row16_allreduce_sum(sc);
                if (kq == 0) eout[r * MAXTS + tl] = (tl < TS && tau0 + tl < Te) ? sc : 0.f;
            }
        }
        __syncthreads();
        GREEDY_STAMP()
        if (wave0 && tid < R * MAXTS) {
            const int tl = tid % MAXTS, r = tid / MAXTS;      // all MAXTS slots (quads are polled whole; slots >= TS are masked later)
            if (rok(r)) tagged_publish(tE + (size_t)r * G * MAXTS + mem * MAXTS + tl, tb, eout[tid], fast);
        }
        // ---- (5) gather all scores, softmax over tau < len (replicated), context slice
        if (poller) {
            const int nq = (TS + 3) / 4;                        // quads per (utterance, source workgroup): 1 or 2 (MAXTS = 8), up to 4 (16)
            const int lq = nq > 2 ? 2 : nq - 1;                 // log2 of the quads polled (3 -> 4: every slot is published) ...
            const int mq = (1 << lq) - 1;                       // ... so the index arithmetic is shifts, not four runtime divisions
            static_assert(G == 32, "r = p >> (5 + lq)");
            for (int p = tid - 64; p < ((R * G) << lq); p += NPOLL) {
                const int r = p >> (5 + lq), rem = p & ((G << lq) - 1), m = rem >> lq, q = rem & mq;
                const int off = r * G * MAXTS + m * MAXTS + 4 * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rok(r)) tagged_poll4(tE + off, tb, v, a.err);
                *reinterpret_cast<float4*>(el + off) = v;
            }
        }
        __syncthreads();
        GREEDY_STAMP()
        if (wave < R) {      // wave r: softmax of utterance r; slot (m, tl) <-> tau = m*TS + tl
            const int r = wave, L = blen[r];
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
            for (int j = 0; j < G * MAXTS / 64; ++j) er[lane + 64 * j] = ev[j] * inv;
        }
        __syncthreads();
        GREEDY_STAMP()
        {
            // alpha is exactly zero past the length and in unused slots (tl >= TS), so every thread runs the same loops
            // (positions past Te are clamped onto the last row: alpha = 0 there)
            const int dd = tid & 15, r = (tid >> 4) & 3, tp = tid >> 6;
            float cs = 0.f, cs2 = 0.f;
#pragma unroll
            for (int mm = 0; mm < G / 8; ++mm) {
                const int m = tp + 8 * mm;
                const float* ap = el + r * G * MAXTS + m * MAXTS;
                const float* xr = encl + (size_t)r * Te * DS + dd;
                // all MAXTS slots, loads first (slots >= TS hold alpha = 0; their position is clamped onto a valid row): the serial
                // "load, load, fma" chain over TS positions was 11 % of the step; the unrolled form spilled until round 3 freed registers
#pragma unroll
                for (int h8 = 0; h8 < MAXTS / 8; ++h8) {
                    const float4 a0 = *reinterpret_cast<const float4*>(ap + 8 * h8), a1 = *reinterpret_cast<const float4*>(ap + 8 * h8 + 4);
                    float xv[8];
#pragma unroll
                    for (int tl = 0; tl < 8; ++tl) xv[tl] = xr[min(m * TS + min(8 * h8 + tl, TS - 1), Te - 1) * DS];
                    cs = fmaf(a0.x, xv[0], cs); cs2 = fmaf(a0.y, xv[1], cs2); cs = fmaf(a0.z, xv[2], cs); cs2 = fmaf(a0.w, xv[3], cs2);
                    cs = fmaf(a1.x, xv[4], cs); cs2 = fmaf(a1.y, xv[5], cs2); cs = fmaf(a1.z, xv[6], cs); cs2 = fmaf(a1.w, xv[7], cs2);
                }
            }
            cpart[(tp * R + r) * DS + dd] = cs + cs2;
        }
        __syncthreads();
        GREEDY_STAMP()
        if (wave0) {
            const int r = lane >> 4, dd = lane & 15;
            float cs = 0.f;
#pragma unroll
            for (int tp = 0; tp < 8; ++tp) cs += cpart[(tp * R + r) * DS + dd];
            if (rok(r)) {
                tagged_publish(tC + (size_t)r * D + mem * DS + dd, tb, cs, fast);
                if (TRAIN) a.ctx[(unsigned)((i * a.B + r0 + r) * D + mem * DS + dd)] = cs;
            }
            if (TRAIN && lane < R * MAXTS) {      // alpha of this step -> global: every workgroup stores its own position slice
                const int ra = lane / MAXTS, tl = lane % MAXTS, tau = tau0 + tl;
                if (tl < TS && tau < Te && rok(ra))
                    a.alpha[(unsigned)((i * a.B + r0 + ra) * Te + tau)] = el[ra * G * MAXTS + mem * MAXTS + tl];
            }
        }
        // ---- (6) gather ctx_i; AttnProjection slice
        if (poller) {
            // 512 context quads over 384 threads, and in the training graph the 512 quads of the early LM output of step i+1
            // (published right after this step's outer cell): all of a thread's up to four quads in flight at once
            static_assert(NC / 4 <= 2 * NPOLL, "two context quads per polling thread");
            const int p0 = tid - 64, p1 = p0 + NPOLL;
            const int pc[2] = {p0, p1 < NC / 4 ? p1 : p0};
            if (TRAIN && early) {
                const uint32_t* tL = reinterpret_cast<const uint32_t*>(gbase + (size_t)((i + 1) & 1) * NPAR);
                const uint32_t tbs = tag_bit(i + 1);
                const int pl[2] = {p0, p1 < NLM / 4 ? p1 : p0};
                const uint32_t* ptr[4] = {tC + 4 * pc[0], tC + 4 * pc[1], tL + 4 * pl[0], tL + 4 * pl[1]};
                bool need[4] = {rok((4 * pc[0]) / D), p1 < NC / 4 && rok((4 * pc[1]) / D),
                                rok((2 * pl[0]) / LMH), p1 < NLM / 4 && rok((2 * pl[1]) / LMH)};
                const uint32_t tbb[4] = {tb, tb, tbs, tbs};
                float4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                tagged_poll4_many<4>(ptr, need, tbb, v, a.err);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (!(j == 1 && p1 >= NC / 4)) {
                        const int idx = 4 * pc[j], r = idx / D, k = idx % D;
                        *reinterpret_cast<float4*>(v_ap + r * KA + H + k) = v[j];
                        *reinterpret_cast<float4*>(v_dec + r * KD + LMH + H + k) = v[j];
                    }
                    if (!(j == 1 && p1 >= NLM / 4)) {
                        const int r = (2 * pl[j]) / LMH, k = (2 * pl[j]) % LMH;
                        *reinterpret_cast<float2*>(v_dec + r * KD + k) = make_float2(v[2 + j].x, v[2 + j].z);
                        *reinterpret_cast<float2*>(v_lmh + r * LMH + k) = make_float2(v[2 + j].y, v[2 + j].w);
                    }
                }
            } else {
                const uint32_t* ptr[2] = {tC + 4 * pc[0], tC + 4 * pc[1]};
                bool need[2] = {rok((4 * pc[0]) / D), p1 < NC / 4 && rok((4 * pc[1]) / D)};
                const uint32_t tbb[2] = {tb, tb};
                float4 v[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
                tagged_poll4_many<2>(ptr, need, tbb, v, a.err);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j == 1 && p1 >= NC / 4) break;
                    const int idx = 4 * pc[j], r = idx / D, k = idx % D;
                    *reinterpret_cast<float4*>(v_ap + r * KA + H + k) = v[j];
                    *reinterpret_cast<float4*>(v_dec + r * KD + LMH + H + k) = v[j];
                }
            }
        }
        __syncthreads();
        GREEDY_STAMP()
        lm_ready = early;
        if (!fbi) continue;  // training graph, teacher-forced step: no projection in the loop (uniform: fbi is the same for every
                             // thread of the grid); the next token's LM cell has run already, or this was the last step
        {
            float wap_l[12];
            if (TRAIN) { int z = 0; asm volatile("" : "+s"(z)); load_wap(wap_l, z); }
            const float* wap = TRAIN ? wap_l : wap_r;
            float acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc[r] = 0.f;
                const float* sv = v_ap + r * KA + part * 192 + kq * 4;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float4 x = *reinterpret_cast<const float4*>(sv + j * 64);
                    acc[r] = fmaf(x.x, wap[j * 4 + 0], acc[r]); acc[r] = fmaf(x.y, wap[j * 4 + 1], acc[r]);
                    acc[r] = fmaf(x.z, wap[j * 4 + 2], acc[r]); acc[r] = fmaf(x.w, wap[j * 4 + 3], acc[r]);
                }
            }
            row16_allreduce_sum4(acc[0], acc[1], acc[2], acc[3]);
            if (kq == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) psum[(part * PS + u8) * R + r] = acc[r];
            }
        }
        __syncthreads();
        GREEDY_STAMP()
        if (wave0 && cell) {
            const float pv = pbias + (psum[(0 * PS + cu) * R + cr] + psum[(1 * PS + cu) * R + cr]) +
                             (psum[(2 * PS + cu) * R + cr] + psum[(3 * PS + cu) * R + cr]);
            if (cb_ok) tagged_publish(tP + (size_t)cr * H + mem * PS + cu, tbp, pv, fast);
        }
        // ---- (7) gather p; logits of my vocabulary slice (+ the h-part of the NEXT step's LM gates)
        if (poller) {
            for (int p = tid - 64; p < NP / 4; p += NPOLL) {
                const int idx = 4 * p, r = idx / H;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rok(r)) tagged_poll4(tP + idx, tbp, v, a.err);
                *reinterpret_cast<float4*>(v_p + idx) = v;
            }
        }
        __syncthreads();
        GREEDY_STAMP()
        {
            float wout_l[16];
            if (TRAIN) { int z = 0; asm volatile("" : "+s"(z)); load_wout(wout_l, z); }
            const float* wout = TRAIN ? wout_l : wout_r;
            float acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc[r] = 0.f;
                const float* sv = v_p + r * H + kq * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 x = *reinterpret_cast<const float4*>(sv + j * 64);
                    acc[r] = fmaf(x.x, wout[j * 4 + 0], acc[r]); acc[r] = fmaf(x.y, wout[j * 4 + 1], acc[r]);
                    acc[r] = fmaf(x.z, wout[j * 4 + 2], acc[r]); acc[r] = fmaf(x.w, wout[j * 4 + 3], acc[r]);
                }
            }
            row16_allreduce_sum4(acc[0], acc[1], acc[2], acc[3]);
            if (kq == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) lg[r * VS + row] = acc[r] + outb;
            }
            if (!TRAIN) {
                // LM cell of step i+1, recurrent part: h_lm_i . K_h for my units (v_dec[:, :LMH] holds lm_out_i)
                float al[R][4];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(v_dec + r * KD + part * 64 + kq * 4);
                    const f32x2 xlo = __builtin_shufflevector(x, x, 0, 1), xhi = __builtin_shufflevector(x, x, 2, 3);
                    f32x2 t0 = f32x2{0.f, 0.f}, t1 = t0;
                    pk_fma_alo(t0, xlo, wlm[0][0]); pk_fma_alo(t1, xlo, wlm[0][1]);
                    pk_fma_ahi(t0, xlo, wlm[1][0]); pk_fma_ahi(t1, xlo, wlm[1][1]);
                    pk_fma_alo(t0, xhi, wlm[2][0]); pk_fma_alo(t1, xhi, wlm[2][1]);
                    pk_fma_ahi(t0, xhi, wlm[3][0]); pk_fma_ahi(t1, xhi, wlm[3][1]);
                    al[r][0] = t0.x; al[r][1] = t0.y; al[r][2] = t1.x; al[r][3] = t1.y;
                    row16_allreduce_sum4(al[r][0], al[r][1], al[r][2], al[r][3]);
                }
                if (kq == 0) {
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        *reinterpret_cast<float4*>(lmsum + ((part * 8 + u8) * R + r) * 4) = make_float4(al[r][0], al[r][1], al[r][2], al[r][3]);
                }
            }
        }
        __syncthreads();
        GREEDY_STAMP()
        if (!TRAIN && wave7) {   // logits -> global (raw_rnn emits zeros for finished rows, attn_decoder.py:170); the training
                                 // graph's logits come from the hoisted GEMMs over all steps
            for (int idx = lane; idx < R * VS; idx += 64) {
                const int r = idx / VS, c = idx % VS, vc = mem * VS + c;
                if (rok(r) && vc < V)
                    a.logits[((size_t)i * a.B + r0 + r) * V + vc] = (i < (idx < 64 ? slen7a : slen7b)) ? lg[idx] : 0.f;
            }
        }
        int tok_next = 0;
        if (wave0) {         // argmax over my slice: lane -> (utterance r, columns c and c + 16), then 16-lane butterflies
            const int r = lane >> 4, c = lane & 15;
            const bool live = rok(r) && i < slen0;
            float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int cc = c + 16 * h2, vc = mem * VS + cc;
                if (vc < V) {
                    float val = live ? lg[r * VS + cc] : 0.f;
                    if (TRAIN) {   // tf.multinomial as Gumbel-max, the generator of next_token_kernel (csrc/loss.hip): same draw
                        const float u = fmaxf(uniform01(a.seed, (uint32_t)i * 65537u + (uint32_t)(r0 + r), (uint32_t)vc), 1e-12f);
                        val -= logf(-logf(u));
                    }
                    amax_take(bv, bi, val, vc);
                }
            }
            row16_argmax(bv, bi);
            if (c == 0 && rok(r)) chain_publish2(gM + 2 * ((size_t)r * G + mem), ep, bv, __int_as_float(bi), fast);
        }
        // ---- (8) gather the 32 partial maxima of every utterance -> tok_{i+1} (every workgroup, redundantly)
        if (poller) {
            for (int p = tid - 64; p < R * G; p += NPOLL) {
                const int r = p / G;
                float v0 = -INFINITY, v1 = __int_as_float(0x7fffffff);
                if (rok(r)) chain_poll2(gM + 2 * p, ep, v0, v1, a.err);
                mv[p] = v0; mi[p] = __float_as_int(v1);
            }
        }
        __syncthreads();
        GREEDY_STAMP()
        if (wave0) {
            const int r = lane >> 4, m2 = lane & 15;
            float bv = mv[r * G + m2]; int bi = mi[r * G + m2];
            amax_take(bv, bi, mv[r * G + m2 + 16], mi[r * G + m2 + 16]);
            row16_argmax(bv, bi);
            tok_next = bi == 0x7fffffff ? 0 : bi;
            if (mem == 0 && m2 == 0 && rok(r) && i + 1 < a.T) a.tok[(size_t)(i + 1) * a.B + r0 + r] = tok_next;
            tokr = __shfl(tok_next, cr * 16);
            tokr = min(max(tokr, 0), V - 1);
        }
        GREEDY_STAMP()
        ++nfb;
        // (LDS written by this step's last phases is rewritten only after later barriers of the next step)
    }
    if (STAMP && a.dbg && blockIdx.x == 0 && threadIdx.x == 0) { for (int j = 0; j < 24; ++j) a.dbg[48 + j] = stamp[j]; }
#undef GREEDY_STAMP
}

}  // namespace asr

// positions per workgroup of the instantiation that takes Te (see the kernel), and the dynamic LDS it asks for
static int greedy_maxts(int Te) { return Te <= 256 ? 8 : 16; }
static size_t greedy_lds_bytes(int Te, bool train) {
    constexpr int R = 4, KD = 1024, KA = 768, Hc = 256, Ac = 128, G = 32, lmH = 256;
    const int mts = greedy_maxts(Te);
    return sizeof(float) * (4 + (size_t)R * KD + R * KA + R * Hc + R * Ac + R * G * mts + 2 * (4 * 8 * R * 4) +
                            4 * 4 * R + 4 * 8 * R + 8 * R * 16 + R * mts + R * 32 + 2 * R * G + Ac + (mts == 8 ? R * mts * Ac : 0) +
                            (train ? (size_t)R * lmH : 0) + (size_t)R * Te * 16);
}
extern "C" int asr_decoder_greedy_supported(int B, int Te, int D, int A, int H, int lmH, int E, int V) {
    if (getenv("ASR_DEC_GREEDY") && atoi(getenv("ASR_DEC_GREEDY")) == 0) return 0;
    (void)E;
    if (asr::resident_wg_budget() < 256) return 0;         // 8 one-XCD groups of 32 workgroups per launch must be co-resident
    const char* te_env = getenv("ASR_DEC_GREEDY_TEMAX");      // (256: round 3's limit -- longer memories on the segment chains; read per call: tests switch it)
    const int te_max = te_env ? atoi(te_env) : 512;
    if (!(B > 0 && Te > 0 && Te <= 512 && Te <= te_max && V > 0 && V <= 1024 && H == 256 && D == 512 && A == 128 && lmH == 256)) return 0;
    return greedy_lds_bytes(Te, true) <= 160 * 1024 - 64;  // (Te <= 256 always fits; 16 positions per workgroup: up to Te = 419)
}

static size_t greedy_npar(int D, int A, int H, int lmH, bool train, int maxts = 16) {      // = NPAR of the kernel instantiation
    return 4 * ((train ? 2 : 1) * (size_t)lmH + 2 * (size_t)H + A + 32 * (size_t)maxts + D + H + 2 * 32);
}
static size_t greedy_gran_bytes(int B, int D, int A, int H, int lmH) {     // sized for the largest layout (training, 16 positions)
    const size_t groups = ((size_t)B + 3) / 4;
    return (groups * 2 * greedy_npar(D, A, H, lmH, true, 16) * sizeof(u64) + groups * 32 * sizeof(u64) + 255) / 256 * 256;
}
// granules + XCC slots | EK [V][4 lmH]
extern "C" size_t asr_decoder_greedy_ws_bytes(int B, int D, int A, int H, int lmH, int V) {
    return greedy_gran_bytes(B, D, A, H, lmH) + (size_t)V * 4 * lmH * sizeof(float);
}

extern "C" int asr_gemm_f32(void* stream, int transA, int transB, int M, int N, int K, const float* A, int lda,
                            const float* B, int ldb, float* C, int ldc, const float* bias, int accumulate);

// Shared launcher of the inference-graph and training-graph instantiations; `a` arrives with the mode-specific members
// set.  ws: asr_decoder_greedy_ws_bytes().
static int greedy_launch(void* stream, asr::GreedyArgs a, bool train, const float* embedding, const float* lm_kernel,
                         const float* lm_bias, void* ws, int B, int Te, int D, int A, int H, int lmH, int E, int V) {
    using namespace asr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t gbytes = greedy_gran_bytes(B, D, A, H, lmH);
    float* ek = reinterpret_cast<float*>(static_cast<char*>(ws) + gbytes);
    if (hipMemsetAsync(ws, 0, gbytes, s) != hipSuccess) return ASR_ELAUNCH;
    int rc;
    if ((rc = asr_gemm_f32(stream, 0, 0, V, 4 * lmH, E, embedding, E, lm_kernel, 4 * lmH, ek, 4 * lmH, lm_bias, 0))) return rc;
    a.ek = ek; a.lm_kh = lm_kernel + (size_t)E * 4 * lmH;
    a.gx = static_cast<u64*>(ws);
    const int groups = (B + 3) / 4;
    const int mts = greedy_maxts(Te);
    a.xcc_slots = a.gx + (size_t)groups * 2 * greedy_npar(D, A, H, lmH, train, mts);
    a.B = B; a.Te = Te; a.V = V;
    constexpr int G = 32;
    const size_t lds = greedy_lds_bytes(Te, train);
    if (lds > 160 * 1024 - 64) return ASR_EUNSUPPORTED;
    a.dbg = getenv("ASR_CHAIN_STAMP") ? asr::g_lstm_dbg : nullptr;
    const void* kfn;
    if (mts == 16) kfn = train ? reinterpret_cast<const void*>(&asr::decoder_greedy_kernel<256, 512, 128, 256, false, true, 16>)
                               : reinterpret_cast<const void*>(&asr::decoder_greedy_kernel<256, 512, 128, 256, false, false, 16>);
    else if (a.dbg) kfn = train ? reinterpret_cast<const void*>(&asr::decoder_greedy_kernel<256, 512, 128, 256, true, true>)
                                : reinterpret_cast<const void*>(&asr::decoder_greedy_kernel<256, 512, 128, 256, true>);
    else kfn = train ? reinterpret_cast<const void*>(&asr::decoder_greedy_kernel<256, 512, 128, 256, false, true>)
                     : reinterpret_cast<const void*>(&asr::decoder_greedy_kernel<256, 512, 128, 256>);
    (void)hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int g0 = 0; g0 < groups; g0 += 8) {            // 8 groups (one per XCD) = 256 workgroups per launch
        a.g0 = g0; a.ng = std::min(8, groups - g0);
        if (mts == 16 && train) hipLaunchKernelGGL((asr::decoder_greedy_kernel<256, 512, 128, 256, false, true, 16>), dim3(8 * G), dim3(512), lds, s, a);
        else if (mts == 16) hipLaunchKernelGGL((asr::decoder_greedy_kernel<256, 512, 128, 256, false, false, 16>), dim3(8 * G), dim3(512), lds, s, a);
        else if (train && a.dbg) hipLaunchKernelGGL((asr::decoder_greedy_kernel<256, 512, 128, 256, true, true>), dim3(8 * G), dim3(512), lds, s, a);
        else if (train) hipLaunchKernelGGL((asr::decoder_greedy_kernel<256, 512, 128, 256, false, true>), dim3(8 * G), dim3(512), lds, s, a);
        else if (a.dbg) hipLaunchKernelGGL((asr::decoder_greedy_kernel<256, 512, 128, 256, true>), dim3(8 * G), dim3(512), lds, s, a);
        else hipLaunchKernelGGL((asr::decoder_greedy_kernel<256, 512, 128, 256>), dim3(8 * G), dim3(512), lds, s, a);
        if (hipGetLastError() != hipSuccess) return ASR_ELAUNCH;
    }
    return ASR_OK;
}

// All T steps of the greedy decode.  wk / bprime: the folded InputProjection (decoder.hip); tok row 0 holds the first
// input token of every utterance, rows 1.. are written.  ws: asr_decoder_greedy_ws_bytes().
extern "C" int asr_decoder_greedy_fwd(void* stream, const float* embedding, const float* lm_kernel, const float* lm_bias,
                                      const float* wk, const float* bprime, const float* dec_kh, const float* w_att,
                                      const float* b_att, const float* v, const float* ap_w, const float* ap_b,
                                      const float* out_w, const float* out_b, const float* hf, const float* enc,
                                      const int* enc_len, const int* seq_len, int* tok, float* logits, void* ws, int* err,
                                      int B, int Te, int D, int A, int H, int lmH, int E, int V, int T) {
    using namespace asr;
    if (!asr_decoder_greedy_supported(B, Te, D, A, H, lmH, E, V) || T <= 0) return ASR_EUNSUPPORTED;
    if (!embedding || !lm_kernel || !lm_bias || !wk || !bprime || !dec_kh || !w_att || !b_att || !v || !ap_w || !ap_b ||
        !out_w || !out_b || !hf || !enc || !enc_len || !seq_len || !tok || !logits || !ws || !err) return ASR_EINVAL;
    GreedyArgs a = {};
    a.wk = wk; a.bprime = bprime; a.dec_kh = dec_kh;
    a.w_att = w_att; a.b_att = b_att; a.v = v; a.ap_w = ap_w; a.ap_b = ap_b; a.out_w = out_w; a.out_b = out_b;
    a.hf = hf; a.enc = enc; a.enc_len = enc_len; a.seq_len = seq_len; a.tok = tok; a.logits = logits;
    a.err = err; a.T = T;
    return greedy_launch(stream, a, false, embedding, lm_kernel, lm_bias, ws, B, Te, D, A, H, lmH, E, V);
}

// Training graph (teacher forcing with scheduled-sampling feedback at the steps flagged in fbmask8, LM dropout, saved
// activations): the same one-XCD-group kernel, TRAIN instantiation.  tok holds the teacher tokens of all T steps (rows after
// a feedback step are overwritten with the drawn token); the logits are NOT produced here (hoisted GEMMs, decoder.hip).
int asr_decoder_train_fwd(void* stream, const float* embedding, const float* lm_kernel, const float* lm_bias,
                          const float* wk, const float* bprime, const float* dec_kh, const float* w_att,
                          const float* b_att, const float* v, const float* ap_w, const float* ap_b,
                          const float* out_w, const float* out_b, const float* hf, const float* enc,
                          const int* enc_len, const int* seq_len, int* tok, const unsigned* fbmask8, float keep, unsigned seed,
                          float* lm_out, float* lm_hprev, float* lm_act, float* dec_gates, float* dec_c, float* dec_h,
                          float* y, float* alpha, float* ctx, void* ws, int* err,
                          int B, int Te, int D, int A, int H, int lmH, int E, int V, int T) {
    using namespace asr;
    if (!asr_decoder_greedy_supported(B, Te, D, A, H, lmH, E, V) || T <= 0 || T > 256) return ASR_EUNSUPPORTED;
    if (!embedding || !lm_kernel || !lm_bias || !wk || !bprime || !dec_kh || !w_att || !b_att || !v || !ap_w || !ap_b ||
        !out_w || !out_b || !hf || !enc || !enc_len || !seq_len || !tok || !fbmask8 || !lm_out || !lm_hprev || !lm_act ||
        !dec_gates || !dec_c || !dec_h || !y || !alpha || !ctx || !ws || !err) return ASR_EINVAL;
    GreedyArgs a = {};
    a.wk = wk; a.bprime = bprime; a.dec_kh = dec_kh;
    a.w_att = w_att; a.b_att = b_att; a.v = v; a.ap_w = ap_w; a.ap_b = ap_b; a.out_w = out_w; a.out_b = out_b;
    a.hf = hf; a.enc = enc; a.enc_len = enc_len; a.seq_len = seq_len; a.tok = tok; a.logits = nullptr;
    a.err = err; a.T = T;
    for (int j = 0; j < 8; ++j) a.fbmask[j] = fbmask8[j];
    a.keep = keep; a.seed = seed;
    a.lm_out = lm_out; a.lm_hprev = lm_hprev; a.lm_act = lm_act; a.dec_gates = dec_gates; a.dec_c = dec_c; a.dec_h = dec_h;
    a.y = y; a.alpha = alpha; a.ctx = ctx;
    return greedy_launch(stream, a, true, embedding, lm_kernel, lm_bias, ws, B, Te, D, A, H, lmH, E, V);
}
