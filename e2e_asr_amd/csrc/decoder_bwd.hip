// Attention decoder, backward (tf.gradients through attn_decoder.py:37-172 / raw_rnn).
//
// Everything that does not sit on the reverse-time dependency chain is hoisted out of the
// loop into MFMA GEMMs over all T_out*B rows at once:
//   before the loop:  dP   = dLogits . W_out^T            dQC = dP . W_ap^T  (= [dq | dctx])
//   after the loop:   every weight gradient (X^T . dY), bias gradients (column sums), the
//                     embedding scatter-add, dAttnW = enc^T . dhf and denc += dhf . AttnW^T.
// The loop itself (i = T_out-1 .. 0) carries d(dec c,h), d(lm c,h), dctx and runs per step:
//   K1  per-utterance fused kernel: attention backward (dalpha, softmax bwd, tanh bwd,
//       dhf/denc accumulation, dy) + dq = dy.W_att^T + outer-cell pointwise backward -> dG_dec
//   S1  [dx | dh_prev]      = dG_dec . K_dec^T      (skinny MFMA, transposed weight)
//   S2  [dlm_out | dctx_prev] = dx . W_inp^T
//   (S2b dlm = dlm_out . W_simple^T when SimpleProjection exists)
//   K2  lm-cell pointwise backward (dropout mask re-derived from the counter hash) -> dG_lm
//   S3  [demb | dlm_h_prev] = dG_lm . K_lm^T
// A finished row (t >= seq_len[b]) has dLogits = 0 and zero loss weight, so its state carries
// no gradient; raw_rnn's copy-through therefore needs no special case (see decoder.hip).
#include "common.h"
#include <cstdlib>
#include <algorithm>
#include <type_traits>
#include "../../include/e2e_asr_hip.h"

namespace asr {

struct DecBwdStepArgs {
    // attention operands
    const float* q; const float* w_att; const float* b_att; const float* v;
    const float* hf; const float* enc; const int* enc_len; const float* alpha;
    const float* y_saved;      // [B][A] from the forward, or nullptr (then recomputed)
    const float* dqc;          // [B][H+D]: dq_ap | dctx_ap for this step
    const float* dctx_carry; int ld_carry;   // dLC[i+1][:, P:] or nullptr
    float* dhf; float* dctx_out;   // accumulator [B][Te][A]; this step's total dctx [B][D]
    float* dy;                 // [B][A] this step
    float* dv_part;            // [B][A] accumulated over steps
    // outer cell operands
    float* gates;              // [B][4H] in: activated gates, out: dG
    const float* c_prev;       // [B][H] or nullptr
    const float* dh_carry; int ld_dh;        // dXH[i+1][:, E:] or nullptr
    float* dc_carry;           // [B][H] in/out
    float* dq_out;             // attention-only form (gates == nullptr): dq_ap + dq_att [B][H] out (asr_attn_bwd)
    int B, Te, H, A, D;
};

// dynamic LDS: dctx[D] | al[Te] (alpha, then de) | ys[A] | qs[H] | dys[A] | part[...]
// 512 threads; each phase issues all of its global loads before consuming any.
constexpr int DBW_NT = 512;
__global__ __launch_bounds__(DBW_NT) void dec_attn_cell_bwd_kernel(DecBwdStepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = DBW_NT, RW = NT / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int A = a.A, H = a.H, D = a.D, Te = a.Te;
    float* dctx = smem;
    float* al = dctx + ((D + 3) & ~3);
    float* ys = al + ((Te + 3) & ~3);
    float* qs = ys + ((A + 3) & ~3);
    float* dys = qs + ((H + 3) & ~3);
    float* part = dys + ((A + 3) & ~3);
    __shared__ float wred[8];
    const int L = min(max(a.enc_len[b], 0), Te);

    for (int d = tid; d < D; d += NT) {
        float x = a.dqc[(size_t)b * (H + D) + H + d];
        if (a.dctx_carry) x += a.dctx_carry[(size_t)b * a.ld_carry + d];
        dctx[d] = x;
        a.dctx_out[(size_t)b * D + d] = x;        // denc = sum_i alpha_i^T . dctx_i is a post-loop batched GEMM
    }
    for (int t = tid; t < Te; t += NT) al[t] = a.alpha[(size_t)b * Te + t];
    for (int k = tid; k < H; k += NT) qs[k] = a.q[(size_t)b * H + k];
    __syncthreads();
    const int kq = lane & 15, rr = tid >> 4;
    // ---- dalpha[tau] = dctx . enc[tau]  (one DPP row per tau; PP passes x CD chunks in flight)
    {
        constexpr int PP = 4, CD = 8;                 // D <= 512 fast path (CD*16 float4 per row)
        const int nch = ((D >> 2) + 15) / 16;
        float sdot = 0.f;
        for (int t0 = 0; t0 < L; t0 += RW * PP) {
            if (nch <= CD) {
                float4 ev[PP][CD];
#pragma unroll
                for (int p = 0; p < PP; ++p) {
                    const int tau = t0 + p * RW + rr;
#pragma unroll
                    for (int c = 0; c < CD; ++c) {
                        const int d4 = kq + 16 * c;
                        ev[p][c] = (tau < L && d4 < (D >> 2))
                            ? *reinterpret_cast<const float4*>(a.enc + ((size_t)b * Te + tau) * D + 4 * d4)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int p = 0; p < PP; ++p) {
                    const int tau = t0 + p * RW + rr;
                    float sc = 0.f;
#pragma unroll
                    for (int c = 0; c < CD; ++c) {
                        const int d4 = kq + 16 * c;
                        if (d4 < (D >> 2)) {
                            const float4 dv = *reinterpret_cast<const float4*>(dctx + 4 * d4);
                            sc = fmaf(ev[p][c].x, dv.x, sc); sc = fmaf(ev[p][c].y, dv.y, sc);
                            sc = fmaf(ev[p][c].z, dv.z, sc); sc = fmaf(ev[p][c].w, dv.w, sc);
                        }
                    }
                    sc = row16_allreduce_sum(sc);
                    if (kq == 0 && tau < L) { part[tau] = sc; sdot += al[tau] * sc; }
                }
            } else {
                for (int p = 0; p < PP; ++p) {
                    const int tau = t0 + p * RW + rr;
                    float sc = 0.f;
                    if (tau < L)
                        for (int d4 = kq; d4 < (D >> 2); d4 += 16) {
                            const float4 e4 = *reinterpret_cast<const float4*>(a.enc + ((size_t)b * Te + tau) * D + 4 * d4);
                            const float4 dv = *reinterpret_cast<const float4*>(dctx + 4 * d4);
                            sc = fmaf(e4.x, dv.x, sc); sc = fmaf(e4.y, dv.y, sc); sc = fmaf(e4.z, dv.z, sc); sc = fmaf(e4.w, dv.w, sc);
                        }
                    sc = row16_allreduce_sum(sc);
                    if (kq == 0 && tau < L) { part[tau] = sc; sdot += al[tau] * sc; }
                }
            }
        }
        sdot = wave_allreduce_sum(kq == 0 ? sdot : 0.f);
        if (lane == 0) wred[wave] = sdot;
        __syncthreads();
        float S = 0.f;
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) S += wred[i];
        for (int t = tid; t < L; t += NT) al[t] = al[t] * (part[t] - S);     // de[tau]
        __syncthreads();
    }
    // ---- y = q.W_att + b_att: taken from the forward when saved, else recomputed
    if (a.y_saved) {
        for (int aa = tid; aa < A; aa += NT) ys[aa] = a.y_saved[(size_t)b * A + aa];
        __syncthreads();
    } else {
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
                    s.x = fmaf(qk, wv[i].x, s.x); s.y = fmaf(qk, wv[i].y, s.y); s.z = fmaf(qk, wv[i].z, s.z); s.w = fmaf(qk, wv[i].w, s.w);
                }
            }
        }
        *reinterpret_cast<float4*>(part + 4 * tid) = s;
        __syncthreads();
        for (int aa = tid; aa < A; aa += NT) {
            float acc = a.b_att[aa];
            for (int p = 0; p < kparts; ++p) acc += part[4 * (p * na4 + (aa >> 2)) + (aa & 3)];
            ys[aa] = acc;
        }
        __syncthreads();
    }
    // ---- tanh backward: ds = de*v*(1-th^2); dhf += ds; dy[a] = sum_tau ds; dv[a] += sum_tau de*th
    // lane kq owns float4 chunks a4 = kq, kq+16, ... (<= 4 chunks => A <= 256); PP passes in flight
    {
        constexpr int PP = 4, CA = 4;
        float4 dyl[CA], dvl[CA];
#pragma unroll
        for (int c = 0; c < CA; ++c) { dyl[c] = make_float4(0.f, 0.f, 0.f, 0.f); dvl[c] = dyl[c]; }
        // A <= 128: 2 chunks per lane, 4 passes in flight from registers (one instantiation per case: an array that only one
        // branch fills stays a stack object -- 272 bytes of scratch stores per lane and pass that nothing read back)
        auto passes = [&](auto two_c) {
            constexpr bool two = decltype(two_c)::value;
            for (int t0 = 0; t0 < L; t0 += RW * PP) {
                float4 hv[PP][2], gv[PP][2];
#pragma unroll
                for (int p = 0; p < PP; ++p) {
                    const int tau = t0 + p * RW + rr;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int a4 = kq + 16 * c;
                        const bool ok = two && tau < L && a4 < (A >> 2);
                        const size_t off = ((size_t)b * Te + (ok ? tau : 0)) * A + 4 * (ok ? a4 : 0);
                        hv[p][c] = ok ? *reinterpret_cast<const float4*>(a.hf + off) : make_float4(0.f, 0.f, 0.f, 0.f);
                        gv[p][c] = ok ? *reinterpret_cast<const float4*>(a.dhf + off) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int p = 0; p < PP; ++p) {
                    const int tau = t0 + p * RW + rr;
                    if (tau < L) {
                        const float de = al[tau];
#pragma unroll
                        for (int c = 0; c < (two ? 2 : CA); ++c) {
                            const int a4 = kq + 16 * c;
                            if (a4 < (A >> 2)) {
                                float* gp = a.dhf + ((size_t)b * Te + tau) * A + 4 * a4;
                                float4 h4, g4;
                                if constexpr (two) { h4 = hv[p][c & 1]; g4 = gv[p][c & 1]; }
                                else { h4 = *reinterpret_cast<const float4*>(a.hf + ((size_t)b * Te + tau) * A + 4 * a4);
                                       g4 = *reinterpret_cast<float4*>(gp); }
                                const float4 yv = *reinterpret_cast<const float4*>(ys + 4 * a4);
                                const float4 vv = *reinterpret_cast<const float4*>(a.v + 4 * a4);
                                float th, ds;
#define ASR_TB(f) th = fast_tanh(h4.f + yv.f); ds = de * vv.f * (1.f - th * th); g4.f += ds; dyl[c].f += ds; dvl[c].f = fmaf(de, th, dvl[c].f);
                                ASR_TB(x) ASR_TB(y) ASR_TB(z) ASR_TB(w)
#undef ASR_TB
                                *reinterpret_cast<float4*>(gp) = g4;
                            }
                        }
                    }
                }
            }
        };
        if ((A >> 2) <= 32) passes(std::true_type{}); else passes(std::false_type{});
        // reduce the RW DPP rows of the block through LDS, fixed order
        __syncthreads();
        float* pdy = part;                 // [RW rows][A]
        float* pdv = part + RW * A;
#pragma unroll
        for (int c = 0; c < CA; ++c) {
            const int a4 = kq + 16 * c;
            if (a4 < (A >> 2)) {
                *reinterpret_cast<float4*>(pdy + rr * A + 4 * a4) = dyl[c];
                *reinterpret_cast<float4*>(pdv + rr * A + 4 * a4) = dvl[c];
            }
        }
        __syncthreads();
        for (int aa = tid; aa < A; aa += NT) {
            float sy = 0.f, sv = 0.f;
            for (int r = 0; r < RW; ++r) { sy += pdy[r * A + aa]; sv += pdv[r * A + aa]; }
            dys[aa] = sy;
            a.dy[(size_t)b * A + aa] = sy;
            a.dv_part[(size_t)b * A + aa] += sv;
        }
        __syncthreads();
    }
    // ---- dq = dq_ap + dy.W_att^T + dc_carry : NT/H threads per k share the A range through LDS
    {
        const int tpk = max(1, NT / H);               // threads per output k
        const int k = tid / tpk, sub = tid % tpk;
        float s0 = 0.f;
        if (k < H) {
            const int na4 = A >> 2;
            const int c0 = sub * ((na4 + tpk - 1) / tpk), c1 = min(na4, c0 + (na4 + tpk - 1) / tpk);
            const float* wr = a.w_att + (size_t)k * A;
            constexpr int PB = 16;
            for (int cb = c0; cb < c1; cb += PB) {
                float4 wv[PB];
#pragma unroll
                for (int i = 0; i < PB; ++i)
                    wv[i] = (cb + i < c1) ? *reinterpret_cast<const float4*>(wr + 4 * (cb + i)) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < PB; ++i)
                    if (cb + i < c1) {
                        const float4 y0 = *reinterpret_cast<const float4*>(dys + 4 * (cb + i));
                        s0 = fmaf(wv[i].x, y0.x, s0); s0 = fmaf(wv[i].y, y0.y, s0); s0 = fmaf(wv[i].z, y0.z, s0); s0 = fmaf(wv[i].w, y0.w, s0);
                    }
            }
        }
        part[tid] = s0;
        __syncthreads();
        // ---- outer cell pointwise backward (one thread per unit)
        for (int kk = tid; kk < H; kk += NT) {
            float dq_att = 0.f;
            if (kk * tpk + tpk <= NT) { for (int i = 0; i < tpk; ++i) dq_att += part[kk * tpk + i]; }
            else {       // H > NT: recompute serially (not hit at the supported sizes)
                const float* wr = a.w_att + (size_t)kk * A;
                for (int aa = 0; aa < A; ++aa) dq_att = fmaf(wr[aa], dys[aa], dq_att);
            }
            if (!a.gates) { a.dq_out[(size_t)b * H + kk] = a.dqc[(size_t)b * (H + D) + kk] + dq_att; continue; }     // attention only
            const float dq = a.dqc[(size_t)b * (H + D) + kk] + dq_att + a.dc_carry[(size_t)b * H + kk];
            const float dh = a.dh_carry ? a.dh_carry[(size_t)b * a.ld_dh + kk] : 0.f;
            float* gp = a.gates + (size_t)b * 4 * H + kk;
            const float gi = gp[0], gj = gp[H], gf = gp[2 * H], go = gp[3 * H];
            const float cp = a.c_prev ? a.c_prev[(size_t)b * H + kk] : 0.f;
            const float tc = fast_tanh(qs[kk]);
            const float dct = dq + dh * go * (1.f - tc * tc);
            gp[0] = dct * gj * gi * (1.f - gi);
            gp[H] = dct * gi * (1.f - gj * gj);
            gp[2 * H] = dct * cp * gf * (1.f - gf);
            gp[3 * H] = dh * tc * go * (1.f - go);
            a.dc_carry[(size_t)b * H + kk] = dct * gf;
        }
    }
}

struct LmBwdArgs {
    float* gates; const float* c; const float* c_prev;      // [B][4H] (in: gates, out: dG), [B][H], [B][H]|null
    const float* dlo; int ld_dlo;                            // grad w.r.t. the (dropped) lm output
    const float* dh_carry; int ld_dh;                        // dEH[i+1][:, E:] or nullptr
    float* dc_carry;                                         // [B][H] in/out
    int B, H; float keep; uint32_t seed; uint32_t step;
};

__global__ __launch_bounds__(256) void lm_cell_bwd_kernel(LmBwdArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.B * a.H) return;
    const int b = idx / a.H, k = idx % a.H, H = a.H;
    float dh = a.dlo[(size_t)b * a.ld_dlo + k] * keep_scale(a.seed, a.step * (uint32_t)a.B + (uint32_t)b, (uint32_t)k, a.keep);
    if (a.dh_carry) dh += a.dh_carry[(size_t)b * a.ld_dh + k];
    float* gp = a.gates + (size_t)b * 4 * H + k;
    const float gi = gp[0], gj = gp[H], gf = gp[2 * H], go = gp[3 * H];
    const float cp = a.c_prev ? a.c_prev[(size_t)b * H + k] : 0.f;
    const float tc = fast_tanh(a.c[(size_t)b * H + k]);
    const float dct = a.dc_carry[(size_t)b * H + k] + dh * go * (1.f - tc * tc);
    gp[0] = dct * gj * gi * (1.f - gi);
    gp[H] = dct * gi * (1.f - gj * gj);
    gp[2 * H] = dct * cp * gf * (1.f - gf);
    gp[3 * H] = dh * tc * go * (1.f - go);
    a.dc_carry[(size_t)b * H + k] = dct * gf;
}

}  // namespace asr

extern "C" int asr_colsum_f32(void*, const float*, int, int, int, float*, int);
extern "C" int asr_decoder_chain_supported(int B, int Te, int D, int A, int H);
extern "C" int asr_decoder_chain_bwd_rows(int Te, int D, int A, int H);
extern "C" int asr_decoder_lm_chain_supported(int B, int lmH);
int asr_lstm_rec_bwd_tm(hipStream_t s, float* gates, const float* act, const float* dout, int ldo, const float* kh,
                        const int* full_len, void* hx_ws, int* err, int B, int T, int H, float keep, unsigned seed);
bool asr_decoder_chain_bwd_fits(int Te, int D, int A, int H);
int asr_decoder_chain_bwd(void* stream, float* gates, const float* dec_c, const float* alpha, const float* y,
                          const float* ctx, const float* dqc, const float* wh, const float* wc, const float* w_att,
                          const float* v, const float* hf, const float* enc, const int* enc_len, float* dY, float* dctx,
                          float* dhf, float* dv_part, void* ws, int* err, int B, int Te, int D, int A, int H, int T);
extern "C" int asr_gather_rows(void*, const float*, const int*, float*, int, int);
extern "C" int asr_scatter_add_rows_ld(void*, float*, const int*, const float*, int, int, int);
extern "C" int asr_scatter_add_rows_ordered(void*, float*, int, const int*, const float*, int, int, int);

static size_t dec_bwd_lds(int Te, int H, int A, int D) {
    auto r4 = [](int x) { return (size_t)((x + 3) & ~3); };
    const size_t part = std::max<size_t>(512 * 4, std::max<size_t>(64 * (size_t)A, r4(Te)));
    return sizeof(float) * (r4(D) + r4(Te) + 2 * r4(A) + r4(H) + part);
}

// The LM cell chain's backward (BPTT over all steps, demb, the LM kernel / bias / embedding gradients).  `ss`: the stream it runs
// on -- the library's side stream, already ordered behind the decoder chain by the caller (asr_attn_decoder_bwd, as in rounds
// 1-4), or the caller's own stream later (asr_attn_decoder_bwd_lm, the deferred form).  fork: make `ss` wait for `s` first.
// *e_lm_bptt: event behind the persistent BPTT launch (NULL if the per-step path ran).
static int dec_lm_chain_bwd(hipStream_t s, hipStream_t ss, bool fork, const asr_dec_weights* w, const asr_dec_weights* g,
                            const asr_dec_dims* d, const asr_dec_ws* ws, const asr_dec_bwd_ws* bw, float keep_lm, unsigned seed,
                            hipEvent_t* e_lm_bptt_out) {
    using namespace asr;
    const int B = d->B, Te = d->Te, D = d->D, A = d->A, H = d->H, lmH = d->lmH, E = d->E, V = d->V, T = d->T_out;
    const int TB = T * B;
    const int P = w->simple_w ? H : lmH;
    const int ldLC = P + D, ldEH = E + lmH;
    void* side = static_cast<void*>(ss);
    hipEvent_t e_lm_bptt = nullptr;
    int rc;
    (void)V;
    // ---- LM chain backward on the side stream: it needs only dLC[i] (all produced above) and its
    // own carries, so it runs concurrently with the data-gradient GEMMs the caller's stream does next; the caller's stream
    // waits for its BPTT before this function returns (one persistent kernel at a time, see the end), the remaining
    // side-stream GEMMs overlap the encoder's BPTT.  asr_side_join() orders them before the optimizer.
    if (fork) {
        hipEvent_t e_loop = next_event();
        if (hipEventRecord(e_loop, s) != hipSuccess || hipStreamWaitEvent(ss, e_loop, 0) != hipSuccess) return ASR_ELAUNCH;
    }
    if (hipMemsetAsync(bw->dc_lm, 0, sizeof(float) * B * lmH, ss) != hipSuccess) return ASR_ELAUNCH;
    // persistent LM chain (the forward ran csrc/lstm.hip time-major under the same predicate): one BPTT launch
    // over all steps (dG overwrites lm_gates), then demb = dG . K_x^T for all steps as one GEMM
    const bool lm_chain = ws->chain_ws && ws->w2k && ws->err && ws->y && asr_decoder_chain_supported(B, Te, D, A, H) &&
                          ws->lm_act && ws->lm_hprev && ws->lm_state && ws->lm_len && ws->lm_hx && bw->lm_hx &&
                          asr_decoder_lm_chain_supported(B, lmH);
    if (lm_chain) {
        const float* dlo = bw->dLC; int ld_dlo = ldLC;
        if (w->simple_w) {
            if ((rc = asr_gemm_f32(side, 0, 1, TB, lmH, H, bw->dLC, ldLC, w->simple_w, H, bw->dlm, lmH, nullptr, 0))) return rc;
            dlo = bw->dlm; ld_dlo = lmH;
        }
        if ((rc = asr_lstm_rec_bwd_tm(ss, ws->lm_gates, ws->lm_act, dlo, ld_dlo, w->lm_kernel + (size_t)E * 4 * lmH, ws->lm_len,
                                      bw->lm_hx, ws->err, B, T, lmH, keep_lm, seed))) return rc;
        e_lm_bptt = next_event();
        if (hipEventRecord(e_lm_bptt, ss) != hipSuccess) return ASR_ELAUNCH;
        if ((rc = asr_gemm_f32(side, 0, 1, TB, E, 4 * lmH, ws->lm_gates, 4 * lmH, w->lm_kernel, 4 * lmH, bw->dEH, ldEH, nullptr, 0)))
            return rc;
    }
    for (int i = T - 1; i >= 0 && !lm_chain; --i) {
        const size_t o = (size_t)i * B;
        const bool last = i == T - 1;
        const float* dlo = bw->dLC + o * ldLC; int ld_dlo = ldLC;
        if (w->simple_w) {
            if ((rc = asr_linear_wt_fwd(side, bw->dLC + o * ldLC, ldLC, H, w->simple_w, H, bw->dlm + o * lmH, lmH, B, lmH, 0)))
                return rc;
            dlo = bw->dlm + o * lmH; ld_dlo = lmH;
        }
        LmBwdArgs l;
        l.gates = ws->lm_gates + o * 4 * lmH; l.c = ws->lm_c + o * lmH; l.c_prev = i ? ws->lm_c + (o - B) * lmH : nullptr;
        l.dlo = dlo; l.ld_dlo = ld_dlo;
        l.dh_carry = last ? nullptr : bw->dEH + (o + B) * ldEH + E; l.ld_dh = ldEH;
        l.dc_carry = bw->dc_lm; l.B = B; l.H = lmH; l.keep = keep_lm; l.seed = seed; l.step = (uint32_t)i;
        hipLaunchKernelGGL(lm_cell_bwd_kernel, dim3((B * lmH + 255) / 256), dim3(256), 0, ss, l);
        // [demb | dlm_h_prev] = dG_lm . K_lm^T
        if ((rc = asr_linear_wt_fwd(side, ws->lm_gates + o * 4 * lmH, 4 * lmH, 4 * lmH, w->lm_kernel, 4 * lmH,
                                    bw->dEH + o * ldEH, ldEH, B, E + lmH, 0))) return rc;
    }
    {
        float* gwl = nullptr;
        if ((rc = asr_gather_rows(side, w->embedding, ws->tok, bw->emb_all, TB, E))) return rc;
        gwl = const_cast<float*>(g->lm_kernel);
        if ((rc = asr_gemm_f32(side, 1, 0, E, 4 * lmH, TB, bw->emb_all, E, ws->lm_gates, 4 * lmH, gwl, 4 * lmH, nullptr, 1))) return rc;
        if (lm_chain) {      // h_{t-1} of every step was saved by the recurrent kernel (row 0 = zeros)
            if ((rc = asr_gemm_f32(side, 1, 0, lmH, 4 * lmH, TB, ws->lm_hprev, lmH, ws->lm_gates, 4 * lmH,
                                   gwl + (size_t)E * 4 * lmH, 4 * lmH, nullptr, 1))) return rc;
        } else if (T > 1 && (rc = asr_gemm_f32(side, 1, 0, lmH, 4 * lmH, TB - B, ws->lm_h, lmH, ws->lm_gates + (size_t)B * 4 * lmH, 4 * lmH,
                                        gwl + (size_t)E * 4 * lmH, 4 * lmH, nullptr, 1))) return rc;
        if ((rc = asr_colsum_f32(side, ws->lm_gates, 4 * lmH, TB, 4 * lmH, const_cast<float*>(g->lm_bias), 1))) return rc;
        if (asr::wgrad_slabs() == 1 && E <= 1024) {      // occurrences of a token added in ascending order, no atomics (csrc/splitk.hip)
            if ((rc = asr_scatter_add_rows_ordered(side, const_cast<float*>(g->embedding), V, ws->tok, bw->dEH, TB, E, ldEH))) return rc;
        } else if ((rc = asr_scatter_add_rows_ld(side, const_cast<float*>(g->embedding), ws->tok, bw->dEH, TB, E, ldEH))) return rc;
    }
    if (e_lm_bptt_out) *e_lm_bptt_out = e_lm_bptt;
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

// Gradients are ACCUMULATED into `g` (same field layout as the weights) and into denc
// [B,Te,D]; dec_gates / lm_gates in `ws` are overwritten with the pre-activation gradients.
extern "C" int asr_attn_decoder_bwd(void* stream, const asr_dec_weights* w, const asr_dec_weights* g,
                                    const asr_dec_dims* d, const asr_dec_ws* ws, const asr_dec_bwd_ws* bw,
                                    const float* enc, const int* enc_len, const float* dlogits,
                                    float* denc, float keep_lm, unsigned seed) {
    using namespace asr;
    if (!w || !g || !d || !ws || !bw || !enc || !enc_len || !dlogits || !denc) return ASR_EINVAL;
    const int B = d->B, Te = d->Te, D = d->D, A = d->A, H = d->H, lmH = d->lmH, E = d->E, V = d->V, T = d->T_out;
    if ((A & 3) || (D & 3) || A > 256) return ASR_EUNSUPPORTED;
    const size_t lds = dec_bwd_lds(Te, H, A, D);
    if (lds > 150 * 1024) return ASR_EUNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int TB = T * B;
    const int P = w->simple_w ? H : lmH;
    int rc;
    prof_begin(ASR_PROF_DECODER_BWD, s);
    hipStream_t ss = side_stream();
    void* side = static_cast<void*>(ss);
    hipEvent_t e_lm_bptt = nullptr;
    hipEvent_t e_fork = next_event();
    if (hipEventRecord(e_fork, s) != hipSuccess || hipStreamWaitEvent(ss, e_fork, 0) != hipSuccess) return ASR_ELAUNCH;
    // ---- what the chain needs besides dQC -- the composed context weight wc = W_inp[P:] . K_x (weights only) and three zeroed
    // accumulators -- goes to the side stream, next to the two data-gradient products below instead of in a row behind them
    // (ASR_DEC_SIDE_SMALL=0: on the caller's stream)
    // bw->side_busy: the side stream still holds the previous decoder's weight gradients (config 4: the phone chain waited
    // 150 us for these 30 us of work) -- then on the caller's stream too
    static const bool side_small_env = [] { const char* e = getenv("ASR_DEC_SIDE_SMALL"); return !(e && e[0] == '0'); }();
    const bool side_small = side_small_env && !bw->side_busy;
    const bool chain_ok = bw->chain_ws && bw->wc && ws->y && ws->err &&
                          asr_decoder_chain_supported(B, Te, D, A, H) && asr_decoder_chain_bwd_fits(Te, D, A, H);
    hipEvent_t e_small = nullptr;
    {
        hipStream_t zs = side_small ? ss : s;
        if (hipMemsetAsync(bw->dc_dec, 0, sizeof(float) * B * H, zs) != hipSuccess) return ASR_ELAUNCH;
        if (hipMemsetAsync(bw->dhf, 0, sizeof(float) * (size_t)B * Te * A, zs) != hipSuccess) return ASR_ELAUNCH;
        if (hipMemsetAsync(bw->dv_part, 0, sizeof(float) * B * A, zs) != hipSuccess) return ASR_ELAUNCH;
        if (chain_ok && side_small &&
            (rc = asr_gemm_f32(side, 0, 0, D, 4 * H, E, w->inp_w + (size_t)P * E, E, w->dec_kernel, 4 * H, bw->wc, 4 * H, nullptr, 0)))
            return rc;
        if (side_small) {
            e_small = next_event();
            if (hipEventRecord(e_small, ss) != hipSuccess) return ASR_ELAUNCH;
        }
    }
    // ---- hoisted data gradients: dP = dLogits.W_out^T ; dQC = dP.W_ap^T
    if ((rc = asr_gemm_f32(stream, 0, 1, TB, H, V, dlogits, V, w->out_w, V, bw->dP, H, nullptr, 0))) return rc;
    if ((rc = asr_gemm_f32(stream, 0, 1, TB, H + D, H, bw->dP, H, w->ap_w, H, bw->dQC, H + D, nullptr, 0))) return rc;
    {   // the weight gradients that need only dLogits / dP run on the side stream UNDER the (latency-bound) chain below
        // (dLogits and p are complete when this call starts: the fork at its top orders the side stream behind them.  A second
        //  fork here only delayed the product behind dP / dQC -- and every fork costs the recording stream 10-20 us,
        //  scripts/micro/event_cost.py; ASR_DEC_FORK_PRE=1 restores it)
        static const bool fork_pre = [] { const char* e = getenv("ASR_DEC_FORK_PRE"); return e && e[0] == '1'; }();
        if (fork_pre) {
            hipEvent_t e_pre = next_event();
            if (hipEventRecord(e_pre, s) != hipSuccess || hipStreamWaitEvent(ss, e_pre, 0) != hipSuccess) return ASR_ELAUNCH;
        }
        auto wg0 = [&](int M, int N, int K, const float* Ap, int lda, const float* Bp, int ldb, float* C) {
            return asr_gemm_f32(side, 1, 0, M, N, K, Ap, lda, Bp, ldb, C, N, nullptr, 1);
        };
        float* gw0 = const_cast<float*>(g->out_w);                       // OutputProjection
        if ((rc = wg0(H, V, TB, ws->p, H, dlogits, V, gw0))) return rc;
    }
    if (e_small && hipStreamWaitEvent(s, e_small, 0) != hipSuccess) return ASR_ELAUNCH;
    const int ldXH = E + H, ldLC = P + D, ldEH = E + lmH;
    // ---- persistent chain (csrc/decoder_chain_bwd.hip): the whole reverse-time recursion in one launch
    const bool use_chain = chain_ok;
    int dv_rows = B;
    if (use_chain) {
        // wc = W_inp[P:] . K_x : the context rows of the composed input weight (decoder.hip, forward chain)
        if (!side_small &&
            (rc = asr_gemm_f32(stream, 0, 0, D, 4 * H, E, w->inp_w + (size_t)P * E, E, w->dec_kernel, 4 * H, bw->wc, 4 * H, nullptr, 0)))
            return rc;
        if ((rc = asr_decoder_chain_bwd(stream, ws->dec_gates, ws->dec_c, ws->alpha, ws->y, ws->ctx, bw->dQC,
                                        w->dec_kernel + (size_t)E * 4 * H, bw->wc, w->attn_w, w->attn_v, ws->hf, enc, enc_len,
                                        bw->dY, bw->dctx, bw->dhf, bw->dv_part, bw->chain_ws, ws->err, B, Te, D, A, H, T)))
            return rc;
        { const int cr = asr_decoder_chain_bwd_rows(Te, D, A, H); dv_rows = ((B + cr - 1) / cr) * 16; }      // one partial per workgroup
        // dx = dG . K_x^T for all steps, then dlm_out = dx . W_inp[:P]^T (the dh / dctx carries stayed on chip)
        if ((rc = asr_gemm_f32(stream, 0, 1, TB, E, 4 * H, ws->dec_gates, 4 * H, w->dec_kernel, 4 * H, bw->dXH, ldXH, nullptr, 0)))
            return rc;
        if ((rc = asr_gemm_f32(stream, 0, 1, TB, P, E, bw->dXH, ldXH, w->inp_w, E, bw->dLC, ldLC, nullptr, 0))) return rc;
    }
    for (int i = T - 1; i >= 0 && !use_chain; --i) {
        const size_t o = (size_t)i * B;
        const bool last = i == T - 1;
        DecBwdStepArgs a;
        a.q = ws->dec_c + o * H; a.w_att = w->attn_w; a.b_att = w->attn_b; a.v = w->attn_v;
        a.hf = ws->hf; a.enc = enc; a.enc_len = enc_len; a.alpha = ws->alpha + o * Te;
        a.y_saved = ws->y ? ws->y + o * A : nullptr;
        a.dqc = bw->dQC + o * (H + D);
        a.dctx_carry = last ? nullptr : bw->dLC + (o + B) * ldLC + P; a.ld_carry = ldLC;
        a.dhf = bw->dhf; a.dctx_out = bw->dctx + o * D; a.dy = bw->dY + o * A; a.dv_part = bw->dv_part;
        a.gates = ws->dec_gates + o * 4 * H;
        a.c_prev = i ? ws->dec_c + (o - B) * H : nullptr;
        a.dh_carry = last ? nullptr : bw->dXH + (o + B) * ldXH + E; a.ld_dh = ldXH;
        a.dc_carry = bw->dc_dec; a.dq_out = nullptr;
        a.B = B; a.Te = Te; a.H = H; a.A = A; a.D = D;
        hipLaunchKernelGGL(dec_attn_cell_bwd_kernel, dim3(B), dim3(DBW_NT), lds, s, a);
        // [dx | dh_prev] = dG_dec . K_dec^T
        if ((rc = asr_linear_wt_fwd(stream, ws->dec_gates + o * 4 * H, 4 * H, 4 * H, w->dec_kernel, 4 * H,
                                    bw->dXH + o * ldXH, ldXH, B, E + H, 0))) return rc;
        // [dlm_out | dctx_prev] = dx . W_inp^T
        if ((rc = asr_linear_wt_fwd(stream, bw->dXH + o * ldXH, ldXH, E, w->inp_w, E, bw->dLC + o * ldLC, ldLC, B, P + D, 0)))
            return rc;
    }
    // ---- LM chain backward: on the side stream now (the caller's stream then waits for its BPTT before this function returns:
    // one persistent kernel at a time, see the end), or -- bw->lm_deferred -- not at all here: the caller runs it with
    // asr_attn_decoder_bwd_lm on its own stream behind the encoder's last BPTT, where the LM chain's BPTT (128 workgroups since
    // round 5) shares the chip with the side stream's weight-gradient backlog instead of holding up the encoder's BPTT.
    // (the side stream waits HERE for the backward chain in either case: the weight-gradient GEMMs queued on it further down
    // read dY, dG, dXH ...; with the LM part deferred and this wait inside it they ran on unfinished data -- found by
    // test_decoder_chain_path_vs_oracle_and_autograd, not by the bench)
    {
        hipEvent_t e_loop = next_event();
        if (hipEventRecord(e_loop, s) != hipSuccess || hipStreamWaitEvent(ss, e_loop, 0) != hipSuccess) return ASR_ELAUNCH;
    }
    if (!bw->lm_deferred && (rc = dec_lm_chain_bwd(s, ss, false, w, g, d, ws, bw, keep_lm, seed, &e_lm_bptt))) return rc;
    if (hipGetLastError() != hipSuccess) return ASR_ELAUNCH;
    // ---- the encoder-state gradient is what the caller's stream needs next: do it first, there
    // denc[b] += sum_i alpha_i[b,:]^T . dctx_i[b,:]  -- one batched GEMM over the B utterances
    if ((rc = asr_gemm_f32_batched(stream, 1, 0, Te, D, T, ws->alpha, B * Te, Te, bw->dctx, B * D, D, denc, D,
                                   (long long)Te * D, nullptr, 1, B))) return rc;
    if ((rc = asr_gemm_f32(stream, 0, 1, B * Te, D, A, bw->dhf, A, w->attn_enc_w, A, denc, D, nullptr, 1))) return rc;
    // ---- weight gradients: X^T . dY over all steps (accumulate into g), on the side stream
    void* stream_w = side;
    auto wgrad = [&](int M, int N, int K, const float* Ap, int lda, const float* Bp, int ldb, float* C) {
        return asr_gemm_f32(stream_w, 1, 0, M, N, K, Ap, lda, Bp, ldb, C, N, nullptr, 1);
    };
    float* gw = nullptr;
    // (the OutputProjection weight gradient was issued before the chain, see above; the AttnProjection's and the bias column sums come
    // here: next to the persistent chain a colsum's 1 920 small workgroups crawled for the chain's whole 770 us -- 6.8 % of the
    // GPU time in the kernel statistics for 16 us of work -- and held back the side-stream GEMMs queued behind them)
    gw = const_cast<float*>(g->ap_w);                                    // AttnProjection: rows [q | ctx] (likewise moved here)
    if ((rc = wgrad(H, H, TB, ws->dec_c, H, bw->dP, H, gw))) return rc;
    if ((rc = wgrad(D, H, TB, ws->ctx, D, bw->dP, H, gw + (size_t)H * H))) return rc;
    if ((rc = asr_colsum_f32(stream_w, dlogits, V, TB, V, const_cast<float*>(g->out_b), 1))) return rc;
    if ((rc = asr_colsum_f32(stream_w, bw->dP, H, TB, H, const_cast<float*>(g->ap_b), 1))) return rc;
    // Attention query projection, AttnV
    if ((rc = wgrad(H, A, TB, ws->dec_c, H, bw->dY, A, const_cast<float*>(g->attn_w)))) return rc;
    if ((rc = asr_colsum_f32(stream_w, bw->dY, A, TB, A, const_cast<float*>(g->attn_b), 1))) return rc;
    if ((rc = asr_colsum_f32(stream_w, bw->dv_part, A, dv_rows, A, const_cast<float*>(g->attn_v), 1))) return rc;
    // outer cell kernel: rows [x | h_prev]
    gw = const_cast<float*>(g->dec_kernel);
    if ((rc = wgrad(E, 4 * H, TB, ws->x, E, ws->dec_gates, 4 * H, gw))) return rc;
    if (T > 1 && (rc = wgrad(H, 4 * H, TB - B, ws->dec_h, H, ws->dec_gates + (size_t)B * 4 * H, 4 * H, gw + (size_t)E * 4 * H))) return rc;
    if ((rc = asr_colsum_f32(stream_w, ws->dec_gates, 4 * H, TB, 4 * H, const_cast<float*>(g->dec_bias), 1))) return rc;
    // InputProjection: rows [lm_out' | ctx_prev]
    const float* lo = w->simple_w ? ws->sp : (keep_lm < 1.0f ? ws->lm_hd : ws->lm_h);
    gw = const_cast<float*>(g->inp_w);
    if ((rc = wgrad(P, E, TB, lo, P, bw->dXH, ldXH, gw))) return rc;
    if (T > 1 && (rc = wgrad(D, E, TB - B, ws->ctx, D, bw->dXH + (size_t)B * ldXH, ldXH, gw + (size_t)P * E))) return rc;
    if ((rc = asr_colsum_f32(stream_w, bw->dXH, ldXH, TB, E, const_cast<float*>(g->inp_b), 1))) return rc;
    if (w->simple_w) {
        const float* lmo = keep_lm < 1.0f ? ws->lm_hd : ws->lm_h;
        if ((rc = wgrad(lmH, H, TB, lmo, lmH, bw->dLC, ldLC, const_cast<float*>(g->simple_w)))) return rc;
        if ((rc = asr_colsum_f32(stream_w, bw->dLC, ldLC, TB, H, const_cast<float*>(g->simple_b), 1))) return rc;
    }
    // AttnW and the encoder-state gradient through hf = enc.AttnW
    if ((rc = wgrad(D, A, B * Te, enc, D, bw->dhf, A, const_cast<float*>(g->attn_enc_w)))) return rc;
    {
        hipEvent_t e_done = next_event();
        if (hipEventRecord(e_done, ss) != hipSuccess) return ASR_ELAUNCH;
        set_pending_join(e_done);
    }
    // Never two persistent kernels in flight from two streams: whatever the caller launches next (the encoder's BPTT) waits for
    // the LM chain's BPTT on the side stream.  Each needs all of its workgroups resident, and with workgroup groups spread
    // over the XCDs (a group count that is not a multiple of 8, e.g. 30 utterances) the two starved each other's dispatch:
    // 1 step in ~100 ended in the 2-second exchange time-out under scripts/soak_fixed.py 30 83 27.  The overlap bought
    // nothing (both are latency-bound on the same CUs: 331 us together, ~150 + ~150 apart).
    static const bool lm_overlap = [] { const char* e = getenv("ASR_LM_BPTT_OVERLAP"); return e && e[0] == '1'; }();   // (measurement only: UNSAFE)
    if (e_lm_bptt && !lm_overlap && hipStreamWaitEvent(s, e_lm_bptt, 0) != hipSuccess) return ASR_ELAUNCH;
    prof_end(ASR_PROF_DECODER_BWD, s);
    return ASR_OK;
}

// The deferred LM-chain part of asr_attn_decoder_bwd (bw->lm_deferred = 1 there): everything on `stream`.  Call it after the
// encoder's backward pass was enqueued on the same stream (never next to another persistent kernel) and before asr_side_join.
extern "C" int asr_attn_decoder_bwd_lm(void* stream, const asr_dec_weights* w, const asr_dec_weights* g, const asr_dec_dims* d,
                                       const asr_dec_ws* ws, const asr_dec_bwd_ws* bw, float keep_lm, unsigned seed) {
    if (!w || !g || !d || !ws || !bw) return ASR_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dec_lm_chain_bwd(s, s, false, w, g, d, ws, bw, keep_lm, seed, nullptr);
}

// ---------------------------------------------------------------------------------------------
// Step-level backward entry points (C ABI): the two per-step kernels of the launch-based path above, for callers that
// compose their own decoder loop on the host -- e2e_asr_amd/multi_decoder.py (MultiRNNCell decoders, decoder.py:66-68).
// ---------------------------------------------------------------------------------------------
// Pointwise backward of one BasicLSTMCell step whose OUTPUT went through DropoutWrapper(output_keep_prob = keep):
// dh = dout * mask(seed, step, row, unit) + dh_carry;  gates [B][4H] holds the activated i,j,f,o on entry and dG (gradient of
// the pre-activations) on return; dc_carry [B][H] is read and updated (the cell-state gradient flowing to step - 1).
extern "C" int asr_lstm_cell_bwd(void* stream, float* gates, const float* c, const float* c_prev, const float* dout, int ld_dout,
                                 const float* dh_carry, int ld_dh, float* dc_carry, int B, int H, float keep, unsigned seed,
                                 unsigned step) {
    using namespace asr;
    if (!gates || !c || !dout || !dc_carry || B <= 0 || H <= 0 || ld_dout < H || (dh_carry && ld_dh < H)) return ASR_EINVAL;
    LmBwdArgs l;
    l.gates = gates; l.c = c; l.c_prev = c_prev; l.dlo = dout; l.ld_dlo = ld_dout; l.dh_carry = dh_carry; l.ld_dh = ld_dh;
    l.dc_carry = dc_carry; l.B = B; l.H = H; l.keep = keep; l.seed = seed; l.step = step;
    hipLaunchKernelGGL(lm_cell_bwd_kernel, dim3((B * H + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), l);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// One step of the attention + query-cell backward (kernel K1 above): attention backward for the query q = c of the cell whose
// activated gates are in `gates` (dG on return), dq folded into that cell's state gradient.  dqc [B][H+D] = this step's
// [dq | dctx] from the AttnProjection; dctx_carry (row stride ld_carry) / dh_carry (ld_dh) = what step + 1 sent back, or NULL;
// dhf [B][Te][A] and dv_part [B][A] are accumulated over steps (zero them first); dctx_out [B][D], dy [B][A] are this step's.
extern "C" int asr_attn_cell_bwd(void* stream, const float* q, const float* w_att, const float* b_att, const float* v,
                                 const float* hf, const float* enc, const int* enc_len, const float* alpha, const float* y_saved,
                                 const float* dqc, const float* dctx_carry, int ld_carry, float* dhf, float* dctx_out, float* dy,
                                 float* dv_part, float* gates, const float* c_prev, const float* dh_carry, int ld_dh,
                                 float* dc_carry, int B, int Te, int H, int A, int D) {
    using namespace asr;
    if (!q || !w_att || !b_att || !v || !hf || !enc || !enc_len || !alpha || !dqc || !dhf || !dctx_out || !dy || !dv_part ||
        !gates || !dc_carry || B <= 0 || Te <= 0) return ASR_EINVAL;
    if ((A & 3) || (D & 3) || A > 256) return ASR_EUNSUPPORTED;
    const size_t lds = dec_bwd_lds(Te, H, A, D);
    if (lds > 150 * 1024) return ASR_EUNSUPPORTED;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_attn_cell_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    DecBwdStepArgs a;
    a.q = q; a.w_att = w_att; a.b_att = b_att; a.v = v; a.hf = hf; a.enc = enc; a.enc_len = enc_len; a.alpha = alpha;
    a.y_saved = y_saved; a.dqc = dqc; a.dctx_carry = dctx_carry; a.ld_carry = ld_carry; a.dhf = dhf; a.dctx_out = dctx_out;
    a.dy = dy; a.dv_part = dv_part; a.gates = gates; a.c_prev = c_prev; a.dh_carry = dh_carry; a.ld_dh = ld_dh;
    a.dc_carry = dc_carry; a.dq_out = nullptr; a.B = B; a.Te = Te; a.H = H; a.A = A; a.D = D;
    hipLaunchKernelGGL(dec_attn_cell_bwd_kernel, dim3(B), dim3(DBW_NT), lds, static_cast<hipStream_t>(stream), a);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
// The attention backward of asr_attn_cell_bwd ALONE: dq_out [B][H] = dqc[:, :H] + dy . W_att^T, the gradient w.r.t. the query,
// for callers whose query is not an LSTM cell state (the GRU decoder: the query is the GRU state itself, decoder.py:79-80).
extern "C" int asr_attn_bwd(void* stream, const float* q, const float* w_att, const float* b_att, const float* v,
                            const float* hf, const float* enc, const int* enc_len, const float* alpha,
                            const float* dqc, const float* dctx_carry, int ld_carry, float* dhf, float* dctx_out, float* dy,
                            float* dv_part, float* dq_out, int B, int Te, int H, int A, int D) {
    using namespace asr;
    if (!q || !w_att || !b_att || !v || !hf || !enc || !enc_len || !alpha || !dqc || !dhf || !dctx_out || !dy || !dv_part ||
        !dq_out || B <= 0 || Te <= 0) return ASR_EINVAL;
    if ((A & 3) || (D & 3) || A > 256) return ASR_EUNSUPPORTED;
    const size_t lds = dec_bwd_lds(Te, H, A, D);
    if (lds > 150 * 1024) return ASR_EUNSUPPORTED;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_attn_cell_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    DecBwdStepArgs a;
    a.q = q; a.w_att = w_att; a.b_att = b_att; a.v = v; a.hf = hf; a.enc = enc; a.enc_len = enc_len; a.alpha = alpha;
    a.y_saved = nullptr; a.dqc = dqc; a.dctx_carry = dctx_carry; a.ld_carry = ld_carry; a.dhf = dhf; a.dctx_out = dctx_out;
    a.dy = dy; a.dv_part = dv_part; a.gates = nullptr; a.c_prev = nullptr; a.dh_carry = nullptr; a.ld_dh = 0;
    a.dc_carry = nullptr; a.dq_out = dq_out; a.B = B; a.Te = Te; a.H = H; a.A = A; a.D = D;
    hipLaunchKernelGGL(dec_attn_cell_bwd_kernel, dim3(B), dim3(DBW_NT), lds, static_cast<hipStream_t>(stream), a);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
