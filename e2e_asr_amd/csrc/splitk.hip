// Split-K without float atomics -- the DETERMINISTIC mode (opt-in: asr_set_wgrad_mode(1) / ASR_WGRAD_SLABS=1).
//
// The weight gradients of tf.gradients (seq2seq_model.py:148) are X^T . dY products with K = B*T rows and few output tiles; K is
// split over workgroups to fill the chip.  By default the K slices meet in C through float atomics: the summation order --
// hence the last bits of every weight gradient -- changes from run to run.  In this mode slice s of a launch stores its
// partial tile with plain coalesced stores into slab s of an arena, and `slab_reduce_kernel` (whole chip, float4, HBM-bound)
// adds the slabs in ascending s into C; bias column sums run in two stages and the embedding gradient adds a row's occurrences
// in token order: every gradient is BIT-REPRODUCIBLE run to run (tests/test_gpu_parity3.py).
// Measured in the train step (round 5, scripts/ab_slabs.sh, same box): config 2 8.19 vs 7.90 ms, config 3 6.84 vs 6.77, config 4
// 15.8 vs 14.4 -- the ~20 extra small launches per step (reduce, finish) queue on the side stream, which makes no progress
// while a persistent recurrent kernel owns the CUs, so they land in the tail behind the last BPTT; stand-alone the two forms are
// equal (split3 dK_x of layer 2: 162-166 vs 161-179 TF/s).  Hence opt-in.  (Round 4 tried a last-arriver fix-up INSIDE the GEMM:
// the one reading workgroup is bound by one CU's load bandwidth, +25-50 us per launch; a separate kernel is not.)
//
// The arena belongs to the stream: a launch's slabs are consumed by the reduce kernel queued right behind it on the same
// stream, so the next launch on that stream may overwrite them (stream order); two streams never share an arena.
#include "common.h"
#include <cstdlib>
#include <map>
#include <mutex>

namespace asr {

static int g_wgrad_slabs = -1;
int wgrad_slabs() {
    if (g_wgrad_slabs < 0) { const char* e = getenv("ASR_WGRAD_SLABS"); g_wgrad_slabs = e ? atoi(e) : 0; }
    return g_wgrad_slabs;
}

struct Arena { float* p = nullptr; size_t bytes = 0; };
static std::map<std::pair<int, std::pair<hipStream_t, int>>, Arena> g_arenas;
static std::mutex g_arena_mu;

// >= bytes of device memory owned by (current device, stream s, slot `which`); grow-only (growing synchronises the device:
// warm-up only).  Slot 0: K slices / column-sum partials; slot 1: the transposed product of asr_gemm_f32's ragged-N form.
float* slab_arena(hipStream_t s, size_t bytes, int which) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_arena_mu);
    Arena& a = g_arenas[std::make_pair(dev, std::make_pair(s, which))];
    if (a.bytes < bytes) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return nullptr;   // no allocation inside a capture
        if (a.p) { (void)hipDeviceSynchronize(); (void)hipFree(a.p); a.p = nullptr; a.bytes = 0; }
        const size_t want = (bytes + (bytes >> 2) + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);
        void* q = nullptr;
        if (hipMalloc(&q, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        a.p = static_cast<float*>(q); a.bytes = want;
    }
    return a.p;
}

// C[z][row(m)][col(n)] (+)= sum_{s < nsl} slab[((z * nsl_alloc + s) * M + m) * N + n], s ascending.
//   rows: m < mA maps to C row m when m < mA_valid (else dropped: padding rows of a tile-aligned operand image); m >= mA maps to
//   C row mA_valid + (m - mA)  (the [X | Hprev]^T . dG form of csrc/gemm_p3.hip; plain products pass mA = mA_valid = M).
//   cols: colmap[n] when given (unit-major -> gate-major), else n.
__global__ __launch_bounds__(256) void slab_reduce_kernel(float* __restrict__ C, const float* __restrict__ slab, SlabMap q) {
    const int n4 = q.N >> 2;
    const size_t per = (size_t)q.M * n4;
    const size_t total = per * q.batch;
    const size_t mn = (size_t)q.M * q.N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int z = (int)(i / per);
        const size_t r = i - (size_t)z * per;
        const int m = (int)(r / n4), c4 = (int)(r - (size_t)m * n4);
        int row = m;
        if (m < q.mA) { if (m >= q.mA_valid) continue; }
        else row = q.mA_valid + (m - q.mA);
        const float* sp = slab + ((size_t)z * q.nsl_alloc) * mn + (size_t)m * q.N + 4 * c4;
        float4 acc = *reinterpret_cast<const float4*>(sp);
        for (int s = 1; s < q.nsl; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(sp + (size_t)s * mn);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        float* cp = C + (size_t)z * q.zC + (size_t)row * q.ldc;
        if (4 * c4 + 3 >= q.Nvalid) {          // ragged last chunk: the valid columns one by one
            const float av[4] = {acc.x, acc.y, acc.z, acc.w};
            for (int e = 0; e < 4; ++e) {
                const int n = 4 * c4 + e;
                if (n >= q.Nvalid) break;
                float* d = cp + (q.colmap ? q.colmap[n] : n);
                *d = q.accumulate ? *d + av[e] : av[e];
            }
            continue;
        }
        if (q.colmap) {
            const int n = 4 * c4;
            const int c0 = q.colmap[n], c1 = q.colmap[n + 1], c2 = q.colmap[n + 2], c3 = q.colmap[n + 3];
            if (q.accumulate) { cp[c0] += acc.x; cp[c1] += acc.y; cp[c2] += acc.z; cp[c3] += acc.w; }
            else { cp[c0] = acc.x; cp[c1] = acc.y; cp[c2] = acc.z; cp[c3] = acc.w; }
        } else {
            float* d = cp + 4 * c4;
            if (q.accumulate) {
                if ((reinterpret_cast<uintptr_t>(d) & 15) == 0) {
                    float4 o = *reinterpret_cast<float4*>(d);
                    o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
                    *reinterpret_cast<float4*>(d) = o;
                } else { d[0] += acc.x; d[1] += acc.y; d[2] += acc.z; d[3] += acc.w; }
            } else {
                if ((reinterpret_cast<uintptr_t>(d) & 15) == 0) *reinterpret_cast<float4*>(d) = acc;
                else { d[0] = acc.x; d[1] = acc.y; d[2] = acc.z; d[3] = acc.w; }
            }
        }
    }
}

// C[m][n] (+)= T[n][m]  (T [N][M] row-major, pitch M): 32 x 32 tiles through LDS, coalesced on both sides
__global__ __launch_bounds__(256) void transpose_add_kernel(float* __restrict__ C, int ldc, const float* __restrict__ T, int M, int N, int accumulate) {
    __shared__ float tile[32][33];
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, m = m0 + tx;
        tile[r][tx] = (n < N && m < M) ? T[(size_t)n * M + m] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int m = m0 + r, n = n0 + tx;
        if (m < M && n < N) {
            float* d = C + (size_t)m * ldc + n;
            *d = accumulate ? *d + tile[tx][r] : tile[tx][r];
        }
    }
}
int transpose_add(hipStream_t s, float* C, int ldc, const float* T, int M, int N, int accumulate) {
    hipLaunchKernelGGL(transpose_add_kernel, dim3((M + 31) / 32, (N + 31) / 32), dim3(256), 0, s, C, ldc, T, M, N, accumulate);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

int slab_reduce(hipStream_t s, float* C, const float* slab, const SlabMap& q) {
    if (!C || !slab || q.M <= 0 || q.N <= 0 || (q.N & 3) || q.Nvalid <= 0 || q.Nvalid > q.N || q.nsl < 1 || q.nsl > q.nsl_alloc || q.batch < 1) return ASR_EINVAL;
    const size_t total = (size_t)q.M * (q.N >> 2) * q.batch;
    const unsigned grid = (unsigned)std::min<size_t>(4096, (total + 255) / 256);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid), dim3(256), 0, s, C, slab, q);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ELAUNCH;
}

// ---- embedding gradient without atomics: one workgroup per vocabulary row; wave w walks quarter w of the token list (four
// 64-token ballots in flight per round) and adds the gradient rows of the row's occurrences in ascending token order, then the
// four partial sums are added in wave order -- a FIXED association: ((q0 + q1) + q2) + q3, each q in token order.  No pre-zero,
// no atomics.  The dependent chain is rows / 1024 rounds + the row's own occurrences / 4 -- short enough to survive next to a
// persistent kernel that owns the CUs (the first version, one wave per row over the whole list, took 2 ms there).
// vocab <= 0: the table height is unknown -- workgroup r serves row idx[r] if r is that row's first occurrence.
template <bool BY_VOCAB>
__global__ __launch_bounds__(256) void scatter_rows_ordered_kernel(float* __restrict__ tg, const int* __restrict__ idx,
                                                                   const float* __restrict__ g, int rows, int width, int ldg, int vocab) {
    __shared__ float part[4][1024];
    __shared__ int anyw[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int v;
    if (BY_VOCAB) v = blockIdx.x;
    else {
        const int rme = blockIdx.x;
        v = idx[rme];
        bool dup = false;
        for (int r = threadIdx.x; r < rme; r += 256) dup |= idx[r] == v;
        if (__syncthreads_or(dup)) return;
    }
    const int Q = ((rows + 3) / 4 + 63) & ~63;          // tokens per wave
    const int q0 = wave * Q, q1 = min(rows, q0 + Q);
    float acc[16];                                      // columns lane, lane + 64, ... (width <= 1024)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    bool any = false;
    for (int r0 = q0; r0 < q1; r0 += 256) {
        int t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int r = r0 + 64 * u + lane; t[u] = r < q1 ? idx[r] : -1; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            unsigned long long bal = __ballot(t[u] == v);
            while (bal) {                               // this chunk's occurrences, ascending
                const int h = __builtin_ctzll(bal);
                bal &= bal - 1;
                any = true;
                const float* gr = g + (size_t)(r0 + 64 * u + h) * ldg;
#pragma unroll
                for (int j = 0; j < 16; ++j) { const int c = lane + 64 * j; if (c < width) acc[j] += gr[c]; }
            }
        }
    }
    if (lane == 0) anyw[wave] = any;
#pragma unroll
    for (int j = 0; j < 16; ++j) { const int c = lane + 64 * j; if (c < width) part[wave][c] = acc[j]; }
    __syncthreads();
    if (!(anyw[0] | anyw[1] | anyw[2] | anyw[3])) return;
    for (int c = threadIdx.x; c < width; c += 256)
        tg[(size_t)v * width + c] += ((part[0][c] + part[1][c]) + part[2][c]) + part[3][c];
}

}  // namespace asr

// 1: split-K partial tiles through slabs + fixed-order reduce, ordered embedding scatter, two-stage column sums -- every gradient
// bit-reproducible run to run; 0 (default): float atomics.  Environment: ASR_WGRAD_SLABS.
extern "C" int asr_set_wgrad_mode(int slabs) { asr::g_wgrad_slabs = slabs; return ASR_OK; }       // (2: EXPERIMENT -- the GEMMs only)
extern "C" int asr_get_wgrad_mode(void) { return asr::wgrad_slabs(); }

// tg[idx[r]] += g[r], the occurrences of a row added in ascending r (deterministic form of asr_scatter_add_rows); width <= 1024.
// vocab = the number of rows of tg (0 if unknown: slower form)
extern "C" int asr_scatter_add_rows_ordered(void* stream, float* tg, int vocab, const int* idx, const float* g, int rows, int width, int ldg) {
    if (!tg || !idx || !g || rows <= 0 || width <= 0 || width > 1024 || ldg < width) return ASR_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vocab > 0) hipLaunchKernelGGL(asr::scatter_rows_ordered_kernel<true>, dim3(vocab), dim3(256), 0, s, tg, idx, g, rows, width, ldg, vocab);
    else hipLaunchKernelGGL(asr::scatter_rows_ordered_kernel<false>, dim3(rows), dim3(256), 0, s, tg, idx, g, rows, width, ldg, 0);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
