// Small bandwidth-bound helpers: bias-gradient column sums, embedding gather / scatter-add,
// and the fused global-norm clip + Adam update over the flat parameter buffer.
#include "common.h"
#include <algorithm>

namespace asr {

// out[n] += sum_m x[m][n].  Block = 64 columns x 4 row-strips of its blockIdx.y slab; slabs
// meet through one float atomic per column (out is pre-zeroed by the host when not accumulating).
// (columns from `split` on go to out2 + (c - split): the two directions' bias gradients of a BiLSTM layer in one launch)
__global__ __launch_bounds__(256) void colsum_kernel(const float* x, int ldx, int M, int N, float* out, int rows_per_slab,
                                                     int split = 0x7fffffff, float* out2 = nullptr) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), strip = threadIdx.x >> 6;
    const int s0 = blockIdx.y * rows_per_slab, s1 = min(M, s0 + rows_per_slab);
    float s = 0.f;
    if (c < N) {
        const int rows = (s1 - s0 + 3) / 4;
        const int m0 = s0 + strip * rows, m1 = min(s1, m0 + rows);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int m = m0;
        for (; m + 3 < m1; m += 4) {
            a0 += x[(size_t)m * ldx + c]; a1 += x[(size_t)(m + 1) * ldx + c];
            a2 += x[(size_t)(m + 2) * ldx + c]; a3 += x[(size_t)(m + 3) * ldx + c];
        }
        for (; m < m1; ++m) a0 += x[(size_t)m * ldx + c];
        s = (a0 + a1) + (a2 + a3);
    }
    part[strip][threadIdx.x & 63] = s;
    __syncthreads();
    if (strip == 0 && c < N) {
        const float t = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        float* o = c >= split ? out2 + (c - split) : out + c;
        if (gridDim.y == 1) *o += t; else atomicAdd(o, t);
    }
}

// deterministic form (ASR_WGRAD_SLABS, default): the row slabs' partial sums go to a workspace [slabs][N] instead of meeting in
// `out` through atomics, and colsum_finish_kernel adds them in ascending slab order
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* x, int ldx, int M, int N, float* ws, int rows_per_slab) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), strip = threadIdx.x >> 6;
    const int s0 = blockIdx.y * rows_per_slab, s1 = min(M, s0 + rows_per_slab);
    float s = 0.f;
    if (c < N) {
        const int rows = (s1 - s0 + 3) / 4;
        const int m0 = s0 + strip * rows, m1 = min(s1, m0 + rows);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int m = m0;
        for (; m + 3 < m1; m += 4) {
            a0 += x[(size_t)m * ldx + c]; a1 += x[(size_t)(m + 1) * ldx + c];
            a2 += x[(size_t)(m + 2) * ldx + c]; a3 += x[(size_t)(m + 3) * ldx + c];
        }
        for (; m < m1; ++m) a0 += x[(size_t)m * ldx + c];
        s = (a0 + a1) + (a2 + a3);
    }
    part[strip][threadIdx.x & 63] = s;
    __syncthreads();
    if (strip == 0 && c < N)
        ws[(size_t)blockIdx.y * N + c] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* ws, int slabs, int N, float* out, int split, float* out2) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    float t = 0.f;
    int s = 0;
    for (; s + 8 <= slabs; s += 8) {            // eight independent loads, then added in ascending order
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ws[(size_t)(s + j) * N + c];
#pragma unroll
        for (int j = 0; j < 8; ++j) t += v[j];
    }
    for (; s < slabs; ++s) t += ws[(size_t)s * N + c];
    float* o = c >= split ? out2 + (c - split) : out + c;
    *o += t;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* table, const int* idx, float* out, int rows, int width) {
    const int w4 = width >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)rows * w4; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / w4), c = (int)(i % w4);
        reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(table + (size_t)idx[r] * width)[c];
    }
}

// Up to 8 row-wise concatenations dst_i[r] = [a_i[r] (wa floats) | b_i[r] (wb floats)] in ONE launch (blockIdx.y = i): the
// encoder's per-layer [in, 8H] kernel and [8H] bias concatenations of a whole step (8 small launches otherwise).
struct ConcatJob { const float* a; const float* b; float* dst; int rows, wa, wb, lda, ldb; };
struct ConcatJobs { ConcatJob j[8]; };
__global__ __launch_bounds__(256) void concat2_multi_kernel(ConcatJobs jobs) {
    const ConcatJob j = jobs.j[blockIdx.y];
    const int w = j.wa + j.wb;
    const size_t n = (size_t)j.rows * w;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / w), c = (int)(i % w);
        j.dst[i] = c < j.wa ? j.a[(size_t)r * j.lda + c] : j.b[(size_t)r * j.ldb + (c - j.wa)];
    }
}

__global__ __launch_bounds__(256) void scatter_add_rows_kernel(float* tg, const int* idx, const float* g, int rows, int width, int ldg) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)rows * width; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / width), c = (int)(i % width);
        atomicAdd(tg + (size_t)idx[r] * width + c, g[(size_t)r * ldg + c]);
    }
}

// stage 1: up to 1024 block partials of sum(x^2); stage 2: one block sums them in order.
__global__ __launch_bounds__(256) void sumsq_stage1(const float* x, size_t n, float* ws) {
    __shared__ float red[256];
    float s = 0.f;
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0) for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += x[i] * x[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) ws[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void sumsq_stage2(const float* ws, int nb, float* out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += ws[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = red[0];
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* p, float* m, float* v, const float* g, size_t n,
                                                        const float* sumsq, float gscale, float clip, float lr_t,
                                                        float b1, float b2, float eps) {
    // tf.clip_by_global_norm: scale = clip * min(1/norm, 1/clip) = clip / max(norm, clip)
    const float norm = sqrtf(sumsq[0]) * gscale;
    const float sc = gscale * (norm > 0.f ? clip / fmaxf(norm, clip) : 1.f);
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i], pp = reinterpret_cast<float4*>(p)[i];
#define ASR_ADAM1(c) { const float gx = gg.c * sc; mm.c = b1 * mm.c + (1.f - b1) * gx; vv.c = b2 * vv.c + (1.f - b2) * gx * gx; \
                       pp.c -= lr_t * mm.c / (sqrtf(vv.c) + eps); }
        ASR_ADAM1(x) ASR_ADAM1(y) ASR_ADAM1(z) ASR_ADAM1(w)
#undef ASR_ADAM1
        reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv; reinterpret_cast<float4*>(p)[i] = pp;
    }
    if (blockIdx.x == 0)
        for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
            const float gx = g[i] * sc;
            m[i] = b1 * m[i] + (1.f - b1) * gx; v[i] = b2 * v[i] + (1.f - b2) * gx * gx;
            p[i] -= lr_t * m[i] / (sqrtf(v[i]) + eps);
        }
}

// num_utils.py:6-14 of the reference: elementwise logistic; softmax over a 1-D vector (axis 0), max-shifted.
__global__ __launch_bounds__(256) void sigmoid_kernel(const float* x, float* y, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = 1.0f / (1.0f + expf(-x[i]));
}
__global__ __launch_bounds__(256) void softmax1d_kernel(const float* x, float* y, int n) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, x[i]);
    m = wave_allreduce_max(m);
    if (lane == 0) red[w] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += expf(x[i] - m);
    s = wave_allreduce_sum(s);
    if (lane == 0) red[w] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    for (int i = threadIdx.x; i < n; i += 256) y[i] = expf(x[i] - m) / s;
}

// Pyramid time reduction (encoder.py:94-119) as a standalone copy for callers whose layer output is NOT already
// laid out [B, T_pad, F] with T_pad divisible by skip (inside Encoder it is a view: the recurrent kernel writes the
// zero pad frame itself).  y[b, t', s*F + f] = x[b, t'*skip + s, f], zero where t'*skip + s >= T.  Row-contiguous in
// both directions: each thread moves one float4 (or one float when F % 4 != 0), consecutive lanes consecutive addresses.
template <int VEC>
__global__ __launch_bounds__(256) void pyramid_kernel(const float* x, float* y, int B, int T, int F, int skip, int Tp, int bwd) {
    const size_t rowlen = (size_t)F / VEC;                         // elements (of VEC floats) per input frame
    const size_t total = (size_t)B * Tp * skip * rowlen;           // over the PADDED frames
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t e = i % rowlen, frame = i / rowlen;               // frame index in [B][Tp*skip]
    const size_t b = frame / ((size_t)Tp * skip), t = frame % ((size_t)Tp * skip);
    const size_t yi = (frame * rowlen + e) * VEC;                   // y is [B][Tp][skip*F] == [B][Tp*skip][F] flat
    const size_t xi = ((b * T + t) * rowlen + e) * VEC;
    if (VEC == 4) {
        if (!bwd) *reinterpret_cast<float4*>(y + yi) = t < (size_t)T ? *reinterpret_cast<const float4*>(x + xi) : make_float4(0.f, 0.f, 0.f, 0.f);
        else if (t < (size_t)T) *reinterpret_cast<float4*>(y + xi) = *reinterpret_cast<const float4*>(x + yi);
    } else {
        if (!bwd) y[yi] = t < (size_t)T ? x[xi] : 0.f;
        else if (t < (size_t)T) y[xi] = x[yi];
    }
}
__global__ __launch_bounds__(256) void ceil_div_len_kernel(const int* len_in, int* len_out, int B, int skip) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B) len_out[b] = (len_in[b] + skip - 1) / skip;         // ceil (encoder.py:117-118)
}

}  // namespace asr

extern "C" int asr_colsum_f32(void* stream, const float* x, int ldx, int M, int N, float* out, int accumulate) {
    if (!x || !out || M < 0 || N <= 0 || ldx < N) return ASR_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!accumulate && hipMemsetAsync(out, 0, sizeof(float) * N, s) != hipSuccess) return ASR_ELAUNCH;
    const int nx = (N + 63) / 64;
    // enough workgroups for memory-level parallelism: a thread walks its rows with 4 loads in flight, so short strips
    // (>= 8 rows per thread) and up to ~2048 workgroups; slabs meet through one atomic per column
    int slabs = std::max(1, std::min(M / 32, (2048 + nx - 1) / nx));
    const int rows_per_slab = (M + slabs - 1) / slabs;
    slabs = (M + rows_per_slab - 1) / rows_per_slab;
    if (M == 0) return ASR_OK;
    if (slabs > 1 && asr::wgrad_slabs() == 1) {
        // (at most 64 row slabs: colsum_finish_kernel walks them in order, eight loads in flight)
        const int rps = std::max(rows_per_slab, (M + 63) / 64);
        slabs = (M + rps - 1) / rps;
        float* ws = asr::slab_arena(s, (size_t)slabs * N * sizeof(float));
        if (ws) {
            hipLaunchKernelGGL(asr::colsum_part_kernel, dim3(nx, slabs), dim3(256), 0, s, x, ldx, M, N, ws, rps);
            hipLaunchKernelGGL(asr::colsum_finish_kernel, dim3((N + 63) / 64), dim3(64), 0, s, ws, slabs, N, out, 0x7fffffff, (float*)nullptr);
            ASR_CHECK_LAUNCH();
            return ASR_OK;
        }
    }
    hipLaunchKernelGGL(asr::colsum_kernel, dim3(nx, slabs), dim3(256), 0, s, x, ldx, M, N, out, rows_per_slab);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
// out0[n] += sum_m x[m][n], out1[n] += sum_m x[m][N + n] for n < N (x rows of 2N columns): internal, used by asr_lstm_layer_bwd
int asr_colsum_pair_f32(hipStream_t s, const float* x, int ldx, int M, int N, float* out0, float* out1) {
    if (!x || !out0 || !out1 || M < 0 || N <= 0 || ldx < 2 * N) return ASR_EINVAL;
    if (M == 0) return ASR_OK;
    const int nx = (2 * N + 63) / 64;
    int slabs = std::max(1, std::min(M / 32, (2048 + nx - 1) / nx));
    const int rows_per_slab = (M + slabs - 1) / slabs;
    slabs = (M + rows_per_slab - 1) / rows_per_slab;
    if (slabs > 1 && asr::wgrad_slabs() == 1) {
        const int rps = std::max(rows_per_slab, (M + 63) / 64);
        slabs = (M + rps - 1) / rps;
        float* ws = asr::slab_arena(s, (size_t)slabs * 2 * N * sizeof(float));
        if (ws) {
            hipLaunchKernelGGL(asr::colsum_part_kernel, dim3(nx, slabs), dim3(256), 0, s, x, ldx, M, 2 * N, ws, rps);
            hipLaunchKernelGGL(asr::colsum_finish_kernel, dim3((2 * N + 63) / 64), dim3(64), 0, s, ws, slabs, 2 * N, out0, N, out1);
            ASR_CHECK_LAUNCH();
            return ASR_OK;
        }
    }
    hipLaunchKernelGGL(asr::colsum_kernel, dim3(nx, slabs), dim3(256), 0, s, x, ldx, M, 2 * N, out0, rows_per_slab, N, out1);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
extern "C" int asr_gather_rows(void* stream, const float* table, const int* idx, float* out, int rows, int width) {
    if (!table || !idx || !out || rows <= 0 || width <= 0 || (width & 3)) return ASR_EINVAL;
    const int grid = (int)std::min<size_t>(2048, ((size_t)rows * (width >> 2) + 255) / 256);
    hipLaunchKernelGGL(asr::gather_rows_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), table, idx, out, rows, width);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
// n <= 8 concatenations in one launch; job i: dst[i] [rows[i], wa[i]+wb[i]] = [a[i] (row stride lda[i]) | b[i] (row stride ldb[i])]
extern "C" int asr_concat2_multi(void* stream, int n, const float* const* a, const float* const* b, float* const* dst,
                                 const int* rows, const int* wa, const int* wb, const int* lda, const int* ldb) {
    if (n <= 0 || n > 8 || !a || !b || !dst || !rows || !wa || !wb || !lda || !ldb) return ASR_EINVAL;
    asr::ConcatJobs jobs;
    size_t most = 0;
    for (int i = 0; i < n; ++i) {
        if (!a[i] || !b[i] || !dst[i] || rows[i] <= 0 || wa[i] <= 0 || wb[i] <= 0 || lda[i] < wa[i] || ldb[i] < wb[i]) return ASR_EINVAL;
        jobs.j[i] = asr::ConcatJob{a[i], b[i], dst[i], rows[i], wa[i], wb[i], lda[i], ldb[i]};
        most = std::max(most, (size_t)rows[i] * (wa[i] + wb[i]));
    }
    const int gx = (int)std::min<size_t>(512, (most + 255) / 256);
    hipLaunchKernelGGL(asr::concat2_multi_kernel, dim3(gx, n), dim3(256), 0, static_cast<hipStream_t>(stream), jobs);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
extern "C" int asr_scatter_add_rows_ld(void* stream, float* tg, const int* idx, const float* g, int rows, int width, int ldg);
extern "C" int asr_scatter_add_rows(void* stream, float* tg, const int* idx, const float* g, int rows, int width) {
    return asr_scatter_add_rows_ld(stream, tg, idx, g, rows, width, width);
}
extern "C" int asr_scatter_add_rows_ordered(void* stream, float* tg, int vocab, const int* idx, const float* g, int rows, int width, int ldg);
extern "C" int asr_scatter_add_rows_ld(void* stream, float* tg, const int* idx, const float* g, int rows, int width, int ldg) {
    if (!tg || !idx || !g || rows <= 0 || width <= 0 || ldg < width) return ASR_EINVAL;
    if (asr::wgrad_slabs() == 1 && width <= 1024) return asr_scatter_add_rows_ordered(stream, tg, 0, idx, g, rows, width, ldg);   // no atomics: fixed order
    const int grid = (int)std::min<size_t>(2048, ((size_t)rows * width + 255) / 256);
    hipLaunchKernelGGL(asr::scatter_add_rows_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), tg, idx, g, rows, width, ldg);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
extern "C" int asr_sumsq_f32(void* stream, const float* x, size_t n, float* ws, float* out) {
    if (!x || !ws || !out || n == 0 || (reinterpret_cast<uintptr_t>(x) & 15)) return ASR_EINVAL;
    const int nb = (int)std::min<size_t>(1024, (n / 4 + 255) / 256 + 1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(asr::sumsq_stage1, dim3(nb), dim3(256), 0, s, x, n, ws);
    hipLaunchKernelGGL(asr::sumsq_stage2, dim3(1), dim3(256), 0, s, ws, nb, out);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
extern "C" int asr_clip_adam_f32(void* stream, float* p, float* m, float* v, const float* g, size_t n,
                                 const float* sumsq, float grad_scale, float clip_norm, float lr_t,
                                 float beta1, float beta2, float eps) {
    if (!p || !m || !v || !g || !sumsq || n == 0) return ASR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
         reinterpret_cast<uintptr_t>(g)) & 15) return ASR_EINVAL;
    const int nb = (int)std::min<size_t>(2048, (n / 4 + 255) / 256 + 1);
    asr::prof_begin(ASR_PROF_OPTIM, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(asr::clip_adam_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), p, m, v, g, n,
                       sumsq, grad_scale, clip_norm, lr_t, beta1, beta2, eps);
    asr::prof_end(ASR_PROF_OPTIM, static_cast<hipStream_t>(stream));
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}


extern "C" int asr_sigmoid_f32(void* stream, const float* x, float* y, size_t n) {
    if (!x || !y) return ASR_EINVAL;
    if (n == 0) return ASR_OK;
    hipLaunchKernelGGL(asr::sigmoid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

extern "C" int asr_softmax_f32(void* stream, const float* x, float* y, int n) {
    if (!x || !y || n <= 0) return ASR_EINVAL;
    hipLaunchKernelGGL(asr::softmax1d_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}


// y [B, Tp, skip*F] with Tp = ceil(T / skip); len_out[b] = ceil(len_in[b] / skip) (either may be NULL).
extern "C" int asr_pyramid_reduce_fwd(void* stream, const float* x, const int* len_in, float* y, int* len_out,
                                      int B, int T, int F, int skip) {
    if (!x || !y || B <= 0 || T <= 0 || F <= 0 || skip <= 0) return ASR_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Tp = (T + skip - 1) / skip;
    const bool v4 = F % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    const size_t total = (size_t)B * Tp * skip * (v4 ? F / 4 : F);
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (v4) hipLaunchKernelGGL(asr::pyramid_kernel<4>, dim3(grid), dim3(256), 0, s, x, y, B, T, F, skip, Tp, 0);
    else hipLaunchKernelGGL(asr::pyramid_kernel<1>, dim3(grid), dim3(256), 0, s, x, y, B, T, F, skip, Tp, 0);
    if (len_in && len_out) hipLaunchKernelGGL(asr::ceil_div_len_kernel, dim3((B + 255) / 256), dim3(256), 0, s, len_in, len_out, B, skip);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// dx [B, T, F] = the un-padded frames of dy [B, Tp, skip*F] (the gradient of the zero pad is dropped).
extern "C" int asr_pyramid_reduce_bwd(void* stream, const float* dy, float* dx, int B, int T, int F, int skip) {
    if (!dy || !dx || B <= 0 || T <= 0 || F <= 0 || skip <= 0) return ASR_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Tp = (T + skip - 1) / skip;
    const bool v4 = F % 4 == 0 && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
    const size_t total = (size_t)B * Tp * skip * (v4 ? F / 4 : F);
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (v4) hipLaunchKernelGGL(asr::pyramid_kernel<4>, dim3(grid), dim3(256), 0, s, dy, dx, B, T, F, skip, Tp, 1);
    else hipLaunchKernelGGL(asr::pyramid_kernel<1>, dim3(grid), dim3(256), 0, s, dy, dx, B, T, F, skip, Tp, 1);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
