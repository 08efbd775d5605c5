// Masked, length-normalised sparse softmax cross-entropy (losses.py:7-35) and the
// feedback-token kernels of the decoder (decoder.py:139-180).
#include "common.h"

namespace asr {

// One wave per logits row: log-sum-exp over V, nll = lse - logit[target].
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* logits, const int* targets, float* nll,
                                                      float* lse_out, int rows, int V) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* lp = logits + (size_t)row * V;
    float m = -INFINITY;
    for (int v = lane; v < V; v += 64) m = fmaxf(m, lp[v]);
    m = wave_allreduce_max(m);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(lp[v] - m);
    s = wave_allreduce_sum(s);
    if (lane == 0) {
        const float lse = m + logf(s);
        const int tg = targets[row];
        nll[row] = lse - ((tg >= 0 && tg < V) ? lp[tg] : 0.f);
        if (lse_out) lse_out[row] = lse;
    }
}

// loss = mean_b( sum_{t < len[b]} nll[t,b] / len[b] )  -- fixed summation order (reproducible)
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* nll, const int* len, float* loss, int T, int B) {
    __shared__ float acc[256];
    // 8 lanes per utterance walk its time steps 8 apart (a single thread per utterance waited out one L2 round trip per
    // step: 29 us for 120 steps), then meet through a fixed butterfly
    float s = 0.f;
    const int sub = threadIdx.x & 7;
    for (int b0 = 0; b0 < B; b0 += 32) {
        const int b = b0 + (threadIdx.x >> 3);
        const int L = b < B ? min(len[b], T) : 0;
        float c = 0.f;
        for (int t = sub; t < L; t += 8) c += nll[(size_t)t * B + b];
        c += __shfl_xor(c, 1); c += __shfl_xor(c, 2); c += __shfl_xor(c, 4);
        if (sub == 0 && b < B) s += c / (float)len[b];
    }
    acc[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) acc[threadIdx.x] += acc[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = acc[0] / (float)B;
}

// dlogits[t,b,v] = g * (softmax - onehot) * [t < len[b]] / (len[b] * B)
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* logits, const int* targets, const float* lse,
                                                     const int* len, const float* gscale, float* dlogits,
                                                     int T, int B, int V) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T * B) return;
    const int t = row / B, b = row % B;
    float* dp = dlogits + (size_t)row * V;
    if (t >= len[b]) { for (int v = lane; v < V; v += 64) dp[v] = 0.f; return; }
    const float sc = gscale[0] / ((float)len[b] * (float)B);
    const float* lp = logits + (size_t)row * V;
    const float l = lse[row];
    const int tg = targets[row];
    for (int v = lane; v < V; v += 64) dp[v] = sc * (expf(lp[v] - l) - (v == tg ? 1.f : 0.f));
}

// Forward rows and backward in ONE pass over the logits (training: the gradient scale is known when the loss is formed): the
// row's log-sum-exp, its nll, and d loss / d logits while the row is still in cache -- one launch instead of two dependent
// ones with the row read from memory again in between.
__global__ __launch_bounds__(256) void ce_rows_bwd_kernel(const float* logits, const int* targets, const int* len,
                                                          const float* gscale, float* nll, float* lse_out, float* dlogits,
                                                          int T, int B, int V) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T * B) return;
    const int t = row / B, b = row % B;
    const float* lp = logits + (size_t)row * V;
    float* dp = dlogits + (size_t)row * V;
    float m = -INFINITY;
    for (int v = lane; v < V; v += 64) m = fmaxf(m, lp[v]);
    m = wave_allreduce_max(m);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(lp[v] - m);
    s = wave_allreduce_sum(s);
    const float lse = m + logf(s);
    const int tg = targets[row];
    if (lane == 0) {
        nll[row] = lse - ((tg >= 0 && tg < V) ? lp[tg] : 0.f);
        if (lse_out) lse_out[row] = lse;
    }
    if (t >= len[b]) { for (int v = lane; v < V; v += 64) dp[v] = 0.f; return; }
    const float sc = gscale[0] / ((float)len[b] * (float)B);
    for (int v = lane; v < V; v += 64) dp[v] = sc * (expf(lp[v] - lse) - (v == tg ? 1.f : 0.f));
}

// argmax with first-max tie-breaking (tf.argmax / np.argmax); optional Gumbel-max sampling
// (tf.multinomial draws from softmax(logits); decoder.py:176-177).
__global__ __launch_bounds__(256) void next_token_kernel(const float* logits, int V, int ldl, int* tok_out,
                                                         int sample, uint32_t seed, uint32_t step) {
    __shared__ float bv[256]; __shared__ int bi[256];
    const int b = blockIdx.x;
    const float* lp = logits + (size_t)b * ldl;
    float best = -INFINITY; int idx = 0x7fffffff;
    for (int v = threadIdx.x; v < V; v += 256) {
        float x = lp[v];
        if (sample) {
            const float u = fmaxf(uniform01(seed, step * 65537u + (uint32_t)b, (uint32_t)v), 1e-12f);
            x -= logf(-logf(u));
        }
        if (x > best) { best = x; idx = v; }
    }
    bv[threadIdx.x] = best; bi[threadIdx.x] = idx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float ov = bv[threadIdx.x + o]; const int oi = bi[threadIdx.x + o];
            if (ov > bv[threadIdx.x] || (ov == bv[threadIdx.x] && oi < bi[threadIdx.x])) {
                bv[threadIdx.x] = ov; bi[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) tok_out[b] = bi[0] == 0x7fffffff ? 0 : bi[0];
}

// raw_rnn emit: rows (t,b) with t >= len[b] are zeros (attn_decoder.py:170 output convention)
__global__ __launch_bounds__(256) void zero_finished_rows_kernel(float* logits, const int* len, int T, int B, int V) {
    const int row = blockIdx.x;
    const int t = row / B, b = row % B;
    if (t < len[b]) return;
    float* p = logits + (size_t)row * V;
    for (int v = threadIdx.x; v < V; v += 256) p[v] = 0.f;
}

}  // namespace asr

extern "C" int asr_zero_finished_rows(void* stream, float* logits, const int* len, int T, int B, int V) {
    if (!logits || !len || T <= 0 || B <= 0 || V <= 0) return ASR_EINVAL;
    hipLaunchKernelGGL(asr::zero_finished_rows_kernel, dim3(T * B), dim3(256), 0, static_cast<hipStream_t>(stream), logits, len, T, B, V);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

extern "C" int asr_masked_ce_fwd(void* stream, const float* logits, const int* targets, const int* len,
                                 float* nll_ws, float* lse_ws, float* loss, int T, int B, int V) {
    if (!logits || !targets || !len || !nll_ws || !loss || T <= 0 || B <= 0 || V <= 0) return ASR_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rows = T * B;
    hipLaunchKernelGGL(asr::ce_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, logits, targets, nll_ws, lse_ws, rows, V);
    hipLaunchKernelGGL(asr::ce_reduce_kernel, dim3(1), dim3(256), 0, s, nll_ws, len, loss, T, B);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// asr_masked_ce_fwd + asr_masked_ce_bwd in one pass over the logits (losses.py:20-35 and its gradient): the same values, bit for bit
extern "C" int asr_masked_ce_fwd_bwd(void* stream, const float* logits, const int* targets, const int* len, const float* grad_scale,
                                     float* nll_ws, float* lse_ws, float* loss, float* dlogits, int T, int B, int V) {
    if (!logits || !targets || !len || !grad_scale || !nll_ws || !loss || !dlogits || T <= 0 || B <= 0 || V <= 0) return ASR_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rows = T * B;
    hipLaunchKernelGGL(asr::ce_rows_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, logits, targets, len, grad_scale, nll_ws, lse_ws,
                       dlogits, T, B, V);
    hipLaunchKernelGGL(asr::ce_reduce_kernel, dim3(1), dim3(256), 0, s, nll_ws, len, loss, T, B);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

extern "C" int asr_masked_ce_bwd(void* stream, const float* logits, const int* targets, const float* lse_ws,
                                 const int* len, const float* grad_scale, float* dlogits, int T, int B, int V) {
    if (!logits || !targets || !lse_ws || !len || !grad_scale || !dlogits) return ASR_EINVAL;
    const int rows = T * B;
    hipLaunchKernelGGL(asr::ce_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                       logits, targets, lse_ws, len, grad_scale, dlogits, T, B, V);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

extern "C" int asr_next_token(void* stream, const float* logits, int B, int V, int ldl, int* tok_out,
                              int sample, unsigned seed, unsigned step) {
    if (!logits || !tok_out || B <= 0 || V <= 0) return ASR_EINVAL;
    hipLaunchKernelGGL(asr::next_token_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream),
                       logits, V, ldl, tok_out, sample, seed, step);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
