// Issue-rate probe (diagnostic): cycles per instruction of v_fma_f32 vs v_pk_fma_f32 vs v_mfma_f32_16x16x4_f32, one and two
// waves per SIMD, independent chains.  hipcc --offload-arch=gfx950 -O3 scripts/micro/fma_rate.hip -o /tmp/fma_rate && /tmp/fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, float a0, int iters) {
    float x[8]; f2 p[8]; f4 m[4];
    for (int i = 0; i < 8; ++i) { x[i] = a0 + i + threadIdx.x; p[i] = f2{a0 + i, a0 - i}; }
    for (int i = 0; i < 4; ++i) m[i] = f4{a0, a0, a0, a0};
    const float b = a0 * 0.5f; const f2 pb = f2{b, b + 1.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(pb));
        } else if (MODE == 3) {        // pk_fma with op_sel broadcast of the low half of src1 (the matvec form)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(pb), "v"(p[(i + 1) & 7]));
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) m[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a0, m[i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y; for (int i = 0; i < 4; ++i) s += m[i].x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 1 << 20); hipMallocManaged(&cyc, 64);
    const int iters = 2000;
    for (int threads : {256, 512}) {
        k<0><<<1, threads>>>(out, cyc, 1.0f, iters); k<1><<<1, threads>>>(out, cyc, 1.0f, iters);
        k<2><<<1, threads>>>(out, cyc, 1.0f, iters); k<3><<<1, threads>>>(out, cyc, 1.0f, iters);
        hipDeviceSynchronize();
        const char* nm[4] = {"v_fma_f32", "v_pk_fma_f32", "v_mfma_f32_16x16x4_f32", "v_pk_fma_f32 op_sel"};
        for (int m = 0; m < 4; ++m)
            printf("%d waves/SIMD  %-24s %.2f memtime ticks per instruction per wave (x%d waves sharing the SIMD)\n", threads / 256, nm[m],
                   (double)cyc[m] / (iters * 32.0), threads / 256);
    }
    return 0;
}
