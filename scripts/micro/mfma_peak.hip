// What does the bf16 matrix pipe sustain when NOTHING else is in the way?  (the practical ceiling for gemm_planes_kernel)
// Every wave issues v_mfma_f32_32x32x16_bf16 back to back on four independent accumulators (register operands only, no
// memory, no LDS), for ~1 ms and ~20 ms (clocks under sustained load), with 1, 2 and 4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_peak.hip -o scripts/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(e + 1); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = c1 - c0; }
}

int main() {
    float* out; unsigned long long* clk;
    (void)hipMalloc(&out, 4); (void)hipMallocManaged(&clk, 16);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wgs_per_cu : {1, 2, 4})
        for (int iters : {20000, 400000}) {
            const int grid = cus * wgs_per_cu;
            k<<<grid, 256>>>(out, 1000, clk); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            k<<<grid, 256>>>(out, iters / wgs_per_cu, clk);
            (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)grid * 4 * (iters / wgs_per_cu) * 16.0 * 32 * 32 * 16 * 2;
            printf("%d CUs, %d waves per SIMD, %.2f ms: %.0f TFLOP/s bf16 (%.2f of 2500); s_memtime ticks %llu, shader-clock counter %llu -> %.0f MHz if the tick is 10 ns\n",
                   cus, wgs_per_cu, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500.0, clk[0], clk[1], (double)clk[1] / ((double)clk[0] * 0.01));
        }
    return 0;
}
