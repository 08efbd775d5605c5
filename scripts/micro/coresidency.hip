// Which kernels get a CU's free registers while a persistent 512-thread workgroup sits on it?  (what can run beside the
// groups-of-four recurrent kernels: 8 waves per CU at 200-248 VGPRs)
// Kernel A: 256 workgroups x 512 threads spinning ~1 ms with NA registers live; kernel B (another stream, launched while A
// runs): 1024 workgroups x 256 threads, NB registers live, ~20 us each.  B done long before A => B ran beside A.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/coresidency.hip -o scripts/micro/coresidency
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NV, int NT>
__global__ __launch_bounds__(NT) void spin(float* out, long long ticks, unsigned long long* when) {
    float x[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) x[i] = (float)(threadIdx.x + i);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < ticks) {
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(x[i]));
        __builtin_amdgcn_s_sleep(8);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += x[i];
    if (s == 1.2345f) out[0] = s;
    if (when && threadIdx.x == 0) atomicMax(when, __builtin_amdgcn_s_memtime());
}

template <int NA, int NB>
void run(hipStream_t s1, hipStream_t s2, float* out, unsigned long long* when) {
    hipEvent_t a0, a1, b0, b1;
    (void)hipEventCreate(&a0); (void)hipEventCreate(&a1); (void)hipEventCreate(&b0); (void)hipEventCreate(&b1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a0, s1);
    spin<NA, 512><<<256, 512, 0, s1>>>(out, 2300000, nullptr);       // ~1 ms
    (void)hipEventRecord(a1, s1);
    (void)hipStreamWaitEvent(s2, a0, 0);
    (void)hipEventRecord(b0, s2);
    spin<8, 256><<<1, 256, 0, s2>>>(out, 230000, nullptr);           // 100 us delay: A is resident everywhere by then
    spin<NB, 256><<<1024, 256, 0, s2>>>(out, 46000, nullptr);        // 1024 x ~20 us
    (void)hipEventRecord(b1, s2);
    (void)hipDeviceSynchronize();
    float ta, tb, tab;
    (void)hipEventElapsedTime(&ta, a0, a1); (void)hipEventElapsedTime(&tb, b0, b1); (void)hipEventElapsedTime(&tab, a0, b1);
    hipFuncAttributes fa, fb;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(spin<NA, 512>));
    (void)hipFuncGetAttributes(&fb, reinterpret_cast<const void*>(spin<NB, 256>));
    printf("A: %3d VGPRs x 8 waves per CU, %.2f ms | B: %3d VGPRs, done %.2f ms after A's start  -> %s\n", fa.numRegs, ta, fb.numRegs, tab,
           tab < ta * 0.8 ? "B ran BESIDE A" : "B waited for A");
}

int main() {
    float* out; unsigned long long* when;
    (void)hipMalloc(&out, 4); (void)hipMalloc(&when, 8);
    hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
    spin<8, 256><<<1, 256>>>(out, 1000, nullptr); (void)hipDeviceSynchronize();
    run<100, 24>(s1, s2, out, when); run<100, 48>(s1, s2, out, when);
    run<180, 24>(s1, s2, out, when); run<180, 48>(s1, s2, out, when); run<180, 64>(s1, s2, out, when);
    run<196, 24>(s1, s2, out, when); run<196, 48>(s1, s2, out, when); run<196, 64>(s1, s2, out, when);
    run<208, 24>(s1, s2, out, when); run<208, 32>(s1, s2, out, when); run<208, 48>(s1, s2, out, when); run<208, 64>(s1, s2, out, when);
    run<240, 8>(s1, s2, out, when); run<240, 24>(s1, s2, out, when);
    return 0;
}
