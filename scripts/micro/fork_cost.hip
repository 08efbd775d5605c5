// What the library's fork (event recorded on the main stream, a low-priority side stream waits on it and runs a small kernel)
// costs the MAIN stream's timeline, and whether hipExtLaunchKernelGGL's stop event (the dispatch's own completion signal
// instead of a marker packet behind it) is cheaper.  GPU-bound: ~20 us kernels, the host runs ahead.
// build: hipcc --offload-arch=gfx950 -O3 fork_cost.hip -o fork_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
__global__ void busy(float* x, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] += 1.0f;
}
__global__ void small(float* y) { y[threadIdx.x] += 1.0f; }
int main() {
    const int n = 1 << 24, N = 300;
    float *x, *y;
    hipMalloc(&x, n * 4); hipMalloc(&y, 4096);
    hipMemset(x, 0, n * 4); hipMemset(y, 0, 4096);
    hipStream_t ms, ss;
    hipStreamCreateWithFlags(&ms, hipStreamNonBlocking);
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithPriority(&ss, hipStreamNonBlocking, lo);
    std::vector<hipEvent_t> ev(N);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const char* names[] = {"none", "none", "record", "record+wait (fork)", "ext stop event", "ext stop event + wait (fork)", "none"};
    for (int mode = 0; mode < 7; ++mode) {
        hipDeviceSynchronize();
        hipEventRecord(a, ms);
        for (int i = 0; i < N; ++i) {
            const int m = mode == 6 ? 0 : (mode <= 1 ? 0 : mode - 1);
            if (m == 3 || m == 4) hipExtLaunchKernelGGL(busy, dim3(n / 256), dim3(256), 0, ms, nullptr, ev[i], 0, x, n);
            else hipLaunchKernelGGL(busy, dim3(n / 256), dim3(256), 0, ms, x, n);
            if (m == 1 || m == 2) hipEventRecord(ev[i], ms);
            if (m == 2 || m == 4) { hipStreamWaitEvent(ss, ev[i], 0); hipLaunchKernelGGL(small, dim3(1), dim3(64), 0, ss, y); }
        }
        hipEventRecord(b, ms);
        hipDeviceSynchronize();
        float t = 0; hipEventElapsedTime(&t, a, b);
        printf("%-32s %.2f us per kernel on the main stream's timeline\n", names[mode], t * 1e3 / N);
    }
    return 0;
}
