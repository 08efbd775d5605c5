// Does a line stay in L2 across a kernel boundary?  (diagnostic)
// One lane chases a pointer chain through 256 lines (128 B apart, 32 KB: more than fits... no: it fits the 32 KB L1 only
// partly, so every load is issued with sc1 = an L1 miss by construction) twice per launch: pass 1 = first touch in this
// kernel, pass 2 = the same lines again (an L2 hit for sure).  Launched three times back to back on one stream: if pass 1 of
// launches 2 and 3 costs what pass 2 costs, L2 contents survive the boundary; if it costs what launch 1's pass 1 costs (HBM
// / Infinity Cache), every small kernel of a dependent chain starts with cold weights.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/l2_persist.hip -o scripts/micro/l2_persist
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void chase(const unsigned* buf, int n, unsigned long long* out) {
    if (threadIdx.x) return;
    unsigned idx = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < n; ++i) {
            unsigned x;
            asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(buf + idx) : "memory");
            idx = x;
        }
        out[pass] = (__builtin_amdgcn_s_memtime() - t0) / n;
    }
    out[2] = idx;
}
__global__ void other(float* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
int main() {
    const int n = 256;
    std::vector<unsigned> h(n * 32, 0);
    for (int i = 0; i < n; ++i) h[i * 32] = ((i * 97 + 31) % n) * 32;       // a permutation walk, one word per 128-byte line
    unsigned* buf; unsigned long long* out; float* big;
    (void)hipMalloc(&buf, h.size() * 4); (void)hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMallocManaged(&out, 64); (void)hipMalloc(&big, 1 << 20); (void)hipMemset(big, 0, 1 << 20);
    for (int mode = 0; mode < 2; ++mode) {
        printf(mode ? "-- with a small unrelated kernel between the launches\n" : "-- back to back\n");
        for (int l = 0; l < 4; ++l) {
            if (mode) { other<<<64, 256>>>(big, 1 << 18); }
            chase<<<1, 64>>>(buf, n, out); (void)hipDeviceSynchronize();
            printf("launch %d: first touch %llu ticks per load, second pass %llu\n", l + 1, out[0], out[1]); fflush(stdout);
        }
    }
    return 0;
}
