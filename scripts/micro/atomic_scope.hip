// Split-K epilogue of the weight-gradient GEMMs: float atomicAdd of 128x128 tiles into a small C from many workgroups.
// Question: agent-scope atomics (what atomicAdd() emits: performed past the XCD's L2) against workgroup-scope atomics (performed
// IN the XCD's L2), the latter only legal when every workgroup that adds into a tile runs on the same XCD.
//   build: hipcc --offload-arch=gfx950 -O3 -o atomic_scope atomic_scope.hip      run: ./atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SCOPE, bool PIN>
__global__ __launch_bounds__(256) void add_tiles(float* C, int ldc, int ntn, int ntiles, int splits, float val) {
    const int id = blockIdx.x;
    int tile;
    if (PIN) {               // all K slices of a tile on one XCD (workgroup id % 8 = XCD)
        const int x = id & 7, j = id >> 3;
        tile = x + 8 * (j / splits);
    } else {                 // slice s of every tile on XCD s % 8 (gemm.hip's xcd_split): a tile's slices sit on different XCDs
        tile = (id >> 3) % ntiles;
    }
    if (tile >= ntiles) return;
    const int m0 = (tile / ntn) * 128, n0 = (tile % ntn) * 128;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // C/D map of 2x2 32x32 MFMA tiles per wave (64x64 per wave, 4 waves): col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + (w >> 1) * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int n = n0 + (w & 1) * 64 + j * 32 + (lane & 31);
                float* cp = C + (size_t)m * ldc + n;
                if (SCOPE == 0) atomicAdd(cp, val);
                else __hip_atomic_fetch_add(cp, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
}

template <int SCOPE, bool PIN>
static void run(const char* name, int M, int N, int splits) {
    float* C; hipMalloc(&C, (size_t)M * N * 4); hipMemset(C, 0, (size_t)M * N * 4);
    const int ntn = N / 128, ntiles = (M / 128) * ntn;
    const int grid = PIN ? 8 * ((ntiles + 7) / 8) * splits : ntiles * splits;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((add_tiles<SCOPE, PIN>), dim3(grid), dim3(256), 0, 0, C, N, ntn, ntiles, splits, 0.f);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((add_tiles<SCOPE, PIN>), dim3(grid), dim3(256), 0, 0, C, N, ntn, ntiles, splits, 1.f);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> h((size_t)M * N);
    hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (float v : h) bad += v != (float)(reps * splits);
    const double bytes = (double)ntiles * splits * 128 * 128 * 4;
    printf("%-34s C %4dx%4d splits %2d  grid %4d : %7.1f us/launch  %6.2f TB/s of adds  wrong elements %zu\n", name, M, N, splits, grid,
           ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12, bad);
    hipFree(C);
}

int main() {
    for (int splits : {8, 16}) {
        run<0, false>("agent scope, slices across XCDs", 1024, 1024, splits);
        run<0, true>("agent scope, tile pinned to XCD", 1024, 1024, splits);
        run<1, true>("workgroup scope, tile pinned", 1024, 1024, splits);
    }
    run<0, false>("agent scope, slices across XCDs", 256, 1024, 32);
    run<0, true>("agent scope, tile pinned to XCD", 256, 1024, 32);
    run<1, true>("workgroup scope, tile pinned", 256, 1024, 32);
    return 0;
}
