// One-way latency of the store -> L2 -> polling-load hand-over between two workgroups (diagnostic).
// Workgroups 0 and 8 (same blockIdx % 8 = same XCD under round-robin dispatch) or 0 and 1 (different XCDs) bounce a counter.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/pingpong.hip -o scripts/micro/pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int STORE, int LOAD>
__device__ __forceinline__ void bounce(unsigned* mine, unsigned* theirs, int iters, bool first, unsigned long long* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 1; i <= iters; ++i) {
        if (first) {
            if (STORE == 0) asm volatile("global_store_dword %0, %1, off" :: "v"(mine), "v"(i) : "memory");
            else if (STORE == 1) asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(mine), "v"(i) : "memory");
            else asm volatile("global_atomic_swap %0, %1, off" :: "v"(mine), "v"(i) : "memory");
        }
        unsigned x;
        int guard = 0;
        do {
            if (++guard > 2000000) { if (first) *out = 0; return; }       // never hang the box
            if (LOAD == 0) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(theirs) : "memory");
            else if (LOAD == 1) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(theirs) : "memory");
            else if (LOAD == 2) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(theirs) : "memory");
            else { unsigned sx; asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(sx) : "s"(theirs) : "memory"); x = sx; }
        } while ((int)x < i);
        if (!first) {
            if (STORE == 0) asm volatile("global_store_dword %0, %1, off" :: "v"(mine), "v"(i) : "memory");
            else if (STORE == 1) asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(mine), "v"(i) : "memory");
            else asm volatile("global_atomic_swap %0, %1, off" :: "v"(mine), "v"(i) : "memory");
        }
    }
    if (first) *out = __builtin_amdgcn_s_memtime() - t0;
}
template <int STORE, int LOAD>
__global__ void k(unsigned* buf, int peer, int iters, unsigned long long* out, unsigned* xcc) {
    if (threadIdx.x != 0) return;
    if (blockIdx.x == 0 || blockIdx.x == peer) xcc[blockIdx.x == 0 ? 0 : 1] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xF;
    if (blockIdx.x == 0) bounce<STORE, LOAD>(buf, buf + 64, iters, true, out);
    else if (blockIdx.x == peer) bounce<STORE, LOAD>(buf + 64, buf, iters, false, out);
}
int main() {
    unsigned* buf; unsigned long long* out; unsigned* xcc;
    (void)hipMalloc(&buf, 4096); (void)hipMallocManaged(&out, 64); (void)hipMallocManaged(&xcc, 64);
    const int iters = 2000;
    const char* sn[3] = {"plain store", "sc0 sc1 store", "atomic swap"};
    const char* ln[4] = {"sc1 load", "sc0 sc1 load", "sc0 load", "scalar glc load"};
    for (int peer : {8, 1}) {
#define RUN(S, L) { (void)hipMemset(buf, 0, 4096); k<S, L><<<16, 64>>>(buf, peer, iters, out, xcc); (void)hipDeviceSynchronize(); \
        printf("peer %d (xcc %u vs %u)  %-14s %-13s  %.0f ticks per one-way hand-over\n", peer, xcc[0], xcc[1], sn[S], ln[L], (double)*out / iters / 2); fflush(stdout); }
        RUN(0, 0) RUN(1, 0) RUN(2, 0) RUN(0, 1) RUN(1, 1) RUN(0, 3) RUN(1, 3)
    }
    return 0;
}
