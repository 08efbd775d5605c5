// Does a load that misses to HBM delay ANOTHER wave's L2-hitting (sc1) loads on the same CU?  (diagnostic)
// One workgroup, two waves: wave 1 measures the latency of dependent sc1 loads to one resident line; wave 0 meanwhile
// (mode 0) idles, (mode 1) streams vector loads through a 1 GiB buffer (HBM misses), (mode 2) the same stream by scalar loads,
// (mode 3) vector loads hitting a 64 KB window (L2 hits).
// hipcc --offload-arch=gfx950 -O3 scripts/micro/tcp_order.hip -o scripts/micro/tcp_order
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(128) void k(const unsigned* big, size_t nbig, unsigned* hot, int mode, int spoll, int iters, unsigned long long* out, volatile int* stop) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 1) {
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        unsigned x = 0;
        for (int i = 0; i < iters; ++i) {
            if (spoll) {           // the measuring wave polls by SCALAR loads (glc: served by L2, not by the scalar cache)
                const unsigned* p = hot + (__builtin_amdgcn_readfirstlane(x) & 1);
                unsigned sx;
                asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(sx) : "s"(p) : "memory");
                x = sx;
                continue;
            }
            const unsigned* p = hot + (x & 1);
            asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(p) : "memory");
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { out[0] = t1 - t0; out[1] = x; *stop = 1; }
    } else {
        size_t off = (size_t)blockIdx.x * 7919 * 64 + lane * 32;      // 128 B per lane: every lane its own line
        unsigned acc = 0;
        int guard = 0;
        while (!*stop && ++guard < 4000000) {
            if (mode == 1) { acc += big[off % nbig]; off += 64 * 32 * 97; }
            else if (mode == 3) { acc += big[(off % 16384)]; off += 64 * 32; }
            else if (mode == 2) {
                const unsigned* p = big + (__builtin_amdgcn_readfirstlane((unsigned)(off % nbig)) & ~31u);
                unsigned v;
                asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
                acc += v; off += 64 * 32 * 97;
            } else __builtin_amdgcn_s_sleep(8);
        }
        if (acc == 0x12345678u) out[2] = acc;
    }
}
int main() {
    const size_t nbig = (size_t)1 << 28;      // 1 GiB of unsigned
    unsigned *big, *hot; unsigned long long* out; int* stop;
    (void)hipMalloc(&big, nbig * 4); (void)hipMemset(big, 0, nbig * 4); (void)hipMalloc(&hot, 4096); (void)hipMemset(hot, 0, 4096);
    (void)hipMallocManaged(&out, 64); (void)hipMallocManaged(&stop, 4);
    const char* nm[4] = {"other wave idle", "other wave: vector loads missing to HBM", "other wave: SCALAR loads missing to HBM", "other wave: vector loads hitting L2"};
    for (int spoll = 0; spoll < 2; ++spoll)
    for (int mode = 0; mode < 4; ++mode) {
        *stop = 0; (void)hipDeviceSynchronize();
        k<<<1, 128>>>(big, nbig, hot, mode, spoll, 20000, out, stop); (void)hipDeviceSynchronize();
        printf("%-42s  %.0f ticks per dependent %s load\n", nm[mode], (double)out[0] / 20000, spoll ? "scalar glc" : "sc1"); fflush(stdout);
    }
    return 0;
}
