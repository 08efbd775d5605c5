// Pricing of a 16-byte {tag32, v, v, v} granule for the FORWARD recurrence's exchange (review item, round 4) against the
// 8-byte {tag32, v} granule lstm_rec_fwd4_kernel uses: a 64-unit slice travels in 352 instead of 512 bytes, a polling wave
// reads 176 instead of 256.  The hand-over pattern of lstm_rec_fwd4_kernel with the arithmetic replaced by a fixed FMA chain:
// 256 workgroups x 8 waves, groups of 4 workgroups on one XCD, one batch row per group; per step wave 0 ("cell wave")
// publishes its 64 values with ONE store instruction, waves 1..7 each poll one half slice (32 values) of a source workgroup
// until the step's tag shows, stage it in a wave-private LDS row, multiply (dummy FMAs), one barrier, wave 0 sums the
// partials.  MODE 0: 8-byte granules (64 lanes store, 32 lanes poll 8 B).  MODE 1: 16-byte granules (22 lanes store three
// values each, gathered from their owners' lanes by three ds_bpermute; 11 lanes poll 16 B per half slice).
// hipcc --offload-arch=gfx950 -O3 scripts/micro/granule16.hip -o scripts/micro/granule16 && scripts/micro/granule16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int WORK>
__global__ __launch_bounds__(512) void k(u64* buf, int steps, u64* out, int* err) {
    __shared__ float hs[8][36];
    __shared__ float part[2][8][64];
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int group = xcd * 8 + (idx >> 2), mem = idx & 3;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // MODE 0: [2 parities][256 units] 8-byte granules; MODE 1: [2][4 members][2 halves][11 quads] 16-byte granules (padded to 12)
    u64* gb = buf + (size_t)group * 2 * 256;
    const int src = (mem + (wave >> 1)) & 3, half = wave & 1;
    float h = (float)lane * 1e-3f, acc = 0.f;
    __syncthreads();
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int s = 1; s <= steps; ++s) {
        const int par = s & 1;
        if (wave != 0) {
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
            if (MODE == 0) {
                if (lane < 32) {
                    const u64* g = gb + (size_t)((s - 1) & 1) * 256 + src * 64 + half * 32 + lane;
                    if (s > 1) {
                        for (unsigned spins = 0;; ++spins) {
                            u64 x;
                            asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(g) : "memory");
                            if ((unsigned)(x >> 32) == (unsigned)(s - 1)) { v0 = __uint_as_float((unsigned)x); break; }
                            if (spins > 4000000u) { *err = 1; break; }
                        }
                    }
                    hs[wave][lane] = v0;
                }
            } else {
                if (lane < 11) {
                    const u32x4* g = reinterpret_cast<const u32x4*>(gb) + (size_t)((s - 1) & 1) * 128 + (src * 2 + half) * 12 + lane;
                    if (s > 1) {
                        for (unsigned spins = 0;; ++spins) {
                            u32x4 x;
                            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(g) : "memory");
                            if (x.x == (unsigned)(s - 1)) { v0 = __uint_as_float(x.y); v1 = __uint_as_float(x.z); v2 = __uint_as_float(x.w); break; }
                            if (spins > 4000000u) { *err = 1; break; }
                        }
                    }
                    hs[wave][3 * lane] = v0; hs[wave][3 * lane + 1] = v1; hs[wave][3 * lane + 2] = v2;
                }
            }
            __builtin_amdgcn_wave_barrier();
            float p = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) p = __builtin_fmaf(hs[wave][i], 1.0001f + (float)i * 1e-6f, p);
#pragma unroll
            for (int i = 0; i < WORK; ++i) p = __builtin_fmaf(p, 0.99999f, 1e-7f);      // the rest of the wave's 128 packed FMAs
            part[par][wave][lane] = p;
        } else {
#pragma unroll
            for (int i = 0; i < WORK + 32; ++i) acc = __builtin_fmaf(acc, 0.99999f, h * 1e-7f);
            part[par][0][lane] = acc;
        }
        __syncthreads();
        if (wave == 0) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) sum += part[par][w][lane];
            h = 1.f / (1.f + __expf(-sum * 1e-3f));                                  // "the cell"
            h = h * (1.f - 2.f / (__expf(2.f * sum * 1e-3f) + 1.f));
            if (MODE == 0) {
                u64* dst = gb + (size_t)par * 256 + mem * 64 + lane;
                const u64 gv = ((u64)(unsigned)s << 32) | __float_as_uint(h);
                asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst), "v"(gv) : "memory");
            } else {
                // lane q < 22 gathers units 32 * (q / 11) + 3 * (q % 11) .. + 2 of the slice from their owners
                const int q = lane < 22 ? lane : 0, hh = q / 11, qq = q % 11;
                const int u0 = hh * 32 + 3 * qq;
                const float a0 = __shfl(h, u0), a1 = __shfl(h, min(u0 + 1, 63)), a2 = __shfl(h, min(u0 + 2, 63));
                if (lane < 22) {
                    u32x4* dst = reinterpret_cast<u32x4*>(gb) + (size_t)par * 128 + (mem * 2 + hh) * 12 + qq;
                    const u32x4 gv = {(unsigned)s, __float_as_uint(a0), __float_as_uint(a1), __float_as_uint(a2)};
                    asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(gv) : "memory");
                }
            }
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
    if (threadIdx.x == 0 && h == 12345.f) out[1] = 1;
}

template <int MODE, int WORK>
static void run(const char* name, int steps) {
    u64 *buf, *out; int* err;
    hipMalloc(&buf, (size_t)64 * 2 * 256 * 8); hipMemset(buf, 0, (size_t)64 * 2 * 256 * 8);
    hipMalloc(&out, 64); hipMemset(out, 0, 64);
    hipMalloc(&err, 4); hipMemset(err, 0, 4);
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        hipMemset(buf, 0, (size_t)64 * 2 * 256 * 8);
        hipLaunchKernelGGL((k<MODE, WORK>), dim3(256), dim3(512), 0, 0, buf, steps, out, err);
        hipDeviceSynchronize();
        u64 o[2]; int e;
        hipMemcpy(o, out, 16, hipMemcpyDeviceToHost); hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
        if (e) { printf("%-44s SPIN GUARD HIT\n", name); return; }
        const double t = (double)o[0] / steps;
        if (t < best) best = t;
    }
    printf("%-44s work %3d  %7.1f shader-clock ticks per step\n", name, WORK, best);
    hipFree(buf); hipFree(out); hipFree(err);
}

int main() {
    const int steps = 2000;
    run<0, 0>("8-byte {tag, v} granules, bare exchange", steps);
    run<1, 0>("16-byte {tag, v, v, v} granules, bare exchange", steps);
    run<0, 96>("8-byte granules + the product's FMAs", steps);
    run<1, 96>("16-byte granules + the product's FMAs", steps);
    run<0, 0>("8-byte {tag, v} granules, bare exchange", steps);
    run<1, 0>("16-byte {tag, v, v, v} granules, bare exchange", steps);
    return 0;
}
