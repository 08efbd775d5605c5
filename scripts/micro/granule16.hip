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
            { float a4[4] = {p, 0.f, 0.f, 0.f};
#pragma unroll
              for (int i = 0; i < WORK; ++i) a4[i & 3] = __builtin_fmaf(a4[i & 3], 0.99999f, 1e-7f);      // the rest of the wave's 128 packed FMAs
              p = (a4[0] + a4[1]) + (a4[2] + a4[3]); }
            part[par][wave][lane] = p;
        } else {
            { float a4[4] = {acc, acc * 0.5f, acc * 0.25f, acc * 0.125f};
#pragma unroll
              for (int i = 0; i < WORK + 32; ++i) a4[i & 3] = __builtin_fmaf(a4[i & 3], 0.99999f, h * 1e-7f);
              acc = (a4[0] + a4[1]) + (a4[2] + a4[3]); }
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

// TWO ROWS PER GROUP, phase-shifted (round 5 probe): the group of four workgroups serves rows A and B alternately -- a half-step
// multiplies and runs the cell of ONE row while the other row's published values travel: the exchange latency of a row hides
// behind the other row's half-step.  128 workgroups serve the 64 (row, direction) pairs that 256 serve today.  POLL_AHEAD: the
// polling waves request the NEXT half-step's granules before this half-step's FMAs (the load's own round trip off the chain).
template <int WORK, int POLL_AHEAD>
__global__ __launch_bounds__(512) void k2(u64* buf, int steps, u64* out, int* err) {
    __shared__ float hs[2][8][36];
    __shared__ float part[2][8][64];
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int group = xcd * 4 + (idx >> 2), mem = idx & 3;           // 32 groups of 4
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u64* gb = buf + (size_t)group * 2 * 2 * 256;                     // [row][parity][256 units]
    const int src = (mem + (wave >> 1)) & 3, half = wave & 1;
    float h[2] = {(float)lane * 1e-3f, (float)lane * 2e-3f}, acc = 0.f;
    __syncthreads();
    const u64 t0 = __builtin_amdgcn_s_memtime();
    u64 pre = 0; bool have_pre = false;
    for (int hs_i = 2; hs_i < 2 * steps + 2; ++hs_i) {           // half-step: row = hs_i & 1, that row's step s = hs_i >> 1
        const int rowx = hs_i & 1, s = hs_i >> 1;
        if (wave != 0) {
            float v0 = 0.f;
            if (lane < 32) {
                const u64* g = gb + ((size_t)rowx * 2 + ((s - 1) & 1)) * 256 + src * 64 + half * 32 + lane;
                if (s > 1) {
                    u64 x = pre;
                    bool got = have_pre && (unsigned)(x >> 32) == (unsigned)(s - 1);
                    for (unsigned spins = 0; !got; ++spins) {
                        asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(g) : "memory");
                        got = (unsigned)(x >> 32) == (unsigned)(s - 1);
                        if (spins > 4000000u) { *err = 1; break; }
                    }
                    v0 = __uint_as_float((unsigned)x);
                }
                hs[rowx][wave][lane] = v0;
                if (POLL_AHEAD) {      // request the other row's granule of ITS current step now; judged at the next half-step
                    const int ry = rowx ^ 1, sy = (hs_i + 1) >> 1;
                    const u64* gn = gb + ((size_t)ry * 2 + ((sy - 1) & 1)) * 256 + src * 64 + half * 32 + lane;
                    asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=&v"(pre) : "v"(gn) : "memory");
                    have_pre = sy > 1;
                }
            }
            __builtin_amdgcn_wave_barrier();
            float p = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) p = __builtin_fmaf(hs[rowx][wave][i], 1.0001f + (float)i * 1e-6f, p);
            { float a4[4] = {p, 0.f, 0.f, 0.f};
#pragma unroll
              for (int i = 0; i < WORK; ++i) a4[i & 3] = __builtin_fmaf(a4[i & 3], 0.99999f, 1e-7f);
              p = (a4[0] + a4[1]) + (a4[2] + a4[3]); }
            part[rowx][wave][lane] = p;
            if (POLL_AHEAD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            { float a4[4] = {acc, acc * 0.5f, acc * 0.25f, acc * 0.125f};
#pragma unroll
              for (int i = 0; i < WORK + 32; ++i) a4[i & 3] = __builtin_fmaf(a4[i & 3], 0.99999f, h[rowx] * 1e-7f);
              acc = (a4[0] + a4[1]) + (a4[2] + a4[3]); }
            part[rowx][0][lane] = acc;
        }
        __syncthreads();
        if (wave == 0) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) sum += part[rowx][w][lane];
            float hh = 1.f / (1.f + __expf(-sum * 1e-3f));
            hh = hh * (1.f - 2.f / (__expf(2.f * sum * 1e-3f) + 1.f));
            h[rowx] = hh;
            u64* dst = gb + ((size_t)rowx * 2 + (s & 1)) * 256 + mem * 64 + lane;
            const u64 gv = ((u64)(unsigned)s << 32) | __float_as_uint(hh);
            asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst), "v"(gv) : "memory");
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
    if (threadIdx.x == 0 && h[0] + h[1] == 12345.f) out[1] = 1;
}

template <int WORK, int POLL_AHEAD>
static void run2(const char* name, int steps) {
    u64 *buf, *out; int* err;
    const size_t nb = (size_t)32 * 2 * 2 * 256 * 8;
    hipMalloc(&buf, nb); hipMalloc(&out, 64); hipMalloc(&err, 4); hipMemset(err, 0, 4); hipMemset(out, 0, 64);
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        hipMemset(buf, 0, nb);
        hipLaunchKernelGGL((k2<WORK, POLL_AHEAD>), dim3(128), dim3(512), 0, 0, buf, steps, out, err);
        hipDeviceSynchronize();
        u64 o[2]; int e;
        hipMemcpy(o, out, 16, hipMemcpyDeviceToHost); hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
        if (e) { printf("%-60s SPIN GUARD HIT\n", name); return; }
        const double t = (double)o[0] / steps;
        if (t < best) best = t;
    }
    printf("%-60s work %3d  %7.1f shader-clock ticks per step of BOTH rows\n", name, WORK, best);
    hipFree(buf); hipFree(out); hipFree(err);
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
    // two rows per group, phase-shifted, on HALF the workgroups (128): ticks for one step of BOTH rows -- compare with the
    // one-row numbers above (256 workgroups, one step of one row each)
    run2<0, 0>("two rows per group, phase-shifted, bare", steps);
    run2<96, 0>("two rows per group, phase-shifted + the product's FMAs", steps);
    run2<0, 1>("two rows per group, polls requested a half-step ahead, bare", steps);
    run2<96, 1>("two rows, polls a half-step ahead + the product's FMAs", steps);
    run<0, 96>("(again) one row, 8-byte granules + the product's FMAs", steps);
    return 0;
}
