// Tagged 8-byte granules {tag, value}: the exchange primitive of the persistent decoder kernels
// (decoder_chain.hip, decoder_greedy.hip).  A value is published with ONE store and polled, two granules at a time,
// with 16-byte sc1 loads; polls are bounded (~2 s of wall clock) and raise the device error flag on a timeout.
#pragma once
#include "common.h"

namespace asr {

typedef unsigned int u32x4c __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool chain_poll2(const u64* g, uint32_t epoch, float& v0, float& v1, int* err) {
    ASR_RACE_HUNT_DELAY();
    long long t0 = 0;
    const u32x4c* p = reinterpret_cast<const u32x4c*>(g);
    for (uint32_t spins = 0;; ++spins) {
        u32x4c x;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(p) : "memory");
        if (x.y == epoch && x.w == epoch) { v0 = __uint_as_float(x.x); v1 = __uint_as_float(x.z); return true; }
        ASR_POLL_BACKOFF();
        if ((spins & 1023) == 1023) {
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > 200000000LL) { *err = 41; v0 = v1 = 0.f; return false; }
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { v0 = v1 = 0.f; return false; }
        }
    }
}
__device__ __forceinline__ void chain_publish(u64* dst, uint32_t epoch, float v, bool fast) {
    ASR_RACE_HUNT_DELAY();
    const u64 gv = ((u64)epoch << 32) | __float_as_uint(v);
    if (fast) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst), "v"(gv) : "memory");
    else __hip_atomic_store(dst, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// two adjacent granules (16-byte aligned) with one store on the fast path
__device__ __forceinline__ void chain_publish2(u64* dst, uint32_t epoch, float v0, float v1, bool fast) {
    ASR_RACE_HUNT_DELAY();
    if (fast) {
        const u32x4c q = {__float_as_uint(v0), epoch, __float_as_uint(v1), epoch};
        asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(q) : "memory");
    } else {
        __hip_atomic_store(dst, ((u64)epoch << 32) | __float_as_uint(v0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 1, ((u64)epoch << 32) | __float_as_uint(v1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- tagged floats: the value itself carries a 1-bit tag in its lowest mantissa bit (it is truncated to 23 mantissa bits;
// every consumer, the owner included, uses the truncated value).  A slot of the parity buffer `step & 1` is rewritten every
// second step, so the bit ((step >> 1) & 1) ^ 1 -- 1 for the first write after the host's memset to zero -- tells this
// step's write from the previous one.  Four values = one 16-byte load: half the bytes and loads of {tag32, value32}
// granules, which matters because every workgroup of a group reads every value (csrc/lstm_bwd.hip measured -16 %).
// Valid within ONE launch over a workspace zeroed before it.
__device__ __forceinline__ uint32_t tag_bit(int step) { return ((((uint32_t)step) >> 1) & 1u) ^ 1u; }
__device__ __forceinline__ void tagged_publish(uint32_t* dst, uint32_t tb, float v, bool fast) {
    ASR_RACE_HUNT_DELAY();
    const uint32_t x = (__float_as_uint(v) & ~1u) | tb;
    if (fast) asm volatile("global_store_dword %0, %1, off" :: "v"(dst), "v"(x) : "memory");
    else __hip_atomic_store(dst, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void tagged_publish2(uint32_t* dst, uint32_t tb, float v0, float v1, bool fast) {   // 8-byte aligned
    ASR_RACE_HUNT_DELAY();
    const uint32_t x0 = (__float_as_uint(v0) & ~1u) | tb, x1 = (__float_as_uint(v1) & ~1u) | tb;
    if (fast) {
        const u64 q = ((u64)x1 << 32) | x0;
        asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst), "v"(q) : "memory");
    } else {
        __hip_atomic_store(dst, x0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 1, x1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// four tagged floats (16-byte aligned), polled until all four carry this step's bit
__device__ __forceinline__ bool tagged_poll4(const uint32_t* g, uint32_t tb, float4& v, int* err) {
    ASR_RACE_HUNT_DELAY();
    long long t0 = 0;
    const u32x4c* p = reinterpret_cast<const u32x4c*>(g);
    for (uint32_t spins = 0;; ++spins) {
        u32x4c x;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(p) : "memory");
        if ((x.x & 1u) + (x.y & 1u) + (x.z & 1u) + (x.w & 1u) == 4u * tb) {
            v = make_float4(__uint_as_float(x.x & ~1u), __uint_as_float(x.y & ~1u), __uint_as_float(x.z & ~1u), __uint_as_float(x.w & ~1u));
            return true;
        }
        ASR_POLL_BACKOFF();
        if ((spins & 1023) == 1023) {
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > 200000000LL) { *err = 42; v = make_float4(0.f, 0.f, 0.f, 0.f); return false; }
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { v = make_float4(0.f, 0.f, 0.f, 0.f); return false; }
        }
    }
}

// Several tagged quads per thread with ALL of the thread's loads in flight (one asm statement: loads + wait, "=&v" outputs), re-polled
// together until every needed one carries its tag bit.  The decoder kernels polled a thread's quads one after the other: a phase
// that gathers 512 quads with 384 polling threads paid two L2 round trips, the context + early-LM gather four.
// p[j]: 16-byte aligned; a slot with need[j] == false is not judged (point it at any valid quad); tb[j]: expected bit of slot j.
template <int N>
__device__ __forceinline__ bool tagged_poll4_many(const uint32_t* const* p, bool* need, const uint32_t* tb, float4* v, int* err) {
    ASR_RACE_HUNT_DELAY();
    static_assert(N == 2 || N == 4, "two or four quads per thread");
    long long t0 = 0;
    for (uint32_t spins = 0;; ++spins) {
        u32x4c x[N];
        if constexpr (N == 2)
            asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(x[0]), "=&v"(x[1]) : "v"(p[0]), "v"(p[1]) : "memory");
        else
            asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                         "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory");
        bool pending = false;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (!need[j]) continue;
            if ((x[j].x & 1u) + (x[j].y & 1u) + (x[j].z & 1u) + (x[j].w & 1u) == 4u * tb[j]) {
                v[j] = make_float4(__uint_as_float(x[j].x & ~1u), __uint_as_float(x[j].y & ~1u), __uint_as_float(x[j].z & ~1u),
                                   __uint_as_float(x[j].w & ~1u));
                need[j] = false;
            } else pending = true;
        }
        if (!pending) return true;
        ASR_POLL_BACKOFF();
        if ((spins & 1023) == 1023) {
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > 200000000LL) { *err = 43; return false; }
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        }
    }
}

}  // namespace asr
