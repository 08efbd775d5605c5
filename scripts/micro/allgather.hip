// Step time of the bare all-gather the recurrent kernels do every time step, polled by VECTOR (sc1) loads vs by SCALAR (glc)
// loads (diagnostic).  256 workgroups x 8 waves; a group = 8 workgroups of one XCD (blockIdx & 7 under round-robin dispatch);
// every step wave 0 of a workgroup publishes 64 dwords (plain store) and each of waves 1..7 polls ONE peer's 64 dwords until
// all carry the step number; one barrier per step, slots double-buffered by step parity -- the hand-over pattern of
// lstm_rec_fwd2_kernel with the arithmetic left out; NOISE adds the memory traffic the real kernels have beside the exchange
// (see the kernel).
// hipcc --offload-arch=gfx950 -O3 scripts/micro/allgather.hip -o scripts/micro/allgather
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u16v __attribute__((ext_vector_type(16)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

// the published word is all-ones or zero, alternating per use of a slot (the 1-bit tag of granule.h stretched over the word):
// "all 64 arrived" = AND of the words is all-ones (tag 1) / OR of the words is zero (tag 0) -- bit operations stay on the SALU
__device__ __forceinline__ unsigned and16(u16v a) {
    unsigned m = a[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m &= a[i];
    return m;
}
__device__ __forceinline__ unsigned or16(u16v a) {
    unsigned m = a[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m |= a[i];
    return m;
}

// NOISE bit 0: a ninth wave streams records like the BPTT's loader (36 bytes per lane and step from lines nobody has touched,
// two register sets, consumed two steps after the request); bit 1: wave 0 follows its publish with four dword stores per
// lane to four streaming lines (the BPTT's dG bookkeeping); bit 2: the loader's lines were touched 8 steps earlier by
// scalar loads of a tenth wave.
template <int SCALAR, int NOISE>
__global__ __launch_bounds__(640) void k(unsigned* buf, unsigned* big, size_t nbig, int steps, unsigned long long* out, int* err, int gs) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    // gs = workgroups per group (8, or 4: twice the groups, three peers to hear from instead of seven)
    const int group = gs == 8 ? xcd * 4 + (idx >> 3) : xcd * 8 + (idx >> 2), member = gs == 8 ? (idx & 7) : (idx & 3);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned* gb = buf + (size_t)group * 2 * 8 * 64;
    unsigned acc = 0;
    bool dead = false;            // a wave whose spin guard fired stops polling: the run ends quickly and reports it
    // streaming region of this workgroup: step s -> 4 KB at rec + s * 4096 (records: 64 lanes x 32 B, then 256 B of "dout")
    unsigned* rec = big + (size_t)blockIdx.x * ((size_t)steps + 16) * 1024;
    unsigned* bk = big + nbig / 2 + (size_t)blockIdx.x * ((size_t)steps + 16) * 256;     // bookkeeping stores: 1 KB per step
    u4v ea = {0, 0, 0, 0}, eb = ea, oa = ea, ob = ea; unsigned ed = 0, od = 0;
    u4v ba[8], bb[8]; unsigned bd[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { ba[i] = ea; bb[i] = ea; bd[i] = 0; }
    auto fetch = [&](int s, u4v& a4, u4v& b4, unsigned& d) {
        if (NOISE & 64) s &= 15;         // a 64 KB window per workgroup: misses the 32 KB L1, stays in L2 (2 MB per XCD)
        const unsigned* p = rec + (size_t)s * 1024 + lane * 8;
        a4 = *reinterpret_cast<const u4v*>(p); b4 = *reinterpret_cast<const u4v*>(p + 4); d = rec[(size_t)s * 1024 + 512 + lane];
    };
    if (wave == 8 && (NOISE & 1)) { fetch(1, oa, ob, od); fetch(2, ea, eb, ed); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave == 9) {              // toucher: its own loop, joined to the others by the step barriers only
        unsigned sink = 0;
        for (int s = 1; s <= steps; ++s) {
            if (NOISE & 4) {
                const unsigned* p = rec + (size_t)(s + 8) * 1024;
#pragma unroll
                for (int i = 0; i < ((NOISE & 16) ? 40 : 20); ++i)
                    asm volatile("s_load_dword %0, %1, %2" : "+s"(sink) : "s"(p), "i"(i * ((NOISE & 16) ? 64 : 128)));
            }
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(sink));
        return;
    }
    for (int s = 1; s <= steps; ++s) {
        unsigned* slot = gb + (s & 1) * 8 * 64;
        const bool tag = ((s + 1) >> 1) & 1;            // steps 1,2 -> 1; 3,4 -> 0; ... (the buffer starts zeroed)
        const unsigned word = tag ? ~0u : 0u;
        if (wave == 0) {
            unsigned* p = slot + member * 64 + lane;
            asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(word) : "memory");
            if (NOISE & 2) {
                unsigned* q = bk + (size_t)s * 256 + lane;
                __builtin_nontemporal_store(word, q); __builtin_nontemporal_store(word, q + 64);
                __builtin_nontemporal_store(word, q + 128); __builtin_nontemporal_store(word, q + 192);
            }
        } else if (wave == 8) {
            if (NOISE & 32) {         // the same bytes in BURSTS: every 8th step the records of the 8 steps after the next 8
                if ((s & 7) == 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc += ba[i][0] + bb[i][1] + bd[i];
#pragma unroll
                    for (int i = 0; i < 8; ++i) fetch(s + 8 + i, ba[i], bb[i], bd[i]);
                }
            } else if (NOISE & 8) {          // the same loads, never waited for (results dropped)
                const unsigned* p = rec + (size_t)(s + 2) * 1024 + lane * 8;
                u4v t1, t2; unsigned t3;
                asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:16\n\tglobal_load_dword %2, %4, off"
                             : "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(p), "v"(rec + (size_t)(s + 2) * 1024 + 512 + lane));
            } else if (NOISE & 1) {
                if (s & 1) { acc += oa[0] + ob[1] + od; fetch(s + 2, oa, ob, od); }
                else { acc += ea[0] + eb[1] + ed; fetch(s + 2, ea, eb, ed); }
            }
        } else if (!dead && wave < gs) {
            const int src = (member + wave) & (gs - 1);
            const unsigned* p = slot + src * 64;
            int guard = 0;
            if (SCALAR == 1) {
                for (;;) {
                    u16v a, b, c, d;
                    asm volatile("s_load_dwordx16 %0, %4, 0x0 glc\n\ts_load_dwordx16 %1, %4, 0x40 glc\n\t"
                                 "s_load_dwordx16 %2, %4, 0x80 glc\n\ts_load_dwordx16 %3, %4, 0xc0 glc\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d) : "s"(p) : "memory");
                    const bool all = tag ? and16(a & b & c & d) == ~0u : or16(a | b | c | d) == 0u;
                    if (all) { acc += a[3] + d[15]; break; }
                    if (++guard > 200000) { *err = 1; dead = true; break; }
                }
            } else {
                for (;;) {
                    u4v x = {0u, 0u, 0u, 0u};
                    if (lane < 16) {
                        if (SCALAR == 2)        // L1 invalidated, then a PLAIN load: an ordinary L2 read, not an agent-scope one
                            asm volatile("buffer_inv sc1\n\tglobal_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(p + lane * 4) : "memory");
                        else if (SCALAR == 3)
                            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(p + lane * 4) : "memory");
                        else
                            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(p + lane * 4) : "memory");
                    }
                    const bool mine = lane >= 16 || (tag ? (x[0] & x[1] & x[2] & x[3]) == ~0u : (x[0] | x[1] | x[2] | x[3]) == 0u);
                    if (__builtin_amdgcn_ballot_w64(!mine) == 0) { acc += x[1]; break; }
                    if (++guard > 200000) { *err = 1; dead = true; break; }
                }
            }
        }
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
    if (acc == 0x12345678u) out[1] = acc;
}

int main() {
    const size_t nbig = (size_t)1 << 30;     // 4 GiB of unsigned
    unsigned *buf, *big; unsigned long long* out; int* err;
    (void)hipMalloc(&buf, 64 * 2 * 8 * 64 * 4); (void)hipMalloc(&big, nbig * 4); (void)hipMemset(big, 0, nbig * 4);
    (void)hipMallocManaged(&out, 64); (void)hipMallocManaged(&err, 4);
    const int steps = 2000;        // streaming region: 256 workgroups x 2016 steps x 4 KB = 2.1 GB... halves of `big` (4 GiB)
#define RUN(S, N) { (void)hipMemset(buf, 0, 64 * 2 * 8 * 64 * 4); *err = 0; (void)hipDeviceSynchronize(); \
        k<S, N><<<256, 640>>>(buf, big, nbig, steps, out, err, GS); (void)hipDeviceSynchronize(); \
        printf("groups of %d  %-18s %-70s %6.0f ticks per step%s\n", GS, pn[S], nm[N], (double)out[0] / steps, *err ? "  (SPIN GUARD HIT)" : ""); fflush(stdout); }
    const char* nm[128] = {};
    const char* pn[4] = {"vector sc1 polls", "scalar glc polls", "buffer_inv sc1 + plain", "vector sc0 sc1 polls"};
    nm[0] = ""; nm[1] = "+ loader stream"; nm[2] = "+ bookkeeping stores"; nm[3] = "+ loader stream + bookkeeping stores";
    nm[5] = "+ loader stream, lines touched 8 steps ahead by scalar loads (one per 128 B)";
    nm[21] = "+ loader stream, lines touched 8 steps ahead by scalar loads (one per 64 B)";
    nm[8] = "+ loader stream never waited for";
    nm[65] = "+ loader stream over a 64 KB window (L1 misses, L2 hits)";
    nm[32] = "+ loader stream in bursts: 8 steps' records every 8th step";
    int GS = 8;
    for (int rep = 0; rep < 2; ++rep, GS = 4) { RUN(0, 0) RUN(3, 0) RUN(1, 0) RUN(2, 0) RUN(0, 2) RUN(0, 65) RUN(0, 1) RUN(0, 8) RUN(0, 32) RUN(0, 5) RUN(0, 21) }
    return 0;
}
