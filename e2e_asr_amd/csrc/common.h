// Shared device/host helpers for the e2e_asr gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ASR_OK 0
#define ASR_EINVAL (-1)
#define ASR_ELAUNCH (-2)
#define ASR_EUNSUPPORTED (-3)
// OR-ed into a workspace size argument (hx_bytes): the caller hands over memory it has ALREADY zeroed (one fill per train step
// over an arena of all the step's exchange workspaces instead of one memset launch in front of every persistent kernel)
#define ASR_WS_PREZEROED ((size_t)1 << 62)

#define ASR_CHECK_LAUNCH()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return ASR_ELAUNCH;          \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

#define ASR_PROF_TAGS 8
#define ASR_PROF_LSTM_REC_FWD 0
#define ASR_PROF_LSTM_REC_BWD 1
#define ASR_PROF_GEMM 2
#define ASR_PROF_DECODER_FWD 3
#define ASR_PROF_DECODER_BWD 4
#define ASR_PROF_OPTIM 5
#define ASR_PROF_SIDE_TAIL 6      // asr_side_join: caller's stream at the join -> end of the side stream's work

// Optional back-off between two polls of a granule that has not arrived (s_sleep units of 64 cycles; 0 = none, the
// default: measured, a back-off of 1 makes the forward recurrence 6 % slower and changes nothing else): a spinning wave's
// back-to-back sc1 loads occupy the CU's memory pipeline that the publishing wave of the same workgroup needs.
#ifndef ASR_POLL_SLEEP
#define ASR_POLL_SLEEP 0
#endif
#define ASR_POLL_BACKOFF() do { if (ASR_POLL_SLEEP) __builtin_amdgcn_s_sleep(ASR_POLL_SLEEP); } while (0)

// ---- ASR_RACE_HUNT (debug build of the persistent kernels only: libe2e_asr_hip_hunt.so, __graft_entry__.build()) --------------
// Every exchange of the persistent kernels is "one store, polled; the data is its own flag".  Two bugs of that class were found
// by accident (round 2: overlapping persistent launches; round 4: an exchange numbered by the step index although it only ran at
// feedback steps, so a poller could take the memset's zeros).  The hunt build makes such bugs show on purpose: in front of
// every PUBLISH and every POLL a wave tosses a coin (shader clock x workgroup id, wave-uniform) and one wave in eight sleeps
// ~4 us (s_sleep 127 = 8 128 cycles; a whole decoder step is 6-8 us, a recurrent step ~1 us).  A late publisher sends pollers
// into slots that still hold the previous contents; a late poller lets its publishers run ahead and rewrite a slot of the
// two-deep parity buffers -- the two ways a tag protocol can be wrong.  tests/test_gpu_race_hunt.py runs the
// equal-to-launch-path tests of every persistent kernel on this build.  Empty in the product build.
#ifndef ASR_RACE_HUNT
#define ASR_RACE_HUNT 0
#endif
#if ASR_RACE_HUNT
#define ASR_RACE_HUNT_DELAY() do {                                                                          \
        const unsigned h__ = (((unsigned)__builtin_amdgcn_s_memtime() >> 2) ^ (blockIdx.x * 0x9E3779B1u)) * 0x85EBCA6Bu; \
        if ((h__ >> 29) == 0u) __builtin_amdgcn_s_sleep(127);                                              \
    } while (0)
#else
#define ASR_RACE_HUNT_DELAY() do {} while (0)
#endif

namespace asr {

void prof_begin(int tag, hipStream_t s);   // prof.hip
void prof_end(int tag, hipStream_t s);
// For a family that is ONE kernel launch: the events to hand to hipExtLaunchKernelGGL as its start / stop events (the dispatch's
// own signals: no marker packets in front of and behind the kernel, which cost the stream ~3 us each and more when another
// stream waits on them -- scripts/micro/fork_cost.hip), or nullptrs when the family is not being recorded.
void prof_launch_events(int tag, hipEvent_t* start, hipEvent_t* stop);
// Workgroups of ONE persistent launch that are certainly co-resident: the compute units of the current device (each of
// the persistent kernels is one 512-thread workgroup per CU), optionally lowered by ASR_LSTM_MAXWG.  Every persistent
// launch sizes its grid by this; a group that does not fit returns ASR_EUNSUPPORTED instead of spinning into the
// exchange timeout (prof.hip).
int resident_wg_budget();
// splitk.hip: split-K partial tiles through per-stream slabs + a fixed-order reduce instead of float atomics (opt-in deterministic mode)
int wgrad_slabs();
float* slab_arena(hipStream_t s, size_t bytes, int which = 0);
int transpose_add(hipStream_t s, float* C, int ldc, const float* T, int M, int N, int accumulate);    // C[m][n] (+)= T[n][m], T [N][M]
struct SlabMap {
    int M, N, nsl, nsl_alloc, batch;       // slab s of batch z at ((z * nsl_alloc + s) * M) * N floats (N % 4 == 0); nsl of them hold data
    int Nvalid;                            // columns n >= Nvalid of a slab row are padding (never read)
    int mA, mA_valid;                      // product row m -> C row (plain products: mA = mA_valid = M)
    const int* colmap;                     // product column n -> C column (NULL: identity)
    int ldc; long long zC;                 // C row pitch, batch stride (elements)
    int accumulate;                        // C += sum (else C = sum)
};
int slab_reduce(hipStream_t s, float* C, const float* slab, const SlabMap& q);
hipStream_t side_stream();                 // prof.hip: library-owned side stream + pooled events
hipEvent_t next_event();
void set_pending_join(hipEvent_t e);

// ---- math: v_exp_f32 / v_rcp_f32 based, ~1-2 ulp; saturate cleanly at +-inf ----
// (__frcp_rn would expand to the ~10-instruction IEEE divide; v_rcp_f32 is 1 ulp and one instruction)
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sigmoid(float x) {
    return fast_rcp(1.0f + __expf(-x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    // tanh(x) = 1 - 2/(exp(2x)+1); exp->inf gives 1, exp->0 gives -1.
    return 1.0f - 2.0f * fast_rcp(__expf(2.0f * x) + 1.0f);
}

// ---- DPP butterflies inside a row of 16 lanes (no LDS) ----
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// All 16 lanes of each DPP row end up holding the row's sum.
__device__ __forceinline__ float row16_allreduce_sum(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]  (xor 1)
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]  (xor 2)
    v += dpp_mov<0x141>(v);   // row_half_mirror      (joins the two quads of each 8)
    v += dpp_mov<0x140>(v);   // row_mirror           (joins the two halves of the 16)
    return v;
}
// Four values at once through fused v_add_f32_dpp (round 5): the compiler turns four row16_allreduce_sum calls into 16 v_mov_b32_dpp
// + 8 v_pk_add_f32 per butterfly step... per 16 values, i.e. 1.5 instructions per value and step; fused it is 1 (the matvec
// phases of the decoder kernels reduce 32 values per thread and step: a knock-out build without three of the four steps ran the
// decoder forward 5 % faster).  A DPP read of a VGPR needs 2 wait states behind the VALU write, which the hardware does not
// interlock and the compiler cannot see into inline asm: inside the block every instruction reads a register written four
// instructions earlier, and an s_nop 1 stands at either end.  Same operands meet in the same order: bit-identical sums.
__device__ __forceinline__ void row16_allreduce_sum4(float& a, float& b, float& c, float& d) {
#define ASR_DPP4(ctrl) "v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
                       "v_add_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
                       "v_add_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
                       "v_add_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm("s_nop 1\n\t" ASR_DPP4("quad_perm:[1,0,3,2]") ASR_DPP4("quad_perm:[2,3,0,1]") ASR_DPP4("row_half_mirror") ASR_DPP4("row_mirror") "s_nop 1"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef ASR_DPP4
}
// The value the lane 16 (32) away holds -- what `__shfl_xor(v, 16)` / `(v, 32)` return -- by v_permlane16_swap / v_permlane32_swap
// (gfx950) instead of ds_bpermute: two VALU instructions, no trip through the LDS crossbar (round 3: the cross-row steps of the
// wave-wide reductions below were two dependent ds_bpermute round trips each; the softmax phases of the decoder kernels pay
// two such reductions per step).  Same operands meet, so sums and maxima are bit-identical to the shuffle form.
__device__ __forceinline__ float lane_xor16(float v) {
    const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(((__lane_id() >> 4) & 1) ? s[0] : s[1]);      // odd rows find the even neighbour in s[0], even rows the odd one in s[1]
}
__device__ __forceinline__ float lane_xor32(float v) {
    const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(__lane_id() >= 32 ? s[0] : s[1]);
}
__device__ __forceinline__ float wave_allreduce_sum(float v) {
    v = row16_allreduce_sum(v);
    v += lane_xor16(v);
    v += lane_xor32(v);
    return v;
}
__device__ __forceinline__ float wave_allreduce_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));    // the four DPP row steps of row16_allreduce_sum, with max
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    v = fmaxf(v, lane_xor16(v));
    v = fmaxf(v, lane_xor32(v));
    return v;
}


// ---- packed fp32 FMA and half-wave exchange (version-2 recurrent kernels, csrc/lstm.hip / lstm_bwd.hip) ----
typedef float f32x2 __attribute__((ext_vector_type(2)));
// acc.lo += a.lo * b.{lo|hi}; acc.hi += a.hi * b.{lo|hi}: two fp32 FMAs per issue slot, bitwise the same as two v_fma_f32
__device__ __forceinline__ void pk_fma_blo(f32x2& acc, f32x2 a, f32x2 b) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void pk_fma_bhi(f32x2& acc, f32x2 a, f32x2 b) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(a), "v"(b));
}
// acc.lo += a.{lo|hi} * b.lo; acc.hi += a.{lo|hi} * b.hi
__device__ __forceinline__ void pk_fma_alo(f32x2& acc, f32x2 a, f32x2 b) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void pk_fma_ahi(f32x2& acc, f32x2 a, f32x2 b) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(a), "v"(b));
}
// the same with `a` a wave-uniform pair in scalar registers (e.g. an input row fetched with s_load)
__device__ __forceinline__ void pk_fma_alo_s(f32x2& acc, f32x2 a, f32x2 b) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(a), "v"(b));
}
__device__ __forceinline__ void pk_fma_ahi_s(f32x2& acc, f32x2 a, f32x2 b) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(a), "v"(b));
}
// lanes 32-63 of x <-> lanes 0-31 of y (v_permlane32_swap), then x + y: the lower half-wave ends with x summed over the two
// halves, the upper half-wave with y summed over the two halves
__device__ __forceinline__ float swap32_add(float x, float y) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}

// rows 1, 3 of x <-> rows 0, 2 of y (rows of 16 lanes; v_permlane16_swap), then x + y: even rows end with x summed over the
// row pair, odd rows with y summed over the row pair
__device__ __forceinline__ float swap16_add(float x, float y) {
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}

// ---- agent-scope accesses -------------------------------------------------------------------------------------------
// A kernel whose phases run in workgroups of different XCDs WITHOUT a kernel boundary between them (csrc/beam.hip) cannot
// use plain loads and stores for what one phase hands to the next: the L2s of two XCDs are not coherent for them.  Relaxed
// agent-scope atomics compile to `sc1` loads (served by the coherence point, never by a stale line) and `sc1` write-through
// stores.  Mode COH of the tile bodies:
//   0  plain accesses (the launch-per-step kernels: a kernel boundary orders everything);
//   1  every hand-over load and store agent-scope (correct for buffers that are REWRITTEN in place; slow: an uncached
//      4-byte request per lane and word -- 15 us per phase of the beam kernel);
//   2  write-once buffers: DATA (activations, states, logits) is stored agent-scope to an address nobody has read before
//      (a ring slot per token) and loaded PLAIN -- no L1 or L2 can hold a stale copy of a line that was never read --
//      while the few CONTROL words that live at fixed addresses (tokens, parent rows, counters) stay agent-scope.
template <int COH> __device__ __forceinline__ float ld_data(const float* p) {
    if constexpr (COH == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else return *p;
}
template <int COH> __device__ __forceinline__ float4 ld_data4(const float* p) {
    if constexpr (COH == 1) return make_float4(ld_data<1>(p), ld_data<1>(p + 1), ld_data<1>(p + 2), ld_data<1>(p + 3));
    else return *reinterpret_cast<const float4*>(p);
}
template <int COH> __device__ __forceinline__ float ld_f(const float* p) {
    if constexpr (COH != 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else return *p;
}
template <int COH> __device__ __forceinline__ int ld_i(const int* p) {
    if constexpr (COH != 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else return *p;
}
template <int COH> __device__ __forceinline__ double ld_d(const double* p) {
    if constexpr (COH != 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else return *p;
}
template <int COH> __device__ __forceinline__ void st_f(float* p, float v) {
    if constexpr (COH != 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
template <int COH> __device__ __forceinline__ void st_i(int* p, int v) {
    if constexpr (COH != 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
template <int COH> __device__ __forceinline__ void st_d(double* p, double v) {
    if constexpr (COH != 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}

// ---- same-XCD agreement (speed only; correctness never depends on placement) -------------------
// Every workgroup of a group publishes its HW_REG_XCC_ID through the placement-independent sc1
// protocol; if all G ids are equal the group's granules may be published with PLAIN stores: the line
// then stays in that XCD's L2, where the peers' L1-bypassing sc1 loads find it (measured -12 % per
// recurrent step), instead of being written through to the fabric.  Any other placement keeps sc1.
__device__ __forceinline__ bool group_shares_xcd(u64* slots, int G, int mem, int tid, int* err, int* lds_flag = nullptr,
                                                 uint32_t epoch = 0, int kid = 0) {
    // epoch: launches that share zeroed-once slots (segments of one decoder call) use distinct tags
    const u64 tagv = 0xA5A50000ull + (epoch & 0xFFFFu);
    __shared__ int s_same_static;
    int* s_same_p = lds_flag ? lds_flag : &s_same_static;   // kernels with dynamic LDS pass their own word (G17)
#define s_same (*s_same_p)
    if (tid == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xF;   // HW_REG_XCC_ID[3:0]
        __hip_atomic_store(slots + mem, (tagv << 32) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool same = true;
        long long t0 = wall_clock64();
        for (int m = 0; m < G && same; ++m) {
            for (;;) {
                const u64 x = __hip_atomic_load(slots + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((x >> 32) == tagv) { same = ((uint32_t)x == xcc); break; }
                if (wall_clock64() - t0 > 200000000LL) { *err = 31 + 100 * kid; same = false; break; }
            }
        }
        s_same = same ? 1 : 0;
    }
    __syncthreads();
    const bool res = s_same != 0;
#undef s_same
    return res;
}

// ---- counter-based uniform [0,1) for dropout (same value in fwd and bwd) ----
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float uniform01(uint32_t seed, uint32_t a, uint32_t b) {
    uint32_t h = mix32(seed ^ mix32(a * 0x9E3779B9U + 0x85EBCA6BU) ^ mix32(b + 0xC2B2AE35U));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
// DropoutWrapper(output_keep_prob): 0 or 1/keep.
__device__ __forceinline__ float keep_scale(uint32_t seed, uint32_t a, uint32_t b, float keep) {
    return (keep >= 1.0f) ? 1.0f : (uniform01(seed, a, b) < keep ? 1.0f / keep : 0.0f);
}

}  // namespace asr
