// Optional HIP-event timing of individual kernels on the stream they are launched on
// (bench.py's roofline leg).  Disabled by default: zero overhead on the product path.
#include "common.h"
#include <cstdlib>
#include <cstdio>
#include <vector>

namespace asr {
struct ProfPool { std::vector<hipEvent_t> a, b; size_t used = 0; };
static ProfPool g_pool[ASR_PROF_TAGS];
int g_prof_on = 0;          // bit t set: events around the launches of family t (asr_prof_enable: all or none; _mask: chosen ones)

void prof_begin(int tag, hipStream_t s) {
    if (!((g_prof_on >> tag) & 1)) return;
    ProfPool& p = g_pool[tag];
    if (p.used == p.a.size()) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        p.a.push_back(e0); p.b.push_back(e1);
    }
    (void)hipEventRecord(p.a[p.used], s);
}
void prof_launch_events(int tag, hipEvent_t* start, hipEvent_t* stop) {
    *start = *stop = nullptr;
    if (!((g_prof_on >> tag) & 1)) return;
    ProfPool& p = g_pool[tag];
    if (p.used == p.a.size()) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        p.a.push_back(e0); p.b.push_back(e1);
    }
    *start = p.a[p.used]; *stop = p.b[p.used];
    p.used++;
}
void prof_end(int tag, hipStream_t s) {
    if (!((g_prof_on >> tag) & 1)) return;
    ProfPool& p = g_pool[tag];
    (void)hipEventRecord(p.b[p.used], s);
    p.used++;
}
}  // namespace asr

extern "C" int asr_prof_enable(int on) {
    asr::g_prof_on = on ? (1 << ASR_PROF_TAGS) - 1 : 0;
    for (auto& p : asr::g_pool) p.used = 0;
    return ASR_OK;
}
// Events for the families in `mask` only (bit = tag): bench.py times its K steps with the roofline kernel's events alone -- every
// pair of events costs the stream a few microseconds (all families on: +0.10 ms per 7.2-ms step, measured) -- and collects the
// other families in extra steps outside the timed region.
extern "C" int asr_prof_enable_mask(unsigned mask) {
    asr::g_prof_on = (int)(mask & ((1u << ASR_PROF_TAGS) - 1));
    for (auto& p : asr::g_pool) p.used = 0;
    return ASR_OK;
}

// Host call (synchronises on the recorded events): total elapsed ms and launch count of `tag`.
extern "C" int asr_prof_read(int tag, double* total_ms, int* launches) {
    if (tag < 0 || tag >= ASR_PROF_TAGS || !total_ms || !launches) return ASR_EINVAL;
    asr::ProfPool& p = asr::g_pool[tag];
    double t = 0;
    for (size_t i = 0; i < p.used; ++i) {
        float ms = 0;
        if (hipEventSynchronize(p.b[i]) != hipSuccess) return ASR_ELAUNCH;
        if (hipEventElapsedTime(&ms, p.a[i], p.b[i]) != hipSuccess) return ASR_ELAUNCH;
        t += ms;
    }
    *total_ms = t; *launches = (int)p.used;
    return ASR_OK;
}

// As asr_prof_read, one value per occurrence (in recording order): out[i] = elapsed ms of occurrence i, *n = how many were
// written (at most cap).  ASR_PROF_SIDE_TAIL values may be negative: the side stream was done before the caller's stream
// reached the join.
extern "C" int asr_prof_read_each(int tag, double* out, int cap, int* n) {
    if (tag < 0 || tag >= ASR_PROF_TAGS || !out || !n || cap < 0) return ASR_EINVAL;
    asr::ProfPool& p = asr::g_pool[tag];
    int k = 0;
    for (size_t i = 0; i < p.used && k < cap; ++i, ++k) {
        float ms = 0;
        if (hipEventSynchronize(p.a[i]) != hipSuccess || hipEventSynchronize(p.b[i]) != hipSuccess) return ASR_ELAUNCH;
        if (hipEventElapsedTime(&ms, p.a[i], p.b[i]) != hipSuccess) return ASR_ELAUNCH;
        out[k] = ms;
    }
    *n = k;
    return ASR_OK;
}

// ---------------------------------------------------------------------------------------------
// Co-residency budget of the persistent kernels.  They exchange data between workgroups of one launch without a grid
// barrier, which is only safe when all those workgroups are resident at once: one 512-thread workgroup per compute unit
// is guaranteed (each fits a CU's registers and LDS alone), so the budget is the CU count the runtime reports for the
// current device -- 256 on an MI355X, fewer on a partitioned (CPX/DPX) or otherwise reduced device.
// ASSUMPTION: the device is unmasked and owned by this process alone.  The attribute does not shrink under a CU mask
// (HSA_CU_MASK / ROC_GLOBAL_CU_MASK) and knows nothing of another process's or stream's resident kernels; on such a device
// set ASR_LSTM_MAXWG to the CUs really available, otherwise the only signal left is the 2-second exchange time-out.  A CU
// mask found in the environment is refused outright below rather than trusted.  (Grid padding to whole octets of groups
// adds workgroups that return at once, so a padded grid equal to the budget needs no more residency than the unpadded one.)
// ---------------------------------------------------------------------------------------------
namespace asr {
int resident_wg_budget() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 1;
        cus[dev] = n;
    }
    int budget = cus[dev];
    const char* maxwg = getenv("ASR_LSTM_MAXWG");
    if (maxwg) { const int x = atoi(maxwg); if (x > 0 && x < budget) budget = x; }
    else if (getenv("HSA_CU_MASK") || getenv("ROC_GLOBAL_CU_MASK")) {
        // a CU mask the attribute does not reflect: without an explicit ASR_LSTM_MAXWG no persistent grid is known to fit
        static bool warned = false;
        if (!warned) { warned = true; fprintf(stderr, "e2e_asr_hip: a CU mask is set; give ASR_LSTM_MAXWG=<CUs available> "
                                                      "(persistent kernels are refused until then)\n"); }
        budget = 0;
    }
    return budget;
}
}  // namespace asr
extern "C" int asr_resident_wg_budget(void) { return asr::resident_wg_budget(); }

// ---------------------------------------------------------------------------------------------
// Side stream for the decoder's LM chain (decoder.hip / decoder_bwd.hip): created once, lazily.
// Events are pooled and reused; record/wait pairs are legal inside hipGraph capture (fork/join).
// ---------------------------------------------------------------------------------------------
namespace asr {
static hipStream_t g_side = nullptr;
static std::vector<hipEvent_t> g_events;
static size_t g_ev_next = 0;
static hipEvent_t g_join = nullptr;      // last event recorded on the side stream, not yet joined

hipStream_t side_stream() {
    if (!g_side) {
        // ASR_SIDE_PRIO: 1 = lowest (default), -1 = highest, 0 = the default priority.  Streams of one priority share a small
        // pool of hardware queues: with RCCL's streams in the process (every multi-GPU run) a default-priority side stream
        // landed on the MAIN stream's queue and nothing overlapped (train step 12.9 instead of 12.1 ms, measured under
        // torchrun with ASR_FORCE_DIST=1); a stream of another priority gets a queue of its own.  Lowest, because the side
        // stream carries throughput work (weight-gradient GEMMs) next to the latency-bound recurrences of the main stream.
        int lo = 0, hi = 0;
        const char* e = getenv("ASR_SIDE_PRIO");
        const int want = e ? atoi(e) : 1;
        if (want != 0 && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
            (void)hipStreamCreateWithPriority(&g_side, hipStreamNonBlocking, want < 0 ? hi : lo);
        if (!g_side) (void)hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking);
    }
    return g_side;
}
hipEvent_t next_event() {
    if (g_ev_next == g_events.size()) {
        hipEvent_t e;
        (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        g_events.push_back(e);
    }
    hipEvent_t e = g_events[g_ev_next++];
    if (g_ev_next >= 4096) g_ev_next = 0;        // ring: far more than one step's worth
    return e;
}
void set_pending_join(hipEvent_t e) { g_join = e; }
}  // namespace asr

// Make `stream` wait for everything the library has queued on its side stream (LM-chain work of
// asr_attn_decoder_bwd).  Must be called before the gradients are consumed.
extern "C" int asr_side_join(void* stream) {
    if (asr::g_join) {
        if (((asr::g_prof_on >> ASR_PROF_SIDE_TAIL) & 1) && asr::g_side) {      // bench.py's side_stream_tail_ms: how long the caller's stream waits here
            asr::prof_begin(ASR_PROF_SIDE_TAIL, static_cast<hipStream_t>(stream));
            asr::prof_end(ASR_PROF_SIDE_TAIL, asr::g_side);
        }
        if (hipStreamWaitEvent(static_cast<hipStream_t>(stream), asr::g_join, 0) != hipSuccess) return ASR_ELAUNCH;
        asr::g_join = nullptr;
    }
    return ASR_OK;
}
// As asr_side_join for a THIRD stream (e.g. the stream a gradient exchange is launched from): `stream` waits for what the
// side stream holds so far; the pending join stays pending, so the caller's own asr_side_join still orders everything.
extern "C" int asr_side_wait(void* stream) {
    if (asr::g_join && hipStreamWaitEvent(static_cast<hipStream_t>(stream), asr::g_join, 0) != hipSuccess) return ASR_ELAUNCH;
    return ASR_OK;
}
