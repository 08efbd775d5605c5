// Optional HIP-event timing of individual kernels on the stream they are launched on
// (bench.py's roofline leg).  Disabled by default: zero overhead on the product path.
#include "common.h"
#include <vector>

namespace asr {
struct ProfPool { std::vector<hipEvent_t> a, b; size_t used = 0; };
static ProfPool g_pool[ASR_PROF_TAGS];
int g_prof_on = 0;

void prof_begin(int tag, hipStream_t s) {
    if (!g_prof_on) return;
    ProfPool& p = g_pool[tag];
    if (p.used == p.a.size()) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        p.a.push_back(e0); p.b.push_back(e1);
    }
    (void)hipEventRecord(p.a[p.used], s);
}
void prof_end(int tag, hipStream_t s) {
    if (!g_prof_on) return;
    ProfPool& p = g_pool[tag];
    (void)hipEventRecord(p.b[p.used], s);
    p.used++;
}
}  // namespace asr

extern "C" int asr_prof_enable(int on) {
    asr::g_prof_on = on;
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
