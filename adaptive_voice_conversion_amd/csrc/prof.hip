// Optional per-kernel-class timing with HIP events on the launch stream
// (used by bench.py to attribute step time and to compute the roofline figures
// from ALGORITHMIC flops/bytes; off by default, never active inside the timed
// region of the throughput measurement).
#include <hip/hip_runtime.h>

#include <vector>

#include "avc_common.h"
#include "avc_internal.h"

namespace {
struct Rec {
    int cls;
    double flops, bytes;
    hipEvent_t e0, e1;
};
bool g_on = false;
std::vector<Rec> g_recs;
}  // namespace

bool avc_prof_on() { return g_on; }

ProfScope::ProfScope(int cls, double flops, double bytes, hipStream_t s) : active_(g_on), s_(s) {
    if (!active_) return;
    Rec r;
    r.cls = cls;
    r.flops = flops;
    r.bytes = bytes;
    hipEventCreate(&r.e0);
    hipEventCreate(&r.e1);
    hipEventRecord(r.e0, s);
    g_recs.push_back(r);
}
ProfScope::~ProfScope() {
    if (active_) hipEventRecord(g_recs.back().e1, s_);
}

extern "C" {
int avc_prof_begin(void) {
    g_recs.clear();
    g_on = true;
    return 0;
}
// out arrays have AVC_K_NCLASS entries: total ms, launches, algorithmic flops, algorithmic bytes
int avc_prof_end(double* ms, long* launches, double* flops, double* bytes) {
    g_on = false;
    for (int i = 0; i < AVC_K_NCLASS; ++i) {
        ms[i] = 0;
        launches[i] = 0;
        flops[i] = 0;
        bytes[i] = 0;
    }
    for (auto& r : g_recs) {
        hipEventSynchronize(r.e1);
        float t = 0.f;
        hipEventElapsedTime(&t, r.e0, r.e1);
        ms[r.cls] += t;
        launches[r.cls] += 1;
        flops[r.cls] += r.flops;
        bytes[r.cls] += r.bytes;
        hipEventDestroy(r.e0);
        hipEventDestroy(r.e1);
    }
    g_recs.clear();
    return 0;
}
const char* avc_prof_class_name(int i) {
    static const char* names[AVC_K_NCLASS] = {"conv_fwd", "conv_dgrad", "conv_wgrad", "slab_reduce", "instnorm_fwd",
                                              "instnorm_bwd", "pack_weights", "clip_adam", "misc"};
    return (i >= 0 && i < AVC_K_NCLASS) ? names[i] : "?";
}
int avc_prof_nclass(void) { return AVC_K_NCLASS; }
}
