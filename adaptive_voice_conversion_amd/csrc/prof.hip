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
    hipStream_t s;
};
bool g_on = false;
std::vector<Rec> g_recs;
}  // namespace

bool avc_prof_on() { return g_on; }

// A handful of named points of a step, each ONE timing event on the stream it is reached on: unlike the per-launch brackets (and unlike
// a tracer) five events do not change the schedule they observe.
namespace {
bool g_marks = false;
hipEvent_t g_mark[AVC_PROF_NMARK];
bool g_mark_set[AVC_PROF_NMARK];
}  // namespace
void avc_prof_mark(int id, hipStream_t s) {
    if (!g_marks || id < 0 || id >= AVC_PROF_NMARK) return;
    if (!g_mark_set[id]) hipEventCreate(&g_mark[id]);
    g_mark_set[id] = true;
    hipEventRecord(g_mark[id], s);
}

ProfScope::ProfScope(int cls, double flops, double bytes, hipStream_t s) : active_(g_on), s_(s) {
    if (!active_) return;
    Rec r;
    r.cls = cls;
    r.flops = flops;
    r.bytes = bytes;
    hipEventCreate(&r.e0);
    hipEventCreate(&r.e1);
    hipEventRecord(r.e0, s);
    r.s = s;
    g_recs.push_back(r);
}
ProfScope::~ProfScope() {
    if (active_) hipEventRecord(g_recs.back().e1, s_);
}

extern "C" {
int avc_prof_marks_begin(void) {
    for (int i = 0; i < AVC_PROF_NMARK; ++i) g_mark_set[i] = false;
    g_marks = true;
    return 0;
}
// ms[i] = time of mark i relative to mark 0 (NaN where the mark was not reached); ends the recording
int avc_prof_marks_end(double* ms) {
    g_marks = false;
    for (int i = 0; i < AVC_PROF_NMARK; ++i) {
        ms[i] = __builtin_nan("");
        if (!g_mark_set[i]) continue;
        hipEventSynchronize(g_mark[i]);
        if (g_mark_set[0]) {
            float t = 0.f;
            hipEventElapsedTime(&t, g_mark[0], g_mark[i]);
            ms[i] = t;
        }
    }
    for (int i = 0; i < AVC_PROF_NMARK; ++i)
        if (g_mark_set[i]) hipEventDestroy(g_mark[i]);
    return AVC_PROF_NMARK;
}
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
// Diagnostic twin of avc_prof_end for a MULTI-stream step: every bracketed launch as (class, stream handle, start, end) in ms from the
// first bracket's start -- a per-stream timeline taken with HIP events instead of a tracer (rocprofv3's kernel trace stretches the step
// by ~10 %; the events cost ~2 us per launch).  Returns the number of records (at most `max` are written); ends the recording.
int avc_prof_timeline(int* cls, long* stream, double* t0_ms, double* t1_ms, int max) {
    g_on = false;
    int n = 0;
    for (auto& r : g_recs) hipEventSynchronize(r.e1);
    for (auto& r : g_recs) {
        if (n < max) {
            float a = 0.f, b = 0.f;
            hipEventElapsedTime(&a, g_recs[0].e0, r.e0);
            hipEventElapsedTime(&b, g_recs[0].e0, r.e1);
            cls[n] = r.cls;
            stream[n] = (long)(size_t)r.s;
            t0_ms[n] = a;
            t1_ms[n] = b;
        }
        ++n;
    }
    for (auto& r : g_recs) {
        hipEventDestroy(r.e0);
        hipEventDestroy(r.e1);
    }
    g_recs.clear();
    return n;
}
const char* avc_prof_class_name(int i) {
    static const char* names[AVC_K_NCLASS] = {"conv_fwd", "conv_dgrad", "conv_wgrad", "slab_reduce", "instnorm_fwd",
                                              "instnorm_bwd", "pack_weights", "clip_adam", "misc"};
    return (i >= 0 && i < AVC_K_NCLASS) ? names[i] : "?";
}
int avc_prof_nclass(void) { return AVC_K_NCLASS; }
}
