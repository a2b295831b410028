// gs_api.hip — status strings, version and the thread-local HIP error text of libgsplat_hip.so.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "gs_device.h"

namespace gs {
static thread_local char g_hip_err[256] = "";
void set_hip_error(hipError_t e, const char *what) {
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
}

// ROCTX ranges around the entry points of the path (SURVEY.md §5 tracing; `rocprofv3 --marker-trace`
// shows them above the kernels).  Off unless GSPLAT_ROCTX is set in the environment; libroctx64 is
// looked up in the process at first use (libtorch-ROCm brings it; otherwise dlopen) — the library
// has no link-time dependency on it.
namespace {
using push_fn = int (*)(const char *);
using pop_fn = int (*)();
push_fn g_push = nullptr;
pop_fn g_pop = nullptr;
bool roctx_ready() {
    static const bool ok = [] {
        if (!getenv("GSPLAT_ROCTX")) return false;
        void *h = dlopen(nullptr, RTLD_NOW);
        g_push = h ? reinterpret_cast<push_fn>(dlsym(h, "roctxRangePushA")) : nullptr;
        g_pop = h ? reinterpret_cast<pop_fn>(dlsym(h, "roctxRangePop")) : nullptr;
        if (!g_push || !g_pop) {
            void *l = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
            if (!l) l = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
            g_push = l ? reinterpret_cast<push_fn>(dlsym(l, "roctxRangePushA")) : nullptr;
            g_pop = l ? reinterpret_cast<pop_fn>(dlsym(l, "roctxRangePop")) : nullptr;
        }
        return g_push && g_pop;
    }();
    return ok;
}
}  // namespace
TraceRange::TraceRange(const char *name) : on_(roctx_ready()) { if (on_) g_push(name); }
TraceRange::~TraceRange() { if (on_) g_pop(); }
}  // namespace gs

// ---- kernel timeline (GS_LAUNCH, gs_device.h) -------------------------------------------------------
namespace gs {
namespace {
struct TimelineEntry {
    const char *name;
    hipEvent_t a, b;
};
struct Timeline {
    bool armed = false;
    std::vector<TimelineEntry> entries;
    std::vector<hipEvent_t> pool;
    hipEvent_t pending = nullptr;
    hipEvent_t take() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return e;
    }
    void clear() {
        for (auto &e : entries) { if (e.a) pool.push_back(e.a); if (e.b) pool.push_back(e.b); }
        entries.clear();
        if (pending) { pool.push_back(pending); pending = nullptr; }
    }
};
thread_local Timeline g_timeline;
}  // namespace
void timeline_before(hipStream_t s) {
    Timeline &t = g_timeline;
    if (!t.armed) return;
    if (!t.pending) t.pending = t.take();
    if (t.pending) (void)hipEventRecord(t.pending, s);
}
void timeline_after(const char *name, hipStream_t s) {
    Timeline &t = g_timeline;
    if (!t.armed || !t.pending) return;
    hipEvent_t b = t.take();
    if (!b) return;
    (void)hipEventRecord(b, s);
    t.entries.push_back({name, t.pending, b});
    t.pending = nullptr;
}
}  // namespace gs

extern "C" int gs_debug_timeline(int enable) {
    gs::g_timeline.clear();
    gs::g_timeline.armed = enable != 0;
    return GS_OK;
}

extern "C" int gs_debug_timeline_read(int capacity, char *names, int name_bytes, float *ms, int *count) {
    if (capacity < 0 || name_bytes < 8 || !count || (capacity > 0 && (!names || !ms))) return GS_ERR_INVALID_ARGUMENT;
    gs::Timeline &t = gs::g_timeline;
    const int n = (int)t.entries.size();
    *count = n;
    for (int i = 0; i < n && i < capacity; i++) {
        GS_HIP_CHECK(hipEventSynchronize(t.entries[i].b));
        float v = 0.0f;
        GS_HIP_CHECK(hipEventElapsedTime(&v, t.entries[i].a, t.entries[i].b));
        ms[i] = v;
        snprintf(names + (size_t)i * name_bytes, (size_t)name_bytes, "%s", t.entries[i].name);
    }
    return GS_OK;
}

extern "C" const char *gs_strerror(int status) {
    switch (status) {
    case GS_OK: return "ok";
    case GS_ERR_INVALID_ARGUMENT: return "invalid argument";
    case GS_ERR_UNSUPPORTED: return "unsupported configuration (image side > 65535 px)";
    case GS_ERR_WORKSPACE: return "workspace too small";
    case GS_ERR_HIP: return "HIP runtime error (see gs_last_hip_error)";
    case GS_ERR_CAPACITY: return "more tile intersections than the id buffer's capacity";
    default: return "unknown status";
    }
}

extern "C" const char *gs_last_hip_error(void) { return gs::g_hip_err; }

extern "C" int gs_version(void) { return GS_ABI_VERSION; }
