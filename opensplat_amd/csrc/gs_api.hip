// gs_api.hip — status strings, version and the thread-local HIP error text of libgsplat_hip.so.
#include <stdio.h>

#include "gs_device.h"

namespace gs {
static thread_local char g_hip_err[256] = "";
void set_hip_error(hipError_t e, const char *what) {
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
}
}  // namespace gs

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

extern "C" int gs_version(void) { return 100; /* 0.1.0 */ }
