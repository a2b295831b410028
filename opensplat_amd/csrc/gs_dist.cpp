// gs_dist.cpp — libgsplat_dist.so: the gradient exchange of the camera-per-rank path on RCCL
// (include/gsplat_dist.h).  Kept out of libgsplat_hip.so so that the rasterizer library itself has
// no communication dependency; links the RCCL that ships with libtorch-ROCm (one RCCL per process).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/gsplat_dist.h"

struct GsDistComm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
    // GSPLAT_DIST_FORCE_COLLECTIVES=1 (read at gs_dist_init): a one-rank communicator really enqueues
    // ncclAllReduce / ncclAllGather instead of taking the world == 1 shortcuts — so that RCCL's
    // kernels run at least once on a one-GPU test box (tests/test_gpu_dist_cabi.py)
    bool force = false;
};

namespace {
thread_local char g_err[256] = "";
int fail_nccl(ncclResult_t r, const char *what) {
    std::snprintf(g_err, sizeof(g_err), "%s: %s", what, ncclGetErrorString(r));
    return GS_ERR_HIP;
}
int fail_hip(hipError_t e, const char *what) {
    std::snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return GS_ERR_HIP;
}
static_assert(sizeof(ncclUniqueId) <= GS_DIST_ID_BYTES, "ncclUniqueId does not fit GS_DIST_ID_BYTES");
}  // namespace

extern "C" const char *gs_dist_last_error(void) { return g_err; }

extern "C" int gs_dist_unique_id(uint8_t id[GS_DIST_ID_BYTES]) {
    if (!id) return GS_ERR_INVALID_ARGUMENT;
    ncclUniqueId u;
    const ncclResult_t r = ncclGetUniqueId(&u);
    if (r != ncclSuccess) return fail_nccl(r, "ncclGetUniqueId");
    std::memset(id, 0, GS_DIST_ID_BYTES);
    std::memcpy(id, &u, sizeof(u));
    return GS_OK;
}

extern "C" int gs_dist_init(GsDistComm **comm, int world_size, int rank, const uint8_t id[GS_DIST_ID_BYTES],
                            int device) {
    if (!comm || !id || world_size < 1 || rank < 0 || rank >= world_size || device < 0)
        return GS_ERR_INVALID_ARGUMENT;
    *comm = nullptr;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
    GsDistComm *c = new (std::nothrow) GsDistComm;
    if (!c) return GS_ERR_INVALID_ARGUMENT;
    c->world = world_size;
    c->rank = rank;
    c->device = device;
    const char *f = std::getenv("GSPLAT_DIST_FORCE_COLLECTIVES");
    c->force = f && f[0] && f[0] != '0';
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    const ncclResult_t r = ncclCommInitRank(&c->comm, world_size, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail_nccl(r, "ncclCommInitRank");
    }
    *comm = c;
    return GS_OK;
}

extern "C" int gs_dist_allreduce_sum(GsDistComm *comm, float *buf, size_t count, gs_stream_t stream) {
    if (!comm || (!buf && count)) return GS_ERR_INVALID_ARGUMENT;
    if (count == 0 || (comm->world == 1 && !comm->force)) return GS_OK;
    const ncclResult_t r = ncclAllReduce(buf, buf, count, ncclFloat32, ncclSum, comm->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return fail_nccl(r, "ncclAllReduce");
    return GS_OK;
}

extern "C" int gs_dist_allreduce_sum_buckets(GsDistComm *comm, float *buf, size_t count, int n_buckets,
                                             void **done_events, gs_stream_t stream) {
    if (!comm || (!buf && count) || n_buckets < 1) return GS_ERR_INVALID_ARGUMENT;
    size_t per = (count + (size_t)n_buckets - 1) / (size_t)n_buckets;
    per = (per + 1023) / 1024 * 1024;
    if (per == 0) per = 1024;
    for (int k = 0; k < n_buckets; k++) {
        const size_t lo = (size_t)k * per;
        if (lo < count) {
            const size_t n = count - lo < per ? count - lo : per;
            const int rc = gs_dist_allreduce_sum(comm, buf + lo, n, stream);
            if (rc != GS_OK) return rc;
        }
        if (done_events && done_events[k]) {   // (behind an empty bucket too: consumers wait on all)
            const hipError_t e = hipEventRecord((hipEvent_t)done_events[k], (hipStream_t)stream);
            if (e != hipSuccess) return fail_hip(e, "hipEventRecord");
        }
    }
    return GS_OK;
}

extern "C" int gs_dist_allgather(GsDistComm *comm, const float *send, float *recv, size_t count,
                                 gs_stream_t stream) {
    if (!comm || ((!send || !recv) && count)) return GS_ERR_INVALID_ARGUMENT;
    if (count == 0) return GS_OK;
    if (comm->world == 1 && !comm->force) {
        if (send == recv) return GS_OK;
        const hipError_t e = hipMemcpyAsync(recv, send, count * sizeof(float), hipMemcpyDeviceToDevice,
                                            (hipStream_t)stream);
        return e == hipSuccess ? GS_OK : fail_hip(e, "hipMemcpyAsync");
    }
    const ncclResult_t r = ncclAllGather(send, recv, count, ncclFloat32, comm->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return fail_nccl(r, "ncclAllGather");
    return GS_OK;
}

extern "C" int gs_dist_world_size(const GsDistComm *comm) { return comm ? comm->world : 0; }
extern "C" int gs_dist_rank(const GsDistComm *comm) { return comm ? comm->rank : -1; }

extern "C" int gs_dist_destroy(GsDistComm *comm) {
    if (!comm) return GS_OK;
    ncclResult_t r = ncclSuccess;
    if (comm->comm) r = ncclCommDestroy(comm->comm);
    delete comm;
    return r == ncclSuccess ? GS_OK : fail_nccl(r, "ncclCommDestroy");
}
