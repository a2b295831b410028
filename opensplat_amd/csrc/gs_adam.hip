// gs_adam.hip — Model::optimizersStep (model.cpp:236-243): the six torch::optim::Adam instances of
// model.cpp:61-66 as ONE launch over all parameter groups.  C ABI: include/gsplat_train.h.
//
// libtorch's Adam::step is ~10 element-wise kernels per optimiser (mul_, add_, mul_, addcmul_,
// sqrt, div, add_, addcdiv_ + temporaries), 60 launches and ~13 passes over each tensor per
// training step.  This is one pass: 16 B read + 12 B written per parameter (p, g, m, v -> p, m, v),
// 128-bit accesses, pure HBM streaming (59 parameters per Gaussian at SH degree 3: 1.65 GB at
// N = 1 M -> 0.26 ms at 6.3 TB/s).
//
// The arithmetic is ATen's, operation for operation (pinned against libtorch 2.10's CPU kernels by
// oracle/train_oracle.c, which this kernel is tested against):
//   m = fma(1 - b1, g, m * b1)            exp_avg.mul_(b1).add_(g, 1 - b1)
//   v = fma((1 - b2) * g, g, v * b2)      exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
//   denom = sqrt(v) / sqrt(1 - b2^t) + eps
//   p = p + (-(lr / (1 - b1^t)) * m) / denom      p.addcdiv_(m, denom, -step_size)
#include <math.h>

#include <algorithm>

#include "gs_device.h"
#include "../../include/gsplat_train.h"

namespace gs {

struct AdamArgs {
    float *p[GS_ADAM_MAX_GROUPS];
    const float *g[GS_ADAM_MAX_GROUPS];
    float *m[GS_ADAM_MAX_GROUPS];
    float *v[GS_ADAM_MAX_GROUPS];
    int64_t n[GS_ADAM_MAX_GROUPS];
    int64_t chunk_start[GS_ADAM_MAX_GROUPS + 1];  // in units of 4 floats
    float neg_step_size[GS_ADAM_MAX_GROUPS];
    uint32_t aligned;  // bit i: all four pointers of group i are 16-byte aligned
    int num;
    float beta1, beta2, omb1, omb2, bc2_sqrt, eps;
};

static __device__ __forceinline__ void adam1(float &p, float g, float &m, float &v, float nss,
                                             float bc2_sqrt, const AdamArgs &a) {
    m = fmaf(a.omb1, g, m * a.beta1);
    v = fmaf(a.omb2 * g, g, v * a.beta2);
    const float denom = sqrtf(v) / bc2_sqrt + a.eps;
    p = p + (nss * m) / denom;
}

// `row` (nullable): the per-step scalars {bc2_sqrt, neg_step_size[groups]} in device memory; NULL: those of
// the argument block.  (The block itself is never written: a kernel that modifies its by-value argument
// gets a private copy of all of it — 464 bytes of scratch per lane here.)
static __device__ __forceinline__ void adam_chunk(const AdamArgs &a, const float *__restrict__ row) {
    const int64_t chunk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (chunk >= a.chunk_start[a.num]) return;
    int grp = 0;
#pragma unroll
    for (int i = 1; i < GS_ADAM_MAX_GROUPS; i++)
        if (i < a.num && chunk >= a.chunk_start[i]) grp = i;
    const int64_t i0 = 4 * (chunk - a.chunk_start[grp]);
    const int64_t n = a.n[grp];
    float *p = a.p[grp] + i0, *m = a.m[grp] + i0, *v = a.v[grp] + i0;
    const float *g = a.g[grp] + i0;
    const float nss = row ? row[1 + grp] : a.neg_step_size[grp];
    const float bc2 = row ? row[0] : a.bc2_sqrt;
    if (i0 + 4 <= n && ((a.aligned >> grp) & 1u)) {
        // streaming: every byte is touched once per step and the working set (1.65 GB at N = 1 M)
        // is far beyond any cache -> non-temporal loads and stores
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 P = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
        const f4 G = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(g));
        f4 M = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(m));
        f4 V = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(v));
        float pe[4] = {P.x, P.y, P.z, P.w}, me[4] = {M.x, M.y, M.z, M.w}, ve[4] = {V.x, V.y, V.z, V.w};
        const float ge[4] = {G.x, G.y, G.z, G.w};
#pragma unroll
        for (int k = 0; k < 4; k++) adam1(pe[k], ge[k], me[k], ve[k], nss, bc2, a);
        P.x = pe[0]; P.y = pe[1]; P.z = pe[2]; P.w = pe[3];
        M.x = me[0]; M.y = me[1]; M.z = me[2]; M.w = me[3];
        V.x = ve[0]; V.y = ve[1]; V.z = ve[2]; V.w = ve[3];
        __builtin_nontemporal_store(P, reinterpret_cast<f4 *>(p));
        __builtin_nontemporal_store(M, reinterpret_cast<f4 *>(m));
        __builtin_nontemporal_store(V, reinterpret_cast<f4 *>(v));
    } else {
        for (int k = 0; k < 4 && i0 + k < n; k++) {
            float P = p[k], M = m[k], V = v[k];
            adam1(P, g[k], M, V, nss, bc2, a);
            p[k] = P;
            m[k] = M;
            v[k] = V;
        }
    }
}

__global__ void __launch_bounds__(256) k_adam(AdamArgs a) { adam_chunk(a, nullptr); }

// The same update with the per-step scalars read from DEVICE memory (gs_adam_step_scheduled): row
// *row_index of `rows` = {sqrt(1 - beta2^step), -(lr_g / (1 - beta1^step)) for every group}, written by the
// host with gs_adam_schedule_row — so that a captured HIP graph replays the launch unchanged while the step
// count advances.  Nothing happens when the guard says the step is void (*guard > guard_max: a speculative
// id list upstream was too small, the gradients are not to be used) or the row index is outside the table.
__global__ void __launch_bounds__(256)
k_adam_scheduled(AdamArgs a, const float *__restrict__ rows, const int32_t *__restrict__ row_index,
                 int num_rows, const int32_t *__restrict__ guard, int guard_max) {
    if (guard && *guard > guard_max) return;
    const int r = *row_index;
    if (r < 0 || r >= num_rows) return;
    adam_chunk(a, rows + (size_t)r * GS_ADAM_ROW_FLOATS);
}

__global__ void k_adam_advance(int32_t *__restrict__ row_index, const int32_t *__restrict__ guard,
                               int guard_max) {
    if (guard && *guard > guard_max) return;
    *row_index += 1;
}

// host side of both entry points: everything of AdamArgs that does not depend on the step
static int fill_adam_args(AdamArgs &a, int num_groups, const GsAdamGroup *groups, double beta1, double beta2,
                          double eps, int64_t &chunks) {
    a.num = num_groups;
    a.beta1 = (float)beta1;
    a.beta2 = (float)beta2;
    a.omb1 = (float)(1.0 - beta1);
    a.omb2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    chunks = 0;
    for (int i = 0; i < num_groups; i++) {
        const GsAdamGroup &gr = groups[i];
        if (gr.n < 0) return GS_ERR_INVALID_ARGUMENT;
        if (gr.n > 0 && (!gr.param || !gr.grad || !gr.exp_avg || !gr.exp_avg_sq))
            return GS_ERR_INVALID_ARGUMENT;
        a.p[i] = gr.param;
        a.g[i] = gr.grad;
        a.m[i] = gr.exp_avg;
        a.v[i] = gr.exp_avg_sq;
        a.n[i] = gr.n;
        const uintptr_t bits = (uintptr_t)gr.param | (uintptr_t)gr.grad | (uintptr_t)gr.exp_avg |
                               (uintptr_t)gr.exp_avg_sq;
        if ((bits & 15u) == 0) a.aligned |= 1u << i;
        a.chunk_start[i] = chunks;
        chunks += (gr.n + 3) / 4;
    }
    for (int i = num_groups; i <= GS_ADAM_MAX_GROUPS; i++) a.chunk_start[i] = chunks;
    return GS_OK;
}

}  // namespace gs

namespace gs {
// Staging inside a captured graph: what changes from replay to replay is written by the HOST into pinned,
// device-mapped memory before the launch and fetched by the first nodes of the graph itself — no stream copy
// stands between the host and the replay.
__global__ void __launch_bounds__(64) k_stage_words(uint32_t *__restrict__ dst,
                                                    const uint32_t *__restrict__ src_host, int n) {
    for (int i = threadIdx.x; i < n; i += 64) dst[i] = src_host[i];
}
__global__ void __launch_bounds__(256) k_copy_indirect(float *__restrict__ dst,
                                                       const float *const *__restrict__ src_ptr_host,
                                                       int64_t n) {
    const float *__restrict__ src = *src_ptr_host;   // (uniform: one scalar load of the pinned word)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15u) == 0) {
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        for (int64_t i = t; i < n / 4; i += stride) d4[i] = s4[i];
        for (int64_t i = (n / 4) * 4 + t; i < n; i += stride) dst[i] = src[i];
    } else {
        for (int64_t i = t; i < n; i += stride) dst[i] = src[i];
    }
}
}  // namespace gs

extern "C" int gs_stage_f32(float *dst_dev, const float *src_pinned_host, int count, gs_stream_t stream) {
    if (count < 0 || count > 4096) return GS_ERR_INVALID_ARGUMENT;
    if (count == 0) return GS_OK;
    if (!dst_dev || !src_pinned_host) return GS_ERR_INVALID_ARGUMENT;
    GS_LAUNCH(gs::k_stage_words, dim3(1), dim3(64), 0, (hipStream_t)stream,
              reinterpret_cast<uint32_t *>(dst_dev), reinterpret_cast<const uint32_t *>(src_pinned_host), count);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" int gs_copy_indirect_f32(float *dst_dev, const float *const *src_ptr_pinned_host, int64_t count,
                                    gs_stream_t stream) {
    if (count < 0) return GS_ERR_INVALID_ARGUMENT;
    if (count == 0) return GS_OK;
    if (!dst_dev || !src_ptr_pinned_host) return GS_ERR_INVALID_ARGUMENT;
    const int64_t blocks = std::min<int64_t>((count / 4 + 255) / 256 + 1, 2048);
    GS_LAUNCH(gs::k_copy_indirect, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dst_dev,
              src_ptr_pinned_host, count);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" int gs_adam_schedule_row(int num_groups, const double *lrs, int64_t step, double beta1,
                                    double beta2, float *row) {
    if (num_groups < 0 || num_groups > GS_ADAM_MAX_GROUPS || step < 1 || !row || (num_groups && !lrs))
        return GS_ERR_INVALID_ARGUMENT;
    // (the expressions of gs_adam_step, so that both forms hand the kernel the same bits)
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    row[0] = (float)sqrt(bc2);
    for (int i = 0; i < GS_ADAM_MAX_GROUPS; i++) row[1 + i] = i < num_groups ? (float)(-(lrs[i] / bc1)) : 0.0f;
    return GS_OK;
}

extern "C" int gs_adam_step_scheduled(int num_groups, const GsAdamGroup *groups, const float *rows_dev,
                                      const int32_t *row_index_dev, int32_t num_rows,
                                      const int32_t *guard_dev, int32_t guard_max, double beta1,
                                      double beta2, double eps, gs_stream_t stream) {
    using namespace gs;
    if (num_groups < 0 || num_groups > GS_ADAM_MAX_GROUPS || num_rows < 1) return GS_ERR_INVALID_ARGUMENT;
    if (!rows_dev || !row_index_dev) return GS_ERR_INVALID_ARGUMENT;
    if (num_groups == 0) return GS_OK;
    if (!groups) return GS_ERR_INVALID_ARGUMENT;
    AdamArgs a = {};
    int64_t chunks = 0;
    const int rc = fill_adam_args(a, num_groups, groups, beta1, beta2, eps, chunks);
    if (rc != GS_OK) return rc;
    if (chunks == 0) return GS_OK;
    const int64_t blocks = (chunks + 255) / 256;
    if (blocks > 0x7fffffffLL) return GS_ERR_UNSUPPORTED;
    GS_LAUNCH(k_adam_scheduled, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, rows_dev,
              row_index_dev, (int)num_rows, guard_dev, (int)guard_max);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" int gs_adam_advance(int32_t *row_index_dev, const int32_t *guard_dev, int32_t guard_max,
                               gs_stream_t stream) {
    if (!row_index_dev) return GS_ERR_INVALID_ARGUMENT;
    GS_LAUNCH(gs::k_adam_advance, dim3(1), dim3(1), 0, (hipStream_t)stream, row_index_dev, guard_dev,
              (int)guard_max);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" int gs_adam_step(int num_groups, const GsAdamGroup *groups, int64_t step, double beta1,
                            double beta2, double eps, gs_stream_t stream) {
    using namespace gs;
    if (num_groups < 0 || num_groups > GS_ADAM_MAX_GROUPS || step < 1) return GS_ERR_INVALID_ARGUMENT;
    if (num_groups == 0) return GS_OK;
    if (!groups) return GS_ERR_INVALID_ARGUMENT;
    AdamArgs a = {};
    // libtorch computes the bias corrections in double and casts scalars to the tensors' float
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    int64_t chunks = 0;
    const int rc = fill_adam_args(a, num_groups, groups, beta1, beta2, eps, chunks);
    if (rc != GS_OK) return rc;
    a.bc2_sqrt = (float)sqrt(bc2);
    for (int i = 0; i < num_groups; i++) a.neg_step_size[i] = (float)(-(groups[i].lr / bc1));
    if (chunks == 0) return GS_OK;
    const int64_t blocks = (chunks + 255) / 256;
    if (blocks > 0x7fffffffLL) return GS_ERR_UNSUPPORTED;
    GS_LAUNCH(k_adam, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" float gs_sched_lr(float lr_init, float lr_final, int max_steps, int step) {
    float t = (float)step / (float)max_steps;  // optim_scheduler.cpp:5
    t = t < 1.0f ? t : 1.0f;
    t = t > 0.0f ? t : 0.0f;
    return expf(logf(lr_init) * (1.0f - t) + logf(lr_final) * t);
}
