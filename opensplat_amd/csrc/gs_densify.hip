// gs_densify.hip — Model::afterTrain (model.cpp:311-494) on the device: per-iteration statistics,
// the split / duplicate / cull decisions of a refinement step, and the surgery on the six
// parameter tensors and their twelve Adam moment tensors.  C ABI: include/gsplat_densify.h.
//
// The reference does this with ~150 torch ops per refinement (boolean-mask index / index_put, cat,
// repeat, where, ...) and several .item() host syncs, and ~10 masked ops with two syncs EVERY
// iteration for the statistics.  Here: one streaming kernel per iteration; per refinement
//   k_densify_flags   decisions per Gaussian + per-1024-block counts
//   k_densify_scan    exclusive scan of the block counts (one workgroup), totals to pinned memory
//   k_densify_map     destination row of every survivor -> source maps of the new set
//   k_gather_rows     one row gather per tensor (parameters and moments; new rows' moments = 0)
//   k_split_fixup     means / scales of the split samples
// All of it is HBM-streaming integer / copy work; nothing here is GEMM-shaped.
#include <math.h>

#include "gs_device.h"
#include "../../include/gsplat_densify.h"

namespace gs {

constexpr int kDBlock = 1024;
constexpr int kNC = 5;  // counters: splits, kept originals, kept split sources, kept dups, dups
enum : uint8_t { F_SPLIT = 1, F_DUP = 2, F_KEEP_O = 4, F_KEEP_S = 8, F_KEEP_D = 16 };

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_densify_stats(int N, const float2 *__restrict__ xys_grad, const int32_t *__restrict__ radii,
                float max_side, int first, float *__restrict__ gnorm, float *__restrict__ vis,
                float *__restrict__ m2d) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float2 g = xys_grad[n];
    const float norm = sqrtf(g.x * g.x + g.y * g.y);  // linalg_vector_norm(xys.grad(), 2, -1)
    const int r = radii[n];
    const bool visible = r > 0;
    float a, c, m;
    if (first) {  // model.cpp:321-323,329-331: for every Gaussian, visible or not
        a = norm;
        c = 1.0f;
        m = 0.0f;
    } else {
        a = gnorm[n];
        c = vis[n];
        m = m2d[n];
        if (visible) {  // model.cpp:325-326
            c = c + 1.0f;
            a = norm + a;
        }
    }
    if (visible) m = fmaxf(m, (float)r / max_side);  // model.cpp:333-336
    gnorm[n] = a;
    vis[n] = c;
    m2d[n] = m;
}

// ---------------------------------------------------------------------------------------------
struct DensifyLayout {
    int nb;
    size_t flags_off, sums_off, offs_off, totals_off, map_p_off, map_m_off, sample_off, total;
    explicit DensifyLayout(int N) {
        nb = (N + kDBlock - 1) / kDBlock;
        size_t o = 0;
        auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
        flags_off = take((size_t)N);
        sums_off = take((size_t)nb * kNC * 4);
        offs_off = take((size_t)nb * kNC * 4);
        totals_off = take(8 * 4);
        const size_t worst = (size_t)N * (2 + GS_SPLIT_SAMPLES);  // originals + samples + dups
        map_p_off = take(worst * 4);
        map_m_off = take(worst * 4);
        sample_off = take(worst * 4);
        total = o;
    }
};

static __device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void __launch_bounds__(kDBlock)
k_densify_flags(int N, GsDensifyConfig cfg, const float *__restrict__ gnorm,
                const float *__restrict__ vis, const float *__restrict__ m2d,
                const float *__restrict__ log_scales, const float *__restrict__ logits,
                uint8_t *__restrict__ flags, int32_t *__restrict__ sums) {
    const int n = blockIdx.x * kDBlock + threadIdx.x;
    uint8_t f = 0;
    if (n < N) {
        const float avg = (gnorm[n] / vis[n]) * cfg.half_max_side;
        const bool high = avg > cfg.densify_grad_thresh;
        const float e0 = expf(log_scales[3 * (size_t)n + 0]), e1 = expf(log_scales[3 * (size_t)n + 1]),
                    e2 = expf(log_scales[3 * (size_t)n + 2]);
        const float size = fmaxf(e0, fmaxf(e1, e2));
        const float m = m2d[n];
        bool split = size > cfg.densify_size_thresh;
        if (cfg.check_screen_size) split = split || (m > cfg.split_screen_size);
        split = split && high;
        const bool dup = (size <= cfg.densify_size_thresh) && high;
        const bool faint = sigmoidf(logits[n]) < cfg.cull_alpha_thresh;
        // the cull runs over cat({originals, split samples, dups}) with max2DSize = 0 for the new ones
        bool huge_o = false, huge_s = false, huge_d = false;
        if (cfg.cull_huge) {
            huge_o = size > cfg.cull_scale_thresh;
            if (cfg.check_screen_size) huge_o = huge_o || (m > cfg.cull_screen_size);
            huge_d = size > cfg.cull_scale_thresh;
            const float s0 = expf(logf(e0 / 1.6f)), s1 = expf(logf(e1 / 1.6f)), s2 = expf(logf(e2 / 1.6f));
            huge_s = fmaxf(s0, fmaxf(s1, s2)) > cfg.cull_scale_thresh;
        }
        if (split) f |= F_SPLIT;
        if (dup) f |= F_DUP;
        if (!(faint || split || huge_o)) f |= F_KEEP_O;
        if (split && !(faint || huge_s)) f |= F_KEEP_S;
        if (dup && !(faint || huge_d)) f |= F_KEEP_D;
        flags[n] = f;
    }
    const int c0 = __syncthreads_count(f & F_SPLIT), c1 = __syncthreads_count(f & F_KEEP_O),
              c2 = __syncthreads_count(f & F_KEEP_S), c3 = __syncthreads_count(f & F_KEEP_D),
              c4 = __syncthreads_count(f & F_DUP);
    if (threadIdx.x == 0) {
        int32_t *s = sums + (size_t)blockIdx.x * kNC;
        s[0] = c0; s[1] = c1; s[2] = c2; s[3] = c3; s[4] = c4;
    }
}

__global__ void __launch_bounds__(256)
k_densify_scan(int N, int nb, const int32_t *__restrict__ sums, int32_t *__restrict__ offs,
               int32_t *__restrict__ totals, int32_t *__restrict__ counts_host) {
    __shared__ int32_t chunk[256][kNC];
    const int t = threadIdx.x;
    const int per = (nb + 255) / 256, b0 = t * per, b1 = min(nb, b0 + per);
    int32_t acc[kNC] = {0, 0, 0, 0, 0};
    for (int b = b0; b < b1; b++)
        for (int k = 0; k < kNC; k++) acc[k] += sums[(size_t)b * kNC + k];
    for (int k = 0; k < kNC; k++) chunk[t][k] = acc[k];
    __syncthreads();
    if (t == 0) {
        int32_t run[kNC] = {0, 0, 0, 0, 0};
        for (int i = 0; i < 256; i++)
            for (int k = 0; k < kNC; k++) {
                const int32_t v = chunk[i][k];
                chunk[i][k] = run[k];
                run[k] += v;
            }
        const int32_t n_splits = run[0], kept_o = run[1], kept_s = run[2], kept_d = run[3],
                      n_dups = run[4];
        const int32_t new_n = kept_o + GS_SPLIT_SAMPLES * kept_s + kept_d;
        const int32_t added = GS_SPLIT_SAMPLES * n_splits + n_dups;
        const int32_t out[8] = {n_splits, n_dups, kept_o, kept_s, kept_d, new_n, added,
                                N + added - new_n};
        for (int k = 0; k < 8; k++) {
            totals[k] = out[k];
            if (counts_host) counts_host[k] = out[k];
        }
    }
    __syncthreads();
    for (int k = 0; k < kNC; k++) acc[k] = chunk[t][k];
    for (int b = b0; b < b1; b++)
        for (int k = 0; k < kNC; k++) {
            offs[(size_t)b * kNC + k] = acc[k];
            acc[k] += sums[(size_t)b * kNC + k];
        }
}

// Exclusive rank of this thread among the threads of the 1024-thread block with `pred`.
static __device__ __forceinline__ int block_rank(bool pred, int32_t *wave_sums /* [16] */) {
    const uint64_t b = __builtin_amdgcn_ballot_w64(pred);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int in_wave = __builtin_popcountll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wave_sums[wave] = __builtin_popcountll(b);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; w++) base += wave_sums[w];
    __syncthreads();
    return base + in_wave;
}

__global__ void __launch_bounds__(kDBlock)
k_densify_map(int N, const uint8_t *__restrict__ flags, const int32_t *__restrict__ offs,
              const int32_t *__restrict__ totals, int32_t *__restrict__ map_p,
              int32_t *__restrict__ map_m, int32_t *__restrict__ sample_row) {
    __shared__ int32_t ws[16];
    const int n = blockIdx.x * kDBlock + threadIdx.x;
    const uint8_t f = n < N ? flags[n] : 0;
    const int32_t *o = offs + (size_t)blockIdx.x * kNC;
    const int r_split = o[0] + block_rank(f & F_SPLIT, ws);
    const int r_o = o[1] + block_rank(f & F_KEEP_O, ws);
    const int r_s = o[2] + block_rank(f & F_KEEP_S, ws);
    const int r_d = o[3] + block_rank(f & F_KEEP_D, ws);
    const int n_splits = totals[GS_DENSIFY_N_SPLITS], kept_o = totals[GS_DENSIFY_KEPT_ORIG],
              kept_s = totals[GS_DENSIFY_KEPT_SPLIT];
    if (f & F_KEEP_O) {
        map_p[r_o] = n;
        map_m[r_o] = n;       // survivors keep their optimiser state (removeFromOptimizer)
        sample_row[r_o] = -1;
    }
    if (f & F_KEEP_S) {
#pragma unroll
        for (int j = 0; j < GS_SPLIT_SAMPLES; j++) {
            const int r = kept_o + j * kept_s + r_s;  // sample-major: repeat({nSplitSamples, 1})
            map_p[r] = n;
            map_m[r] = -1;                            // zeros_like(...) (addToOptimizer)
            sample_row[r] = j * n_splits + r_split;   // row of randn({2 * nSplits, 3})
        }
    }
    if (f & F_KEEP_D) {
        const int r = kept_o + GS_SPLIT_SAMPLES * kept_s + r_d;
        map_p[r] = n;
        map_m[r] = -1;
        sample_row[r] = -1;
    }
}

__global__ void __launch_bounds__(256)
k_gather_rows(int64_t total, int row_len, const float *__restrict__ src,
              const int32_t *__restrict__ map, float *__restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / row_len;
        const int c = (int)(i - r * row_len);
        const int32_t s = map[r];
        dst[i] = (s >= 0 && src) ? src[(size_t)s * row_len + c] : 0.0f;
    }
}

// means / scales of the split samples, model.cpp:360-365,372-373 and tensor_math.cpp:5-28
__global__ void __launch_bounds__(256)
k_split_fixup(int new_N, const int32_t *__restrict__ map_p, const int32_t *__restrict__ sample_row,
              const float *__restrict__ samples, const float *__restrict__ means,
              const float *__restrict__ log_scales, const float *__restrict__ quats,
              float *__restrict__ dst_means, float *__restrict__ dst_scales) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= new_N) return;
    const int sr = sample_row[r];
    if (sr < 0) return;
    const size_t n = (size_t)map_p[r];
    const float e0 = expf(log_scales[3 * n + 0]), e1 = expf(log_scales[3 * n + 1]),
                e2 = expf(log_scales[3 * n + 2]);
    // scaledSamples = exp(scales) * centeredSamples
    const float v0 = e0 * samples[3 * (size_t)sr + 0], v1 = e1 * samples[3 * (size_t)sr + 1],
                v2 = e2 * samples[3 * (size_t)sr + 2];
    // qs = quats / |quats|, then quatToRotMat normalises once more (F::normalize, eps 1e-12)
    float q0 = quats[4 * n + 0], q1 = quats[4 * n + 1], q2 = quats[4 * n + 2], q3 = quats[4 * n + 3];
    const float nrm = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
    q0 = q0 / nrm; q1 = q1 / nrm; q2 = q2 / nrm; q3 = q3 / nrm;
    const float nrm2 = fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);
    const float w = q0 / nrm2, x = q1 / nrm2, y = q2 / nrm2, z = q3 / nrm2;
    const float r00 = 1.0f - 2.0f * (y * y + z * z), r01 = 2.0f * (x * y - w * z),
                r02 = 2.0f * (x * z + w * y);
    const float r10 = 2.0f * (x * y + w * z), r11 = 1.0f - 2.0f * (x * x + z * z),
                r12 = 2.0f * (y * z - w * x);
    const float r20 = 2.0f * (x * z - w * y), r21 = 2.0f * (y * z + w * x),
                r22 = 1.0f - 2.0f * (x * x + y * y);
    dst_means[3 * (size_t)r + 0] = (r00 * v0 + r01 * v1 + r02 * v2) + means[3 * n + 0];
    dst_means[3 * (size_t)r + 1] = (r10 * v0 + r11 * v1 + r12 * v2) + means[3 * n + 1];
    dst_means[3 * (size_t)r + 2] = (r20 * v0 + r21 * v1 + r22 * v2) + means[3 * n + 2];
    dst_scales[3 * (size_t)r + 0] = logf(e0 / 1.6f);  // sizeFac, model.cpp:372-373
    dst_scales[3 * (size_t)r + 1] = logf(e1 / 1.6f);
    dst_scales[3 * (size_t)r + 2] = logf(e2 / 1.6f);
}

__global__ void __launch_bounds__(256)
k_reset_opacity(int N, float max_logit, float *__restrict__ logits, float *__restrict__ m,
                float *__restrict__ v) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    logits[n] = fminf(logits[n], max_logit);  // clamp_max
    if (m) m[n] = 0.0f;
    if (v) v[n] = 0.0f;
}

}  // namespace gs

extern "C" int gs_densify_stats(int N, const float *xys_grad, const int32_t *radii, float max_side,
                                int first, float *xys_grad_norm, float *vis_counts,
                                float *max_2d_size, gs_stream_t stream) {
    if (N < 0 || !(max_side > 0.0f)) return GS_ERR_INVALID_ARGUMENT;
    if (N == 0) return GS_OK;
    if (!xys_grad || !radii || !xys_grad_norm || !vis_counts || !max_2d_size)
        return GS_ERR_INVALID_ARGUMENT;
    GS_LAUNCH(gs::k_densify_stats, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       N, reinterpret_cast<const float2 *>(xys_grad), radii, max_side, first,
                       xys_grad_norm, vis_counts, max_2d_size);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" size_t gs_densify_workspace_bytes(int N) {
    if (N <= 0) return 0;
    return gs::DensifyLayout(N).total;
}

extern "C" int gs_densify_plan(int N, const GsDensifyConfig *cfg, const float *xys_grad_norm,
                               const float *vis_counts, const float *max_2d_size,
                               const float *log_scales, const float *opacity_logits,
                               int32_t *counts_host, void *workspace, size_t workspace_bytes,
                               gs_stream_t stream) {
    using namespace gs;
    if (N <= 0 || !cfg || !xys_grad_norm || !vis_counts || !max_2d_size || !log_scales ||
        !opacity_logits || !workspace)
        return GS_ERR_INVALID_ARGUMENT;
    if ((int64_t)N * (2 + GS_SPLIT_SAMPLES) > 0x7fffffffLL) return GS_ERR_UNSUPPORTED;
    const DensifyLayout L(N);
    if (workspace_bytes < L.total) return GS_ERR_WORKSPACE;
    char *ws = (char *)workspace;
    hipStream_t s = (hipStream_t)stream;
    GS_LAUNCH(k_densify_flags, dim3(L.nb), dim3(kDBlock), 0, s, N, *cfg, xys_grad_norm,
                       vis_counts, max_2d_size, log_scales, opacity_logits,
                       (uint8_t *)(ws + L.flags_off), (int32_t *)(ws + L.sums_off));
    GS_LAUNCH_CHECK();
    GS_LAUNCH(k_densify_scan, dim3(1), dim3(256), 0, s, N, L.nb,
                       (const int32_t *)(ws + L.sums_off), (int32_t *)(ws + L.offs_off),
                       (int32_t *)(ws + L.totals_off), counts_host);
    GS_LAUNCH_CHECK();
    GS_LAUNCH(k_densify_map, dim3(L.nb), dim3(kDBlock), 0, s, N,
                       (const uint8_t *)(ws + L.flags_off), (const int32_t *)(ws + L.offs_off),
                       (const int32_t *)(ws + L.totals_off), (int32_t *)(ws + L.map_p_off),
                       (int32_t *)(ws + L.map_m_off), (int32_t *)(ws + L.sample_off));
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" int gs_densify_apply(int N, int K, int new_N, const float *samples,
                                const GsGaussianSet *src, const GsGaussianSet *dst,
                                const void *workspace, size_t workspace_bytes, gs_stream_t stream) {
    using namespace gs;
    if (N <= 0 || K < 1 || new_N < 0 || !src || !dst || !workspace) return GS_ERR_INVALID_ARGUMENT;
    const DensifyLayout L(N);
    if (workspace_bytes < L.total) return GS_ERR_WORKSPACE;
    if ((int64_t)new_N > (int64_t)N * (2 + GS_SPLIT_SAMPLES)) return GS_ERR_INVALID_ARGUMENT;
    if (new_N == 0) return GS_OK;
    const char *ws = (const char *)workspace;
    const int32_t *map_p = (const int32_t *)(ws + L.map_p_off);
    const int32_t *map_m = (const int32_t *)(ws + L.map_m_off);
    const int32_t *sample_row = (const int32_t *)(ws + L.sample_off);
    hipStream_t s = (hipStream_t)stream;
    const int rest_len = (K - 1) * 3;
    for (int set = 0; set < 3; set++) {
        const GsGaussianSet &a = src[set], &b = dst[set];
        const float *srcs[6] = {a.means, a.log_scales, a.quats, a.opacity_logits, a.features_dc,
                                a.features_rest};
        float *dsts[6] = {b.means, b.log_scales, b.quats, b.opacity_logits, b.features_dc,
                          b.features_rest};
        const int lens[6] = {3, 3, 4, 1, 3, rest_len};
        for (int t = 0; t < 6; t++) {
            if (lens[t] == 0) continue;
            if (!dsts[t]) {
                if (set == 0) return GS_ERR_INVALID_ARGUMENT;  // parameters are mandatory
                continue;                                      // no optimiser state wanted
            }
            if (set == 0 && !srcs[t]) return GS_ERR_INVALID_ARGUMENT;
            const int64_t total = (int64_t)new_N * lens[t];
            const int64_t want = (total + 255) / 256;
            const int blocks = (int)(want < 16384 ? want : 16384);
            GS_LAUNCH(k_gather_rows, dim3(blocks), dim3(256), 0, s, total, lens[t], srcs[t],
                               set == 0 ? map_p : map_m, dsts[t]);
            GS_LAUNCH_CHECK();
        }
    }
    if (samples) {
        GS_LAUNCH(k_split_fixup, dim3((new_N + 255) / 256), dim3(256), 0, s, new_N, map_p,
                           sample_row, samples, src[0].means, src[0].log_scales, src[0].quats,
                           dst[0].means, dst[0].log_scales);
        GS_LAUNCH_CHECK();
    }
    return GS_OK;
}

extern "C" int gs_reset_opacity(int N, float reset_value, float *opacity_logits, float *exp_avg,
                                float *exp_avg_sq, gs_stream_t stream) {
    if (N < 0 || !(reset_value > 0.0f && reset_value < 1.0f)) return GS_ERR_INVALID_ARGUMENT;
    if (N == 0) return GS_OK;
    if (!opacity_logits) return GS_ERR_INVALID_ARGUMENT;
    const float max_logit = logf(reset_value / (1.0f - reset_value));  // torch::logit
    GS_LAUNCH(gs::k_reset_opacity, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       N, max_logit, opacity_logits, exp_avg, exp_avg_sq);
    GS_LAUNCH_CHECK();
    return GS_OK;
}
