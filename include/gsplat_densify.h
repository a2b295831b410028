/*
 * gsplat_densify.h — C ABI of the densification / culling kernels (first
 * half: Model::afterTrain, model.cpp:311-494; the on-disk formats of f4 are host code and not here).
 * Exported by `libgsplat_hip.so`; conventions as in gsplat_hip.h (device pointers unless marked
 * "host", fp32 / int32, work enqueued on `stream`, nothing allocated, GsStatus return codes).
 *
 * What each entry point replaces in OpenSplat:
 *   gs_densify_stats    the per-iteration statistics, model.cpp:317-337: visCounts / xysGradNorm /
 *                       max2DSize updated under `visibleMask` — in the reference ~10 boolean-mask
 *                       index ops, each with a host sync (nonzero); here one streaming kernel
 *   gs_densify_plan     the decisions of a refinement step, model.cpp:345-358,378-379,419-441:
 *                       split / duplicate / cull masks and the layout of the new Gaussian set
 *   gs_densify_apply    the tensor surgery, model.cpp:360-417,443-458 and the optimiser-state
 *                       surgery addToOptimizer / removeFromOptimizer, model.cpp:253-309: builds the
 *                       six parameter tensors and their twelve Adam moment tensors of the new set
 *   gs_reset_opacity    the "alpha reset", model.cpp:464-479
 *
 * Element order of the new set = the reference's: cat({originals, split samples (sample-major:
 * repeat({nSplitSamples, 1})), duplicates}) followed by index({~culls}).
 */
#ifndef GSPLAT_DENSIFY_H
#define GSPLAT_DENSIFY_H

#include "gsplat_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GS_SPLIT_SAMPLES 2 /* nSplitSamples, model.cpp:357 */

/* model.cpp:317-337, once per training iteration while step < stopSplitAt.
 *   xys_grad [N,2]   d loss / d xys (what xys.grad() holds; gs_rasterize_backward's v_xy)
 *   radii    [N]     this iteration's radii (visible = radii > 0)
 *   max_side         max(lastHeight, lastWidth)
 *   first            non-zero on the first call after a refinement cleared the statistics
 *                    (model.cpp:321-323,329-331: xysGradNorm = grads, visCounts = ones — for ALL
 *                    Gaussians, visible or not — max2DSize = zeros, then the visible update)
 * The three accumulators [N] are updated in place. */
int gs_densify_stats(int N, const float *xys_grad, const int32_t *radii, float max_side, int first,
                     float *xys_grad_norm, float *vis_counts, float *max_2d_size, gs_stream_t stream);

/* Scalars of one refinement step (all derived from `step` and the CLI options by the caller). */
typedef struct GsDensifyConfig {
    float half_max_side;       /* 0.5f * max(lastWidth, lastHeight), model.cpp:346          */
    float densify_grad_thresh; /* --densify-grad-thresh (0.0002)                            */
    float densify_size_thresh; /* --densify-size-thresh (0.01)                              */
    float split_screen_size;   /* --split-screen-size (0.05), used iff check_screen_size    */
    int32_t check_screen_size; /* step < stopScreenSizeAt, model.cpp:352,438                */
    float cull_alpha_thresh;   /* 0.1, model.cpp:343                                        */
    int32_t cull_huge;         /* step > refineEvery * resetAlphaEvery, model.cpp:434       */
    float cull_scale_thresh;   /* 0.5, model.cpp:435                                        */
    float cull_screen_size;    /* 0.15, model.cpp:436                                       */
} GsDensifyConfig;

/* counts written by gs_densify_plan (int32[8], HOST memory the device can write: pinned) */
enum {
    GS_DENSIFY_N_SPLITS = 0,    /* splits.sum(): the caller draws randn({2 * n_splits, 3})  */
    GS_DENSIFY_N_DUPS = 1,
    GS_DENSIFY_KEPT_ORIG = 2,   /* originals that survive the cull                          */
    GS_DENSIFY_KEPT_SPLIT = 3,  /* split SOURCES whose samples survive (x2 samples)         */
    GS_DENSIFY_KEPT_DUP = 4,
    GS_DENSIFY_NEW_N = 5,       /* kept_orig + 2 * kept_split + kept_dup                    */
    GS_DENSIFY_ADDED = 6,       /* 2 * n_splits + n_dups  ("Added ... gaussians")           */
    GS_DENSIFY_CULLED = 7       /* N + added - new_N      ("Culled ... gaussians")          */
};

size_t gs_densify_workspace_bytes(int N);

/* Phase 1.  Per Gaussian n (model.cpp line in brackets):
 *   high  = (xys_grad_norm / vis_counts) * 0.5 * max_side > densify_grad_thresh          [346-347]
 *   size  = max_k exp(log_scales[n,k])
 *   split = (size > densify_size_thresh | (check_screen & max_2d_size > split_screen)) & high [350-355]
 *   dup   = (size <= densify_size_thresh) & high   (a Gaussian can be BOTH, as in the reference) [378-379]
 *   culls: sigmoid(opacity) < cull_alpha | was split | (cull_huge & (size > cull_scale |
 *          (check_screen & max_2d_size > cull_screen)));  new Gaussians have max_2d_size = 0, split
 *          samples have scales log(exp(s) / 1.6)                                           [423-441]
 * and the exclusive scans that place every survivor.  `counts_host` receives the eight totals once
 * the stream reaches this point (synchronise before reading).  The plan stays in `workspace`. */
int gs_densify_plan(int N, const GsDensifyConfig *cfg, const float *xys_grad_norm,
                    const float *vis_counts, const float *max_2d_size, const float *log_scales,
                    const float *opacity_logits, int32_t *counts_host, void *workspace,
                    size_t workspace_bytes, gs_stream_t stream);

/* The six tensors of a Gaussian set (model.hpp): means [N,3], log-scales [N,3], raw quats [N,4],
 * opacity logits [N,1], featuresDc [N,3], featuresRest [N,K-1,3] (NULL when K == 1). */
typedef struct GsGaussianSet {
    float *means, *log_scales, *quats, *opacity_logits, *features_dc, *features_rest;
} GsGaussianSet;

/* Phase 2, after the caller has read the counts, drawn `samples` = randn({2 * n_splits, 3}) with
 * its own generator (torch::randn in the reference, model.cpp:360) and allocated the destination
 * tensors with NEW_N rows.  src[0] / dst[0] = parameters, [1] = Adam exp_avg, [2] = exp_avg_sq
 * (src[1], src[2] may hold NULL pointers: no optimiser state yet -> dst moments are zero-filled;
 * dst[1], dst[2] may hold NULL pointers: moments not wanted).  `samples` may be NULL only when
 * n_splits == 0.
 *   split sample j of the i-th split Gaussian n (i in splits order) uses samples row j * n_splits + i:
 *     mean = R(q / |q|) (exp(s) * sample) + mean_n,  scale = log(exp(s) / 1.6)            [361-373]
 *   everything else is copied; new Gaussians get zero moments (addToOptimizer), survivors keep
 *   theirs (removeFromOptimizer). */
int gs_densify_apply(int N, int K, int new_N /* counts[GS_DENSIFY_NEW_N] */, const float *samples,
                     const GsGaussianSet *src, const GsGaussianSet *dst, const void *workspace,
                     size_t workspace_bytes, gs_stream_t stream);

/* model.cpp:464-479: opacities = clamp_max(opacities, logit(reset_value)), and the opacity
 * optimiser's moments zeroed when given.  (The reference builds the zeroed AdamParamState but never
 * installs it, model.cpp:475-477 — pass NULL moments to reproduce exactly that.) */
int gs_reset_opacity(int N, float reset_value, float *opacity_logits, float *exp_avg,
                     float *exp_avg_sq, gs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_DENSIFY_H */
