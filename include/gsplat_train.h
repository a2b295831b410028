/*
 * gsplat_train.h — C ABI of the training-step kernels around the rasterizer,
 * exported by the same `libgsplat_hip.so` as gsplat_hip.h and following its conventions (device
 * pointers unless marked "host", fp32, work enqueued on `stream`, nothing allocated, GsStatus
 * return codes, re-entrant).
 *
 * What each entry point replaces in OpenSplat:
 *   gs_main_loss        Model::mainLoss              model.cpp:780-784
 *                         = (1 - w) * l1(rgb, gt)    model.cpp:54-56
 *                         + w * (1 - SSIM::eval)     ssim.cpp:7-33 (11x11 window ssim.cpp:33-45,
 *                                                    model.hpp:32 ssim(11, 3))
 *                       and the autograd backward of that expression (opensplat.cpp:160-161
 *                       `mainLoss.backward()`): the gradient w.r.t. the rendered image — exactly the
 *                       cotangent RasterizeGaussians::backward consumes — comes out of the same call;
 *   gs_adam_step        Model::optimizersStep        model.cpp:236-243: six torch::optim::Adam
 *                       instances built with AdamOptions(lr) only (model.cpp:61-66), i.e. libtorch's
 *                       defaults betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad — all
 *                       six parameter groups in ONE launch;
 *   gs_sched_lr         OptimScheduler::getLearningRate  optim_scheduler.cpp:4-7 (host function).
 */
#ifndef GSPLAT_TRAIN_H
#define GSPLAT_TRAIN_H

#include "gsplat_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GS_SSIM_WINDOW 11 /* model.hpp:32 */

/* The reference's 1-D window (ssim.cpp:39-45), normalised as libtorch does; the 2-D window of
 * createWindow() (ssim.cpp:33-37) is its outer product.  NB the reference evaluates
 * exp(-floor((i - 11) / 2)^2 / (2 sigma^2)): the window is NOT symmetric; that is reproduced.
 * `g` is a HOST array of 11 floats. */
int gs_ssim_window(float *g);

/* Bytes of device workspace gs_main_loss needs for a W x H image. */
size_t gs_loss_workspace_bytes(int W, int H);

/* loss = (1 - ssim_weight) * mean|gt - rendered| + ssim_weight * (1 - mean(ssim_map)), and
 * v_rendered = grad_scale * d loss / d rendered.
 *   rendered, gt   [H, W, 3]   (Model::forward's output and Camera::getImage's, model.cpp:222)
 *   loss           float[3]    { mainLoss, l1, ssim } (device memory; written by the last kernel)
 *   v_rendered     [H, W, 3]   or NULL for the value only
 *   grad_scale     the upstream gradient of the scalar loss: 1 for `mainLoss.backward()`,
 *                  1/B when B cameras' losses are averaged in one optimiser step
 * ssim_weight = 0 skips the SSIM kernels entirely (`--ssim-weight 0`, opensplat.cpp:36).
 * Zero padding, per-channel (grouped) convolution and the constants C1 = 0.01^2, C2 = 0.03^2 are
 * those of ssim.cpp:16-29.  The convolutions are evaluated separably (the window is an outer
 * product) in fp32, so values agree with libtorch's direct conv2d to rounding, not bit for bit. */
int gs_main_loss(int W, int H, const float *rendered, const float *gt, float ssim_weight,
                 float grad_scale, float *loss, float *v_rendered, void *workspace,
                 size_t workspace_bytes, gs_stream_t stream);

/* One parameter group of the optimiser: Model's means / scales / quats / featuresDc /
 * featuresRest / opacities (model.cpp:61-66), each with its own learning rate. */
typedef struct GsAdamGroup {
    float *param;       /* [n] updated in place                                   */
    const float *grad;  /* [n]                                                    */
    float *exp_avg;     /* [n] first moment, zero-initialised by the caller       */
    float *exp_avg_sq;  /* [n] second moment, zero-initialised by the caller      */
    int64_t n;
    double lr;          /* AdamOptions::lr is a double (model.cpp:61-66)          */
} GsAdamGroup;

#define GS_ADAM_MAX_GROUPS 8

/* torch::optim::Adam::step for up to GS_ADAM_MAX_GROUPS groups (host array) in one launch:
 *   m = beta1 m + (1 - beta1) g;   v = beta2 v + (1 - beta2) g g;
 *   p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * `step` is the 1-based count of optimiser steps taken including this one (all of Model's
 * optimisers step together).  Bias corrections in double on the host, tensor arithmetic in fp32
 * with the operation order (and fused multiply-adds) of ATen's CPU kernels: the states stay
 * bit-identical to libtorch's CPU Adam (tests). */
int gs_adam_step(int num_groups, const GsAdamGroup *groups, int64_t step, double beta1, double beta2,
                 double eps, gs_stream_t stream);

/* The same step for a CAPTURED HIP GRAPH (the training iteration of a small frame is launch-bound: ~25
 * launches of a few microseconds each; replayed as one graph it is not).  A graph replays its
 * launches with the arguments of the capture, so whatever changes from step to step must come from device
 * memory: the per-step scalars sit in a table the host fills ahead of time,
 *     row = { sqrt(1 - beta2^step), -(lr_g / (1 - beta1^step)) for the GS_ADAM_MAX_GROUPS groups }
 * (gs_adam_schedule_row: the very expressions gs_adam_step evaluates, so both forms move the same bits), the
 * kernel takes row *row_index_dev, and gs_adam_advance — one thread, also captured — steps the index.
 * guard_dev (optional): the launch does nothing when *guard_dev > guard_max — the caller points it at the
 * intersection count of a speculative binning (gs_bin_scan's num_isects_dev) with guard_max = the id list's
 * capacity: a step whose lists were truncated leaves parameters, moments and the row index untouched, and
 * the host, seeing the count afterwards, re-runs it with a larger list.  GsAdamGroup.lr is ignored here. */
#define GS_ADAM_ROW_FLOATS (1 + GS_ADAM_MAX_GROUPS)
/* Inputs of a captured iteration that change from replay to replay (the camera, which target image) are
 * written by the host into PINNED, device-mapped memory and fetched by the graph's own first nodes:
 *   gs_stage_f32          dst_dev[0..count) <- src_pinned_host[0..count)            (count <= 4096)
 *   gs_copy_indirect_f32  dst_dev[0..count) <- (*src_ptr_pinned_host)[0..count): the pinned word holds the
 *                         DEVICE address of the source (e.g. this iteration's ground-truth image)
 * so that nothing but the graph launch itself is enqueued between the host's writes and the replay. */
int gs_stage_f32(float *dst_dev, const float *src_pinned_host, int count, gs_stream_t stream);
int gs_copy_indirect_f32(float *dst_dev, const float *const *src_ptr_pinned_host, int64_t count,
                         gs_stream_t stream);
int gs_adam_schedule_row(int num_groups, const double *lrs /* host */, int64_t step, double beta1,
                         double beta2, float *row /* host, GS_ADAM_ROW_FLOATS */);
int gs_adam_step_scheduled(int num_groups, const GsAdamGroup *groups /* host */, const float *rows_dev,
                           const int32_t *row_index_dev, int32_t num_rows, const int32_t *guard_dev,
                           int32_t guard_max, double beta1, double beta2, double eps, gs_stream_t stream);
int gs_adam_advance(int32_t *row_index_dev, const int32_t *guard_dev, int32_t guard_max, gs_stream_t stream);

/* exp(log(lr_init) (1 - t) + log(lr_final) t), t = clamp(step / max_steps, 0, 1)
 * (optim_scheduler.cpp:4-7; Model uses it for the means only, model.cpp:68,245-247). */
float gs_sched_lr(float lr_init, float lr_final, int max_steps, int step);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_TRAIN_H */
