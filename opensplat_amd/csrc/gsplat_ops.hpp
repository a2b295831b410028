// gsplat_ops.hpp — libtorch operator surface of the MI355X rasterizer.
//
// Same class names, argument order and meaning as OpenSplat's GPU operators, so that
// Model::forward (model.cpp:147-218) and simple_trainer.cpp:173-192 compile against this header
// unchanged:
//   ProjectGaussians     <- project_gaussians.hpp:12-30   (+ a 7th output, cov2d)
//   RasterizeGaussians   <- rasterize_gaussians.hpp:23-37 (+ an optional trailing cov2d argument)
//   SphericalHarmonics   <- spherical_harmonics.hpp:15-22
//   degFromSh / rgb2sh / sh2rgb <- spherical_harmonics.hpp:9-11
//   TileBounds           <- tile_bounds.hpp:6
// All device work goes through the C ABI in include/gsplat_hip.h (libgsplat_hip.so) on the
// current torch HIP stream; tensors are allocated with torch's caching allocator.
// Errors follow the reference's convention (bindings.h:14-19): TORCH_CHECK -> c10::Error.
#pragma once

#include <torch/torch.h>

#include <cstdint>
#include <functional>
#include <memory>
#include <tuple>
#include <vector>

typedef std::tuple<int, int, int> TileBounds;

#ifndef BLOCK_X
#define BLOCK_X 16
#define BLOCK_Y 16
#endif

int degFromSh(int numBases);
torch::Tensor rgb2sh(const torch::Tensor &rgb);
torch::Tensor sh2rgb(const torch::Tensor &sh);

class ProjectGaussians : public torch::autograd::Function<ProjectGaussians> {
public:
    // returns { xys[N,2], depths[N], radii[N] i32, conics[N,3], numTilesHit[N] i32, cov3d[N,6],
    //           cov2d[N,3] }  — the first six are the reference's outputs, in its order.
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext *ctx,
                                                  torch::Tensor means, torch::Tensor scales,
                                                  double globScale, torch::Tensor quats,
                                                  torch::Tensor viewMat, torch::Tensor projMat,
                                                  double fx, double fy, double cx, double cy,
                                                  int64_t imgHeight, int64_t imgWidth,
                                                  TileBounds tileBounds, double clipThresh = 0.01);
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext *ctx,
                                                 torch::autograd::tensor_list grad_outputs);
};

// Per-tile depth-ordered lists of one frame, as the compositing kernels consume them.
//   packed[N,12]               the 48-byte 2-D records (gs_pack_splats / gs_gaussian_forward)
//   gaussianIdsSorted[cap] i32 + blockMasks[cap] i16 (which 4x4-pixel blocks of its tile an entry
//                              reaches, gs_block_masks), tileBins[tiles,2] i32,
//   tileOrder[tiles] i32       tiles by descending list length,
//   count                      pinned host i32[2]: {M, longest list}, stored by the scan kernel,
//   listStats                  the same two numbers once validateBinning has run.
struct BinnedLists {
    torch::Tensor packed, gaussianIdsSorted, blockMasks, tileBins, tileOrder, count;
    int32_t listStats[2] = {0, 0};
    int width = 0, height = 0, device = 0;
    std::shared_ptr<void> scanDone;   // event recorded behind the scan kernel
};

// binAndSortPacked packs the per-Gaussian 2-D records and ENQUEUES tile counting, scan, scatter, the
// per-tile sorts and the coverage masks with an id-list capacity taken from the running maximum of the
// intersection counts seen so far for this (device, image size) — no host synchronisation.
// binPackedRecords does the same for records that already exist (gs_gaussian_forward).
// After enqueuing the compositing kernel the caller runs validateBinning(lists): it waits for the
// scan kernel's event (not for the stream), stores the frame's {M, longest list} in `lists` and returns
// false if the capacity was too small — repeat both steps then (the capacity has grown).
// (The reference's binAndSortGaussians, rasterize_gaussians.hpp:11-20, blocks on cumsum().item()
// before it can allocate, and takes radius-square tile counts; its five-tuple contract — caller-side
// cumulative counts, one global sort — is binAndSortGaussians below, on the launchers of
// bindings_hip_native.h.)
// The reference's own function, signature and five-tuple contract unchanged
// (rasterize_gaussians.hpp:11-20, rasterize_gaussians.cpp:6-37): (isectIds int64 [M], gaussianIds int32
// [M], isectIdsSorted, gaussianIdsSorted, tileBins int32 [rows, 2]) from caller-side cumulative
// radius-square tile counts — map_gaussian_to_intersects, one global torch::sort, gather, bin edges.
// For callers written against that contract; RasterizeGaussians::forward itself takes the route above.
typedef std::tuple<int, int, int> TileBounds;   // (tile_bounds.hpp's own typedef, repeated)
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
binAndSortGaussians(int numPoints, int numIntersects, torch::Tensor xys, torch::Tensor depths,
                    torch::Tensor radii, torch::Tensor cumTilesHit, TileBounds tileBounds);

BinnedLists binAndSortPacked(
    const torch::Tensor &xys, const torch::Tensor &depths, const torch::Tensor &radii,
    const torch::Tensor &conics, const torch::Tensor &colors, const torch::Tensor &opacity,
    const torch::Tensor &cov2d, int imgHeight, int imgWidth, bool opacityIsLogit = false);
BinnedLists binPackedRecords(const torch::Tensor &packed, const torch::Tensor &depths, int imgHeight,
                             int imgWidth);
bool validateBinning(BinnedLists &lists);
// Speculative-binning state: forget every capacity (e.g. after loading another scene); counters
// {binning calls, forwards repeated because the capacity was too small} since the last reset; the
// capacity currently held for an image size.
void gsplatResetBinningState();
std::tuple<int64_t, int64_t> gsplatBinningCounters();
int64_t gsplatBinningCapacity(int device, int imgWidth, int imgHeight);

// Reference-signature rasterize calls (ten arguments, no cov2d): {calls that found the frame's cov2d
// behind ProjectGaussians' own conics tensor, calls that had to invert the conic} since the last reset.
std::tuple<int64_t, int64_t> gsplatCov2dChannelCounters(bool reset = false);
// throws (c10::Error) if libgsplat_hip.so is not the ABI version libgsplat_torch.so was built against; C++ callers
// call it once at start-up (the Python side checks on import)
void gsplatCheckAbi();

class RasterizeGaussians : public torch::autograd::Function<RasterizeGaussians> {
public:
    // cov2d: the 7th output of ProjectGaussians.  When absent — the reference's own ten-argument call,
    // model.cpp:208-218 — it is recovered from the storage ProjectGaussians::forward shares between
    // conics and cov2d (same lists, same image as the eleven-argument call); only a `conics` tensor that
    // is not that operator's untouched output gets its rectangle from conic^-1 (counted above).
    static torch::Tensor forward(torch::autograd::AutogradContext *ctx, torch::Tensor xys,
                                 torch::Tensor depths, torch::Tensor radii, torch::Tensor conics,
                                 torch::Tensor numTilesHit, torch::Tensor colors,
                                 torch::Tensor opacity, int64_t imgHeight, int64_t imgWidth,
                                 torch::Tensor background,
                                 c10::optional<torch::Tensor> cov2d = c10::nullopt);
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext *ctx,
                                                 torch::autograd::tensor_list grad_outputs);
};

class SphericalHarmonics : public torch::autograd::Function<SphericalHarmonics> {
public:
    static torch::Tensor forward(torch::autograd::AutogradContext *ctx, int64_t degreesToUse,
                                 torch::Tensor viewDirs, torch::Tensor coeffs);
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext *ctx,
                                                 torch::autograd::tensor_list grad_outputs);
};

// SURVEY.md §8 row f1 — Model::forward's whole render chain (model.cpp:114-222) as one autograd
// node with the element-wise glue fused into the kernels.  Inputs are the RAW parameters OpenSplat
// optimises (log-scales, unnormalised quats, opacity logits, featuresDc / featuresRest), viewMat,
// projMat (= proj @ view), camPos = camera centre in world space (T of camToWorld, model.cpp:95).
// Returns { rgb[H,W,3] (clamp_max 1 applied), xys[N,2] (detached), radii[N] }.  The gradient of the
// loss w.r.t. xys (Model::afterTrain reads xys.grad(), model.cpp:318) is written to xysGradOut.
class SplatRender : public torch::autograd::Function<SplatRender> {
public:
    static torch::autograd::variable_list forward(
        torch::autograd::AutogradContext *ctx, torch::Tensor means, torch::Tensor logScales,
        torch::Tensor quats, torch::Tensor opacityLogits, torch::Tensor featuresDc,
        torch::Tensor featuresRest, torch::Tensor viewMat, torch::Tensor projMat,
        torch::Tensor camPos, double fx, double fy, double cx, double cy, int64_t imgHeight,
        int64_t imgWidth, int64_t degreesToUse, torch::Tensor background,
        c10::optional<torch::Tensor> xysGradOut = c10::nullopt);
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext *ctx,
                                                 torch::autograd::tensor_list grad_outputs);
};

// SURVEY.md §8 row f2 — Model::mainLoss (model.cpp:780-784): (1 - w) * l1(rgb, gt) + w * (1 - ssim)
// as ONE autograd node (fused L1 + SSIM kernels, include/gsplat_train.h).  The gradient w.r.t. rgb
// is computed together with the value; backward only scales it by the incoming gradient.
// `mainLoss(rgb, gt, ssimWeight)` is the drop-in for the member function (returns a 0-dim tensor).
class MainLoss : public torch::autograd::Function<MainLoss> {
public:
    static torch::Tensor forward(torch::autograd::AutogradContext *ctx, torch::Tensor rgb,
                                 torch::Tensor gt, double ssimWeight);
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext *ctx,
                                                 torch::autograd::tensor_list grad_outputs);
};
torch::Tensor mainLoss(const torch::Tensor &rgb, const torch::Tensor &gt, float ssimWeight);

// Model's six torch::optim::Adam instances (model.cpp:61-66) and Model::optimizersStep /
// optimizersZeroGrad (model.cpp:227-243) as one object: every group steps in ONE kernel launch.
// Parameters keep their identity (updated in place, like torch::optim::Adam does); gradients are
// read from param.grad().  Betas (0.9, 0.999), eps 1e-8 = libtorch's AdamOptions defaults.
class FusedAdam {
public:
    FusedAdam(std::vector<torch::Tensor> params, std::vector<double> lrs);
    void step();                      // all groups; groups without a gradient are skipped
    void zeroGrad();                  // optimizersZeroGrad
    void setLr(size_t group, double lr) { lrs_.at(group) = lr; }   // OptimScheduler::step
    double getLr(size_t group) const { return lrs_.at(group); }
    int64_t stepCount() const { return step_; }
    // optimiser-state access for densification surgery (model.cpp:253-309)
    torch::Tensor &expAvg(size_t group) { return expAvg_.at(group); }
    torch::Tensor &expAvgSq(size_t group) { return expAvgSq_.at(group); }
    void replaceParam(size_t group, torch::Tensor param, torch::Tensor expAvg, torch::Tensor expAvgSq);

private:
    std::vector<torch::Tensor> params_, expAvg_, expAvgSq_;
    std::vector<double> lrs_;
    int64_t step_ = 0;
};
// The same launch over state the CALLER owns (e.g. the exp_avg / exp_avg_sq tensors inside six
// torch::optim::Adam objects, model_fused.inl): params / expAvg / expAvgSq are updated in place;
// `step` is the 1-based step count after this update.  Contiguous float32 GPU tensors.
void fusedAdamStep(const std::vector<torch::Tensor> &params, const std::vector<torch::Tensor> &grads,
                   const std::vector<torch::Tensor> &expAvg, const std::vector<torch::Tensor> &expAvgSq,
                   const std::vector<double> &lrs, int64_t step);
// OptimScheduler::getLearningRate (optim_scheduler.cpp:4-7)
float schedulerLearningRate(float lrInit, float lrFinal, int maxSteps, int step);

// SURVEY.md §8 row f4 — Model::afterTrain's tensor work (model.cpp:311-494) on the device
// (include/gsplat_densify.h).  The schedule (refineEvery, warmupLength, ...) stays with the caller.
//   densifyStats   model.cpp:317-337; allocates the three accumulators when they are empty (the
//                  reference's `!xysGradNorm.numel()` first-call branch), else updates in place
//   densify        model.cpp:345-458 + addToOptimizer / removeFromOptimizer: returns the six
//                  parameter tensors and their Adam moments of the refined set; the normal samples
//                  come from torch::randn on the parameters' device, as in the reference
//   resetOpacity   model.cpp:464-479 (moments optional, see gs_reset_opacity)
void densifyStats(const torch::Tensor &xysGrad, const torch::Tensor &radii, int lastHeight,
                  int lastWidth, torch::Tensor &xysGradNorm, torch::Tensor &visCounts,
                  torch::Tensor &max2DSize);
struct DensifyResult {
    std::vector<torch::Tensor> params, expAvg, expAvgSq;  // means, scales, quats, opacities, featuresDc, featuresRest
    int nSplits = 0, nDups = 0, added = 0, culled = 0;
};
DensifyResult densify(const std::vector<torch::Tensor> &params, const std::vector<torch::Tensor> &expAvg,
                      const std::vector<torch::Tensor> &expAvgSq, const torch::Tensor &xysGradNorm,
                      const torch::Tensor &visCounts, const torch::Tensor &max2DSize, int lastWidth,
                      int lastHeight, float densifyGradThresh, float densifySizeThresh,
                      bool checkScreenSize, float splitScreenSize, bool cullHuge);
void resetOpacity(torch::Tensor &opacities, float resetValue,
                  c10::optional<torch::Tensor> expAvg = c10::nullopt,
                  c10::optional<torch::Tensor> expAvgSq = c10::nullopt);

// SURVEY.md §8e — the one exchange step of the camera-per-rank path for a C++ caller: a sum
// all-reduce of the flat gradient buffer (RCCL over xGMI, include/gsplat_dist.h) enqueued on the
// current torch HIP stream, i.e. right behind the backward kernels that filled the buffer.
// Bootstrap like ncclUniqueId: rank 0 calls uniqueId() and ships the bytes to the other ranks.
class GradExchange {
public:
    static std::vector<uint8_t> uniqueId();
    GradExchange(int worldSize, int rank, const std::vector<uint8_t> &id, int device);
    ~GradExchange();
    GradExchange(const GradExchange &) = delete;
    GradExchange &operator=(const GradExchange &) = delete;
    // in place; flat: contiguous float32 GPU tensor (e.g. the gradients of the six parameter
    // tensors viewed as slices of one buffer, or each tensor in turn)
    void allReduce(torch::Tensor flat);
    // as nBuckets collectives; returns nothing to wait for: consumers are ordered by the stream
    void allReduceBuckets(torch::Tensor flat, int nBuckets);
    // every rank's `message` (contiguous float32) into `gathered` [worldSize * message.numel()], rank order
    void allGather(torch::Tensor message, torch::Tensor gathered);
    // The factored exchange (include/gsplat_dist.h): `geometry` = the [v_means | v_scales | v_quats |
    // v_opacity] gradients as one contiguous buffer (summed in place over the ranks); `message` =
    // [camera centre, 4 floats | v_colour N x 3] as gs_gaussian_backward wrote it under
    // GS_FLAG_EMIT_VCOLOR; v_dc / v_rest receive the SH gradients of ALL ranks' cameras.
    void exchangeFactored(torch::Tensor geometry, torch::Tensor message, torch::Tensor gathered,
                          torch::Tensor means, int degreesToUse, torch::Tensor v_dc, torch::Tensor v_rest);
    int worldSize() const;
    int rank() const;

private:
    void *comm_ = nullptr;
};

// SURVEY.md §8e / f2 — the render work of ONE optimiser step over a BATCH of cameras on this rank, two of
// them in flight: camera j + 1's per-Gaussian forward, binning and compositing forward (HBM- / latency-bound) run
// on a second HIP stream under camera j's compositing backward (VALU-bound) — 1.2 x the serial loop at 1 M
// Gaussians, 1080p.  Generalises the per-image body of opensplat.cpp:151-170 (model.forward + backward); the
// Python twin is opensplat_amd.train.Trainer.train_step_batch.  The gradients of the six raw parameter tensors
// are overwritten by camera 0 and ACCUMULATED in camera order (an event orders the lanes' accumulation), i.e.
// exactly the sums of the serial loop, bit-identical with deterministic = true.  With an exchange the summed
// gradients are all-reduced once, behind the last camera (the factored exchange of the Python path is not
// offered here).  The caller's stream waits for the batch; no host synchronisation beyond the intersection-count
// validation of each frame (validateBinning).
struct BatchCamera {
    torch::Tensor viewMat, projMat, camPos;   // [4,4], [4,4] (= proj @ view), [3]; host or device
    double fx = 0, fy = 0, cx = 0, cy = 0;
};
class CameraBatch {
public:
    // cotangent(j, rgb): d loss / d rgb [H,W,3] of camera j from its clamped image; called with the lane's stream
    // current, so whatever it enqueues (mainLoss, a copy) joins that lane
    using Cotangent = std::function<torch::Tensor(int, const torch::Tensor &)>;
    CameraBatch(int64_t imgHeight, int64_t imgWidth);
    ~CameraBatch();
    CameraBatch(const CameraBatch &) = delete;
    CameraBatch &operator=(const CameraBatch &) = delete;
    // grads = { v_means [N,3], v_logScales [N,3], v_quats [N,4], v_opacityLogits [N] or [N,1], v_featuresDc [N,3],
    // v_featuresRest [N,K-1,3] (ignored for K = 1) }: contiguous float32 on the parameters' device.
    // serial: one camera after the other on one lane (the reference order: tests)
    void forwardBackward(const torch::Tensor &means, const torch::Tensor &logScales, const torch::Tensor &quats,
                         const torch::Tensor &opacityLogits, const torch::Tensor &featuresDc,
                         const torch::Tensor &featuresRest, const std::vector<BatchCamera> &cameras,
                         int64_t degreesToUse, const torch::Tensor &background, const Cotangent &cotangent,
                         std::vector<torch::Tensor> &grads, bool deterministic = false, bool serial = false,
                         GradExchange *exchange = nullptr);
    int64_t lastIntersections() const { return lastM_; }

private:
    struct Lane;
    std::unique_ptr<Lane> lanes_[2];
    int64_t H_, W_, lastM_ = 0;
};

// Process-wide switch for the compositing kernels' exponential: false (default) = glibc-bit-exact
// expf (contributor sets identical to gsplat-cpu); true = hardware v_exp_f32 (GS_FLAG_FAST_EXP).
void gsplatSetFastExp(bool enabled);
bool gsplatGetFastExp();
// Process-wide switch for SplatRender on frames of few tiles (gsplat_hip.h: gs_rasterize_checkpoint_plan):
// true (default) = the forward leaves checkpoints along the tile lists and the backward runs the pieces of a
// list side by side; false = the one-pass backward everywhere.  Scheduling only.
void gsplatSetSegmentedBackward(bool enabled);
bool gsplatGetSegmentedBackward();
