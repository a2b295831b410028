// TEST INFRASTRUCTURE ONLY — never linked into, imported by, or called from the product path.
//
// C-ABI shim around the reference's own training-step pieces of SURVEY.md §8 row f2, compiled IN
// PLACE from /root/reference by oracle/Makefile into oracle/_ref/libgsplat_ref.so:
//   SSIM::eval / createWindow / gaussian     ssim.cpp:7-45          (compiled from the reference)
//   OptimScheduler::getLearningRate          optim_scheduler.cpp:4-7 (compiled from the reference)
//   l1                                       model.cpp:54-56   } model.cpp needs OpenCV / nanoflann /
//   Model::mainLoss                          model.cpp:780-784 } json and cannot be compiled here:
//                                                              these five lines are RESTATED below
//   optimiser                                model.cpp:61-66: torch::optim::Adam(AdamOptions(lr)),
//                                            i.e. libtorch's own Adam (third-party, the version in
//                                            this image: 2.10) with betas (0.9, 0.999), eps 1e-8
// Used to pin oracle/train_oracle.c and to generate tests/golden/train_*.npz.
//
// All pointers are HOST pointers to contiguous fp32 data; 0 on success, -1 if libtorch threw.

#include <torch/torch.h>

#include <chrono>
#include <cstring>
#include <string>

// SSIM keeps its window private; the fixture wants to look at it.  Test infrastructure only.
#define private public
#include "ssim.hpp"
#undef private
#include "optim_scheduler.hpp"

namespace {
thread_local std::string g_train_err;
double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
}  // namespace

extern "C" const char *ref_train_last_error() { return g_train_err.c_str(); }

// 1-D Gaussian of the reference's 11x11 window: createWindow() = outer(g, g), so g = row sums.
extern "C" int ref_ssim_window(float *window2d_121, float *g11) {
    try {
        SSIM ssim(11, 3);
        torch::Tensor w = ssim.window.index({0, 0}).contiguous();  // [11, 11]
        std::memcpy(window2d_121, w.data_ptr<float>(), sizeof(float) * 121);
        torch::Tensor g = ssim.gaussian(1.5f).contiguous();
        std::memcpy(g11, g.data_ptr<float>(), sizeof(float) * 11);
        return 0;
    } catch (const std::exception &e) {
        g_train_err = e.what();
        return -1;
    }
}

// loss3 = { mainLoss, l1, ssim (the similarity, not 1 - ssim) }; v_rendered = d mainLoss / d rendered
// through libtorch autograd (opensplat.cpp:160-161: mainLoss.backward()).
extern "C" int ref_main_loss(int W, int H, const float *rendered, const float *gt, float ssim_weight,
                             float *loss3, float *v_rendered, double *ms) {
    try {
        SSIM ssim(11, 3);
        torch::Tensor rgb = torch::from_blob(const_cast<float *>(rendered), {H, W, 3}, torch::kFloat32)
                                .clone().requires_grad_(true);
        torch::Tensor g = torch::from_blob(const_cast<float *>(gt), {H, W, 3}, torch::kFloat32).clone();
        const double t0 = now_ms();
        torch::Tensor ssimVal = ssim.eval(rgb, g);
        torch::Tensor ssimLoss = 1.0f - ssimVal;                               // model.cpp:781
        torch::Tensor l1Loss = torch::abs(g - rgb).mean();                     // model.cpp:54-56,782
        torch::Tensor loss = (1.0f - ssim_weight) * l1Loss + ssim_weight * ssimLoss;  // :783
        if (v_rendered) loss.backward();
        const double t1 = now_ms();
        if (ms) *ms = t1 - t0;
        loss3[0] = loss.item<float>();
        loss3[1] = l1Loss.item<float>();
        loss3[2] = ssimVal.item<float>();
        if (v_rendered)
            std::memcpy(v_rendered, rgb.grad().contiguous().data_ptr<float>(),
                        sizeof(float) * (size_t)H * W * 3);
        return 0;
    } catch (const std::exception &e) {
        g_train_err = e.what();
        return -1;
    }
}

// `steps` Adam steps on one parameter tensor with the given per-step gradients [steps, n], exactly
// as Model::optimizersStep drives each of its six optimisers (model.cpp:61-66,236-243).
extern "C" int ref_adam_steps(int64_t n, float *param, const float *grads, int steps, double lr,
                              float *exp_avg, float *exp_avg_sq) {
    try {
        torch::Tensor p = torch::from_blob(param, {n}, torch::kFloat32).clone().requires_grad_(true);
        torch::optim::Adam opt({p}, torch::optim::AdamOptions(lr));
        for (int s = 0; s < steps; s++) {
            opt.zero_grad();
            p.mutable_grad() = torch::from_blob(const_cast<float *>(grads + (size_t)s * n), {n},
                                                torch::kFloat32).clone();
            opt.step();
        }
        std::memcpy(param, p.detach().contiguous().data_ptr<float>(), sizeof(float) * n);
        auto &st = static_cast<torch::optim::AdamParamState &>(*opt.state().begin()->second);
        if (exp_avg) std::memcpy(exp_avg, st.exp_avg().contiguous().data_ptr<float>(), sizeof(float) * n);
        if (exp_avg_sq)
            std::memcpy(exp_avg_sq, st.exp_avg_sq().contiguous().data_ptr<float>(), sizeof(float) * n);
        return 0;
    } catch (const std::exception &e) {
        g_train_err = e.what();
        return -1;
    }
}

// OptimScheduler(meansOpt, lrFinal, maxSteps).getLearningRate(step), model.cpp:68.
extern "C" float ref_sched_lr(float lr_init, float lr_final, int max_steps, int step) {
    torch::Tensor p = torch::zeros({1}).requires_grad_(true);
    torch::optim::Adam opt({p}, torch::optim::AdamOptions(lr_init));
    OptimScheduler sched(&opt, lr_final, max_steps);
    return sched.getLearningRate(step);
}

// ---- Model::afterTrain (model.cpp:311-494) --------------------------------
// model.cpp itself cannot be compiled here (OpenCV, nanoflann, json, ...), so the statements of
// afterTrain that touch tensors are RESTATED below, in the reference's order and with the
// reference's torch calls, as free functions over explicit tensors; quatToRotMat is the reference's
// own (tensor_math.cpp, compiled in place).  The optimiser-state surgery follows addToOptimizer /
// removeFromOptimizer (model.cpp:253-309).
#include "tensor_math.hpp"

using namespace torch::indexing;

namespace {
torch::Tensor tf(const float *p, std::initializer_list<int64_t> shape) {
    return torch::from_blob(const_cast<float *>(p), shape, torch::kFloat32).clone();
}
void out_f(const torch::Tensor &t, float *dst) {
    if (!dst) return;
    torch::Tensor c = t.detach().to(torch::kFloat32).contiguous();
    std::memcpy(dst, c.data_ptr<float>(), sizeof(float) * c.numel());
}
}  // namespace

// model.cpp:317-337.  `first`: xysGradNorm.numel() == 0.  Accumulators [N] in/out.
extern "C" int ref_densify_stats(int N, const float *xys_grad, const int32_t *radii_in, int lastHeight,
                                 int lastWidth, int first, float *xysGradNorm_io, float *visCounts_io,
                                 float *max2DSize_io) {
    try {
        torch::Tensor xysGrad = tf(xys_grad, {N, 2});
        torch::Tensor radii = torch::from_blob(const_cast<int32_t *>(radii_in), {N}, torch::kInt32).clone();
        torch::Tensor xysGradNorm = first ? torch::Tensor() : tf(xysGradNorm_io, {N});
        torch::Tensor visCounts = first ? torch::Tensor() : tf(visCounts_io, {N});
        torch::Tensor max2DSize = first ? torch::Tensor() : tf(max2DSize_io, {N});

        torch::Tensor visibleMask = (radii > 0).flatten();
        torch::Tensor grads = torch::linalg_vector_norm(xysGrad.detach(), 2, {-1}, false, torch::kFloat32);
        if (!xysGradNorm.numel()) {
            xysGradNorm = grads;
            visCounts = torch::ones_like(xysGradNorm);
        } else {
            visCounts.index_put_({visibleMask}, visCounts.index({visibleMask}) + 1);
            xysGradNorm.index_put_({visibleMask}, grads.index({visibleMask}) + xysGradNorm.index({visibleMask}));
        }
        if (!max2DSize.numel()) {
            max2DSize = torch::zeros_like(radii, torch::kFloat32);
        }
        torch::Tensor newRadii = radii.detach().index({visibleMask});
        max2DSize.index_put_({visibleMask}, torch::maximum(
            max2DSize.index({visibleMask}), newRadii / static_cast<float>((std::max)(lastHeight, lastWidth))));
        out_f(xysGradNorm, xysGradNorm_io);
        out_f(visCounts, visCounts_io);
        out_f(max2DSize, max2DSize_io);
        return 0;
    } catch (const std::exception &e) {
        g_train_err = e.what();
        return -1;
    }
}

// model.cpp:345-458 (the doDensification branch: split, duplicate, cull) on parameter tensors and
// the two Adam moment tensors of each.  Two-call protocol so that the caller can supply the normal
// samples (torch::randn in the reference) that the product path is given too:
//   call 1: samples == NULL -> returns n_splits in counts[0]
//   call 2: samples [2 * n_splits, 3] -> fills the outputs (capacity 4N rows each), new N in counts[1]
// params / moments order: means, scales, quats, opacities, featuresDc, featuresRest.
extern "C" int ref_densify_refine(int N, int K, const float *const *params, const float *const *exp_avg_in,
                                  const float *const *exp_avg_sq_in, const float *xysGradNorm_in,
                                  const float *visCounts_in, const float *max2DSize_in, int lastWidth,
                                  int lastHeight, float densifyGradThresh, float densifySizeThresh,
                                  int checkScreenSize /* step < stopScreenSizeAt */, float splitScreenSize,
                                  int cullHuge /* step > refineEvery * resetAlphaEvery */,
                                  const float *samples, float *const *out_params, float *const *out_exp_avg,
                                  float *const *out_exp_avg_sq, int32_t *counts) {
    try {
        torch::Tensor means = tf(params[0], {N, 3}), scales = tf(params[1], {N, 3}),
                      quats = tf(params[2], {N, 4}), opacities = tf(params[3], {N, 1}),
                      featuresDc = tf(params[4], {N, 3}),
                      featuresRest = tf(params[5], {N, K - 1, 3});
        std::vector<torch::Tensor> ea, es;
        const std::vector<std::vector<int64_t>> shapes = {{N, 3}, {N, 3}, {N, 4}, {N, 1}, {N, 3}, {N, K - 1, 3}};
        for (int i = 0; i < 6; i++) {
            ea.push_back(torch::from_blob(const_cast<float *>(exp_avg_in[i]), shapes[i], torch::kFloat32).clone());
            es.push_back(torch::from_blob(const_cast<float *>(exp_avg_sq_in[i]), shapes[i], torch::kFloat32).clone());
        }
        torch::Tensor xysGradNorm = tf(xysGradNorm_in, {N}), visCounts = tf(visCounts_in, {N}),
                      max2DSize = tf(max2DSize_in, {N});
        const float cullAlphaThresh = 0.1f;

        torch::Tensor avgGradNorm = (xysGradNorm / visCounts) * 0.5f * static_cast<float>((std::max)(lastWidth, lastHeight));
        torch::Tensor highGrads = (avgGradNorm > densifyGradThresh).squeeze();
        torch::Tensor splits = (std::get<0>(scales.exp().max(-1)) > densifySizeThresh).squeeze();
        if (checkScreenSize) {
            splits |= (max2DSize > splitScreenSize).squeeze();
        }
        splits &= highGrads;
        const int nSplitSamples = 2;
        int nSplits = splits.sum().item<int>();
        counts[0] = nSplits;
        if (!samples && nSplits > 0) return 0;  // call 1

        torch::Tensor centeredSamples = samples ? tf(samples, {nSplitSamples * nSplits, 3})
                                                : torch::zeros({0, 3});
        torch::Tensor scaledSamples = torch::exp(scales.index({splits}).repeat({nSplitSamples, 1})) * centeredSamples;
        torch::Tensor qs = quats.index({splits}) / torch::linalg_vector_norm(quats.index({splits}), 2, {-1}, true, torch::kFloat32);
        torch::Tensor rots = quatToRotMat(qs.repeat({nSplitSamples, 1}));
        torch::Tensor rotatedSamples = torch::bmm(rots, scaledSamples.index({"...", None})).squeeze(-1);
        torch::Tensor splitMeans = rotatedSamples + means.index({splits}).repeat({nSplitSamples, 1});
        torch::Tensor splitFeaturesDc = featuresDc.index({splits}).repeat({nSplitSamples, 1});
        torch::Tensor splitFeaturesRest = featuresRest.index({splits}).repeat({nSplitSamples, 1, 1});
        torch::Tensor splitOpacities = opacities.index({splits}).repeat({nSplitSamples, 1});
        const float sizeFac = 1.6f;
        torch::Tensor splitScales = torch::log(torch::exp(scales.index({splits})) / sizeFac).repeat({nSplitSamples, 1});
        scales.index({splits}) = torch::log(torch::exp(scales.index({splits})) / sizeFac);  // (no effect: index() copies)
        torch::Tensor splitQuats = quats.index({splits}).repeat({nSplitSamples, 1});

        torch::Tensor dups = (std::get<0>(scales.exp().max(-1)) <= densifySizeThresh).squeeze();
        dups &= highGrads;
        torch::Tensor dupMeans = means.index({dups});
        torch::Tensor dupFeaturesDc = featuresDc.index({dups});
        torch::Tensor dupFeaturesRest = featuresRest.index({dups});
        torch::Tensor dupOpacities = opacities.index({dups});
        torch::Tensor dupScales = scales.index({dups});
        torch::Tensor dupQuats = quats.index({dups});

        means = torch::cat({means.detach(), splitMeans, dupMeans}, 0);
        featuresDc = torch::cat({featuresDc.detach(), splitFeaturesDc, dupFeaturesDc}, 0);
        featuresRest = torch::cat({featuresRest.detach(), splitFeaturesRest, dupFeaturesRest}, 0);
        opacities = torch::cat({opacities.detach(), splitOpacities, dupOpacities}, 0);
        scales = torch::cat({scales.detach(), splitScales, dupScales}, 0);
        quats = torch::cat({quats.detach(), splitQuats, dupQuats}, 0);
        max2DSize = torch::cat({max2DSize, torch::zeros_like(splitScales.index({Slice(), 0})),
                                torch::zeros_like(dupScales.index({Slice(), 0}))}, 0);

        torch::Tensor splitIdcs = torch::where(splits)[0];
        torch::Tensor dupIdcs = torch::where(dups)[0];
        // addToOptimizer (model.cpp:253-287): exp_avg / exp_avg_sq grow by zeros
        auto grow = [](torch::Tensor &state, const torch::Tensor &idcs, int nSamples) {
            std::vector<int64_t> repeats;
            repeats.push_back(nSamples);
            for (long int i = 0; i < state.dim() - 1; i++) repeats.push_back(1);
            state = torch::cat({state, torch::zeros_like(state.index({idcs.squeeze()})).repeat(repeats)}, 0);
        };
        for (int i = 0; i < 6; i++) { grow(ea[i], splitIdcs, nSplitSamples); grow(es[i], splitIdcs, nSplitSamples); }
        for (int i = 0; i < 6; i++) { grow(ea[i], dupIdcs, 1); grow(es[i], dupIdcs, 1); }

        torch::Tensor splitsMask = torch::cat({splits,
            torch::full({nSplitSamples * splits.sum().item<int>() + dups.sum().item<int>()}, false,
                        torch::TensorOptions().dtype(torch::kBool))}, 0);

        // Cull (model.cpp:419-458)
        torch::Tensor culls = (torch::sigmoid(opacities) < cullAlphaThresh).squeeze();
        if (splitsMask.numel()) culls |= splitsMask;
        if (cullHuge) {
            const float cullScaleThresh = 0.5f;
            const float cullScreenSize = 0.15f;
            torch::Tensor huge = std::get<0>(torch::exp(scales).max(-1)) > cullScaleThresh;
            if (checkScreenSize) huge |= max2DSize > cullScreenSize;
            culls |= huge;
        }
        int cullCount = torch::sum(culls).item<int>();
        if (cullCount > 0) {
            means = means.index({~culls});
            scales = scales.index({~culls});
            quats = quats.index({~culls});
            featuresDc = featuresDc.index({~culls});
            featuresRest = featuresRest.index({~culls});
            opacities = opacities.index({~culls});
            for (int i = 0; i < 6; i++) {   // removeFromOptimizer (model.cpp:289-309)
                ea[i] = ea[i].index({~culls});
                es[i] = es[i].index({~culls});
            }
        }
        counts[1] = (int32_t)means.size(0);
        counts[2] = dups.sum().item<int>();
        counts[3] = cullCount;
        const torch::Tensor outs[6] = {means, scales, quats, opacities, featuresDc, featuresRest};
        for (int i = 0; i < 6; i++) {
            out_f(outs[i], out_params[i]);
            out_f(ea[i], out_exp_avg[i]);
            out_f(es[i], out_exp_avg_sq[i]);
        }
        return 0;
    } catch (const std::exception &e) {
        g_train_err = e.what();
        return -1;
    }
}

// ---- row f4, second half: Model::savePly / saveSplat (model.cpp:505-598) restated ----------------
// Same statements, free functions over explicit tensors (model.cpp cannot be compiled here).
#include <fstream>
#include <numeric>

#include "spherical_harmonics.hpp"

extern "C" int ref_save_ply(const char *filename, int numPoints, int K, const float *means_, const float *scales_,
                            const float *quats_, const float *opacities_, const float *featuresDc_,
                            const float *featuresRest_, int step, int keepCrs, float scale,
                            const float *translation_) {
    try {
        torch::Tensor means = tf(means_, {numPoints, 3}), scales = tf(scales_, {numPoints, 3}),
                      quats = tf(quats_, {numPoints, 4}), opacities = tf(opacities_, {numPoints, 1}),
                      featuresDc = tf(featuresDc_, {numPoints, 3}),
                      featuresRest = tf(featuresRest_, {numPoints, K - 1, 3});
        torch::Tensor translation = translation_ ? tf(translation_, {3}) : torch::zeros({3});
        std::ofstream o(filename, std::ios::binary);
        o << "ply" << std::endl;
        o << "format binary_little_endian 1.0" << std::endl;
        o << "comment Generated by opensplat at iteration " << step << std::endl;
        o << "element vertex " << numPoints << std::endl;
        o << "property float x" << std::endl;
        o << "property float y" << std::endl;
        o << "property float z" << std::endl;
        o << "property float nx" << std::endl;
        o << "property float ny" << std::endl;
        o << "property float nz" << std::endl;
        for (int i = 0; i < featuresDc.size(1); i++) o << "property float f_dc_" << i << std::endl;
        torch::Tensor featuresRestCpu = featuresRest.cpu().transpose(1, 2).reshape({numPoints, -1});
        for (int i = 0; i < featuresRestCpu.size(1); i++) o << "property float f_rest_" << i << std::endl;
        o << "property float opacity" << std::endl;
        o << "property float scale_0" << std::endl;
        o << "property float scale_1" << std::endl;
        o << "property float scale_2" << std::endl;
        o << "property float rot_0" << std::endl;
        o << "property float rot_1" << std::endl;
        o << "property float rot_2" << std::endl;
        o << "property float rot_3" << std::endl;
        o << "end_header" << std::endl;
        float zeros[] = {0.0f, 0.0f, 0.0f};
        torch::Tensor meansCpu = keepCrs ? (means.cpu() / scale) + translation : means.cpu();
        torch::Tensor featuresDcCpu = featuresDc.cpu();
        torch::Tensor opacitiesCpu = opacities.cpu();
        torch::Tensor scalesCpu = keepCrs ? torch::log((torch::exp(scales.cpu()) / scale)) : scales.cpu();
        torch::Tensor quatsCpu = quats.cpu();
        for (size_t i = 0; i < (size_t)numPoints; i++) {
            o.write(reinterpret_cast<const char *>(meansCpu[i].data_ptr()), sizeof(float) * 3);
            o.write(reinterpret_cast<const char *>(zeros), sizeof(float) * 3);
            o.write(reinterpret_cast<const char *>(featuresDcCpu[i].data_ptr()), sizeof(float) * featuresDcCpu.size(1));
            o.write(reinterpret_cast<const char *>(featuresRestCpu[i].contiguous().data_ptr()), sizeof(float) * featuresRestCpu.size(1));
            o.write(reinterpret_cast<const char *>(opacitiesCpu[i].data_ptr()), sizeof(float) * 1);
            o.write(reinterpret_cast<const char *>(scalesCpu[i].data_ptr()), sizeof(float) * 3);
            o.write(reinterpret_cast<const char *>(quatsCpu[i].data_ptr()), sizeof(float) * 4);
        }
        o.close();
        return 0;
    } catch (const std::exception &e) {
        g_train_err = e.what();
        return -1;
    }
}

extern "C" int ref_save_splat(const char *filename, int numPoints, const float *means_, const float *scales_,
                              const float *quats_, const float *opacities_, const float *featuresDc_,
                              int keepCrs, float scale, const float *translation_) {
    try {
        torch::Tensor means = tf(means_, {numPoints, 3}), scales = tf(scales_, {numPoints, 3}),
                      quats = tf(quats_, {numPoints, 4}), opacities = tf(opacities_, {numPoints, 1}),
                      featuresDc = tf(featuresDc_, {numPoints, 3});
        torch::Tensor translation = translation_ ? tf(translation_, {3}) : torch::zeros({3});
        std::ofstream o(filename, std::ios::binary);
        torch::Tensor meansCpu = keepCrs ? (means.cpu() / scale) + translation : means.cpu();
        torch::Tensor scalesCpu = keepCrs ? (torch::exp(scales.cpu()) / scale) : torch::exp(scales.cpu());
        torch::Tensor rgbsCpu = (sh2rgb(featuresDc.cpu()) * 255.0f).toType(torch::kUInt8);
        torch::Tensor opac = (1.0f + torch::exp(-opacities.cpu()));
        torch::Tensor opacitiesCpu = torch::clamp(((1.0f / opac) * 255.0f), 0.0f, 255.0f).toType(torch::kUInt8);
        torch::Tensor quatsCpu = torch::clamp(quats.cpu() * 128.0f + 128.0f, 0.0f, 255.0f).toType(torch::kUInt8);
        std::vector<size_t> splatIndices(numPoints);
        std::iota(splatIndices.begin(), splatIndices.end(), 0);
        torch::Tensor order = (scalesCpu.index({"...", 0}) + scalesCpu.index({"...", 1}) + scalesCpu.index({"...", 2})) /
                              opac.index({"...", 0});
        order = order.contiguous();
        float *orderPtr = reinterpret_cast<float *>(order.data_ptr());
        std::sort(splatIndices.begin(), splatIndices.end(),
                  [&orderPtr](size_t const &a, size_t const &b) { return orderPtr[a] > orderPtr[b]; });
        for (int i = 0; i < numPoints; i++) {
            size_t idx = splatIndices[i];
            o.write(reinterpret_cast<const char *>(meansCpu[idx].data_ptr()), sizeof(float) * 3);
            o.write(reinterpret_cast<const char *>(scalesCpu[idx].data_ptr()), sizeof(float) * 3);
            o.write(reinterpret_cast<const char *>(rgbsCpu[idx].data_ptr()), sizeof(uint8_t) * 3);
            o.write(reinterpret_cast<const char *>(opacitiesCpu[idx].data_ptr()), sizeof(uint8_t) * 1);
            o.write(reinterpret_cast<const char *>(quatsCpu[idx].data_ptr()), sizeof(uint8_t) * 4);
        }
        o.close();
        return 0;
    } catch (const std::exception &e) {
        g_train_err = e.what();
        return -1;
    }
}

// ---- COLMAP poses, camera set-up, model initialisation --------------------------
// quatToRotMat and autoScaleAndCenterPoses are the reference's own (tensor_math.cpp, compiled in place);
// the statements of colmap.cpp:88-118 (pose of one image record), model.cpp:35-47,85-113 (projection
// matrix, camera block of Model::forward) and model.cpp:23-33 + model.hpp:37,41 (random quaternions)
// are restated verbatim: colmap.cpp / model.cpp include OpenCV headers and cannot be compiled here.
#include "constants.hpp"

extern "C" int ref_quat_to_rotmat(const float *q4, float *R9) {
    try {
        torch::Tensor R = quatToRotMat(tf(q4, {4}));
        out_f(R, R9);
        return 0;
    } catch (const std::exception &e) { g_train_err = e.what(); return -1; }
}

extern "C" int ref_colmap_pose(const double *q, const double *t, float *pose16) {
    try {
        torch::Tensor qVec = torch::tensor({q[0], q[1], q[2], q[3]}, torch::kFloat32);
        torch::Tensor R = quatToRotMat(qVec);
        torch::Tensor T = torch::tensor({{t[0]}, {t[1]}, {t[2]}}, torch::kFloat32);
        torch::Tensor Rinv = R.transpose(0, 1);
        torch::Tensor Tinv = torch::matmul(-Rinv, T);
        torch::Tensor pose = torch::zeros({4, 4}, torch::kFloat32);
        pose.index_put_({Slice(None, 3), Slice(None, 3)}, Rinv);
        pose.index_put_({Slice(None, 3), Slice(3, 4)}, Tinv);
        pose[3][3] = 1.0f;
        // Convert COLMAP's camera CRS (OpenCV) to OpenGL
        pose.index_put_({Slice(0, 3), Slice(1, 3)}, pose.index({Slice(0, 3), Slice(1, 3)}) * -1.0f);
        out_f(pose, pose16);
        return 0;
    } catch (const std::exception &e) { g_train_err = e.what(); return -1; }
}

extern "C" int ref_auto_scale_and_center(int n, const float *poses_in, float *poses_out, float *center3,
                                         float *scale) {
    try {
        torch::Tensor poses = tf(poses_in, {n, 4, 4});
        auto r = autoScaleAndCenterPoses(poses);
        out_f(std::get<0>(r), poses_out);
        out_f(std::get<1>(r), center3);
        *scale = std::get<2>(r);
        return 0;
    } catch (const std::exception &e) { g_train_err = e.what(); return -1; }
}

// model.cpp:85-113: out8 = {fx, fy, cx, cy, height, width, fovX, fovY}; projview = projMat @ viewMat (:152)
extern "C" int ref_render_camera(const float *camToWorld16, float camFx, float camFy, float camCx, float camCy,
                                 int camHeight, int camWidth, float scaleFactor, float *view16,
                                 float *projview16, float *out8) {
    try {
        torch::Tensor camToWorld = tf(camToWorld16, {4, 4});
        const float fx = camFx / scaleFactor;
        const float fy = camFy / scaleFactor;
        const float cx = camCx / scaleFactor;
        const float cy = camCy / scaleFactor;
        const int height = static_cast<int>(static_cast<float>(camHeight) / scaleFactor);
        const int width = static_cast<int>(static_cast<float>(camWidth) / scaleFactor);
        torch::Tensor R = camToWorld.index({Slice(None, 3), Slice(None, 3)});
        torch::Tensor T = camToWorld.index({Slice(None, 3), Slice(3, 4)});
        R = torch::matmul(R, torch::diag(torch::tensor({1.0f, -1.0f, -1.0f})));
        torch::Tensor Rinv = R.transpose(0, 1);
        torch::Tensor Tinv = torch::matmul(-Rinv, T);
        torch::Tensor viewMat = torch::eye(4);
        viewMat.index_put_({Slice(None, 3), Slice(None, 3)}, Rinv);
        viewMat.index_put_({Slice(None, 3), Slice(3, 4)}, Tinv);
        float fovX = 2.0f * std::atan(width / (2.0f * fx));
        float fovY = 2.0f * std::atan(height / (2.0f * fy));
        const float zNear = 0.001f, zFar = 1000.0f;   // projectionMatrix, model.cpp:35-47
        float t = zNear * std::tan(0.5f * fovY);
        float b = -t;
        float r = zNear * std::tan(0.5f * fovX);
        float l = -r;
        torch::Tensor projMat = torch::tensor({
            {2.0f * zNear / (r - l), 0.0f, (r + l) / (r - l), 0.0f},
            {0.0f, 2 * zNear / (t - b), (t + b) / (t - b), 0.0f},
            {0.0f, 0.0f, (zFar + zNear) / (zFar - zNear), -1.0f * zFar * zNear / (zFar - zNear)},
            {0.0f, 0.0f, 1.0f, 0.0f}});
        out_f(viewMat, view16);
        out_f(torch::matmul(projMat, viewMat), projview16);
        const float o[8] = {fx, fy, cx, cy, (float)height, (float)width, fovX, fovY};
        std::memcpy(out8, o, sizeof(o));
        return 0;
    } catch (const std::exception &e) { g_train_err = e.what(); return -1; }
}

// model.hpp:37,41 + model.cpp:23-33, and rgb2sh (spherical_harmonics.cpp, compiled) of model.hpp:46
extern "C" int ref_model_init(int n, const uint8_t *rgb, float *quats_out, float *features_dc_out,
                              float *opacity_out) {
    try {
        torch::manual_seed(42);
        torch::Tensor u = torch::rand(n);
        torch::Tensor v = torch::rand(n);
        torch::Tensor w = torch::rand(n);
        torch::Tensor quats = torch::stack({torch::sqrt(1 - u) * torch::sin(2 * PI * v),
                                            torch::sqrt(1 - u) * torch::cos(2 * PI * v),
                                            torch::sqrt(u) * torch::sin(2 * PI * w),
                                            torch::sqrt(u) * torch::cos(2 * PI * w)}, -1);
        torch::Tensor colors = torch::from_blob(const_cast<uint8_t *>(rgb), {n, 3}, torch::kUInt8).clone();
        torch::Tensor dc = rgb2sh(colors.toType(torch::kFloat64) / 255.0).toType(torch::kFloat32);
        torch::Tensor op = torch::logit(0.1f * torch::ones({n, 1}));
        out_f(quats, quats_out);
        out_f(dc, features_dc_out);
        out_f(op, opacity_out);
        return 0;
    } catch (const std::exception &e) { g_train_err = e.what(); return -1; }
}
