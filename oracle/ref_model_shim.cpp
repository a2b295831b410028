// ref_model_shim.cpp — TEST INFRASTRUCTURE ONLY.
//
// C-ABI handle on the reference's OWN `Model` (model.hpp / model.cpp, compiled where they lie under
// /root/reference by oracle/Makefile — never copied).  model.cpp names no OpenCV / nanoflann / json
// type itself; the three headers it reaches through its includes are satisfied by declaration-only
// stand-ins (oracle/stubs/), so Model::forward (CPU branch: gsplat-cpu), mainLoss, optimizersStep,
// schedulersStep, afterTrain, savePly and saveSplat are the reference's compiled code, not a
// restatement.  What IS ours here, because kdtree_tensor.cpp needs nanoflann: PointsTensor::scales()
// by brute force (kdtree_tensor.cpp:4-24: mean distance to the three nearest neighbours) and the
// destructor.
//
// Round 2 pinned row f4 to `afterTrain`'s statements restated as free functions
// (ref_train_shim.cpp:117-300); tests now pin the densification, the alpha reset and the two file
// formats to this compiled Model as well.
#include <cstdint>
#include <cstring>
#include <string>

#include "model.hpp"

// ---- kdtree_tensor.cpp:4-28 by brute force --------------------------------------------------------
torch::Tensor PointsTensor::scales() {
    const int64_t n = tensor.size(0);
    torch::Tensor scales = torch::zeros({n, 1}, torch::kFloat32);
    const int count = 4;
    torch::Tensor d2 = torch::cdist(tensor, tensor).pow(2);   // squared L2, like L2_Simple_Adaptor
    for (int64_t i = 0; i < n; i++) {
        torch::Tensor row = std::get<0>(torch::topk(d2[i], std::min<int64_t>(count, n), -1, /*largest=*/false));
        float sum = 0.0;
        for (int64_t j = 1; j < row.size(0); j++) sum += std::sqrt(row[j].item<float>());
        scales[i] = sum / (count - 1);
    }
    return scales;
}
PointsTensor::~PointsTensor() {}

namespace {
thread_local std::string g_err;
torch::Tensor tf(const float *p, std::vector<int64_t> shape) {
    return torch::from_blob(const_cast<float *>(p), shape, torch::kFloat32).clone();
}
void out_f(const torch::Tensor &t, float *dst) {
    if (!dst) return;
    torch::Tensor c = t.detach().to(torch::kFloat32).contiguous();
    std::memcpy(dst, c.data_ptr<float>(), sizeof(float) * c.numel());
}
torch::optim::AdamParamState &state_of(torch::optim::Adam *opt) {
    torch::Tensor param = opt->param_groups()[0].params()[0];
    return static_cast<torch::optim::AdamParamState &>(*opt->state()[param.unsafeGetTensorImpl()]);
}
void inject(torch::optim::Adam *opt, const torch::Tensor &ea, const torch::Tensor &es, int64_t step) {
    torch::Tensor param = opt->param_groups()[0].params()[0];
    auto st = std::make_unique<torch::optim::AdamParamState>();
    st->step(step);
    st->exp_avg(ea);
    st->exp_avg_sq(es);
    opt->state()[param.unsafeGetTensorImpl()] = std::move(st);
}
}  // namespace

extern "C" const char *refm_last_error() { return g_err.c_str(); }

// Model(inputData, numCameras, numDownscales, resolutionSchedule, shDegree, shDegreeInterval, refineEvery,
//       warmupLength, resetAlphaEvery, densifyGradThresh, densifySizeThresh, stopScreenSizeAt,
//       splitScreenSize, maxSteps, keepCrs, device = CPU), model.hpp:23-62.
extern "C" void *refm_create(int n_points, const float *xyz, const uint8_t *rgb, int numCameras,
                             int numDownscales, int resolutionSchedule, int shDegree, int shDegreeInterval,
                             int refineEvery, int warmupLength, int resetAlphaEvery,
                             float densifyGradThresh, float densifySizeThresh, int stopScreenSizeAt,
                             float splitScreenSize, int maxSteps, int keepCrs, float scale,
                             const float *translation3) {
    try {
        InputData in;
        in.scale = scale;
        in.translation = translation3 ? tf(translation3, {3}) : torch::zeros({3});
        in.points.xyz = tf(xyz, {n_points, 3});
        in.points.rgb = torch::from_blob(const_cast<uint8_t *>(rgb), {n_points, 3}, torch::kUInt8).clone();
        return new Model(in, numCameras, numDownscales, resolutionSchedule, shDegree, shDegreeInterval,
                         refineEvery, warmupLength, resetAlphaEvery, densifyGradThresh, densifySizeThresh,
                         stopScreenSizeAt, splitScreenSize, maxSteps, keepCrs != 0, torch::kCPU);
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}
extern "C" void refm_destroy(void *m) { delete static_cast<Model *>(m); }

extern "C" int refm_num_points(void *m) { return (int)static_cast<Model *>(m)->means.size(0); }
extern "C" int refm_sh_bases(void *m) { return 1 + (int)static_cast<Model *>(m)->featuresRest.size(1); }

// Replace the six parameter tensors (means, scales(log), quats, opacities(logit) [N,1], featuresDc,
// featuresRest [N,K-1,3]) and, optionally, the Adam state of each (exp_avg, exp_avg_sq, step count).
extern "C" int refm_set_state(void *mp, int N, int K, const float *const *params, const float *const *exp_avg,
                              const float *const *exp_avg_sq, int64_t adam_step) {
    try {
        Model &m = *static_cast<Model *>(mp);
        const std::vector<std::vector<int64_t>> shapes = {{N, 3}, {N, 3}, {N, 4}, {N, 1}, {N, 3}, {N, K - 1, 3}};
        m.means = tf(params[0], shapes[0]).requires_grad_();
        m.scales = tf(params[1], shapes[1]).requires_grad_();
        m.quats = tf(params[2], shapes[2]).requires_grad_();
        m.opacities = tf(params[3], shapes[3]).requires_grad_();
        m.featuresDc = tf(params[4], shapes[4]).requires_grad_();
        m.featuresRest = tf(params[5], shapes[5]).requires_grad_();
        m.releaseOptimizers();
        m.setupOptimizers();
        if (exp_avg && exp_avg_sq) {
            torch::optim::Adam *opts[6] = {m.meansOpt, m.scalesOpt, m.quatsOpt, m.opacitiesOpt, m.featuresDcOpt,
                                           m.featuresRestOpt};
            for (int i = 0; i < 6; i++) inject(opts[i], tf(exp_avg[i], shapes[i]), tf(exp_avg_sq[i], shapes[i]), adam_step);
        }
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// out arrays must hold refm_num_points rows; moments may be NULL
extern "C" int refm_get_state(void *mp, float *const *params, float *const *exp_avg, float *const *exp_avg_sq) {
    try {
        Model &m = *static_cast<Model *>(mp);
        const torch::Tensor ps[6] = {m.means, m.scales, m.quats, m.opacities, m.featuresDc, m.featuresRest};
        torch::optim::Adam *opts[6] = {m.meansOpt, m.scalesOpt, m.quatsOpt, m.opacitiesOpt, m.featuresDcOpt,
                                       m.featuresRestOpt};
        for (int i = 0; i < 6; i++) {
            if (params) out_f(ps[i], params[i]);
            if (exp_avg && exp_avg[i]) out_f(state_of(opts[i]).exp_avg(), exp_avg[i]);
            if (exp_avg_sq && exp_avg_sq[i]) out_f(state_of(opts[i]).exp_avg_sq(), exp_avg_sq[i]);
        }
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// Model::afterTrain(step) (model.cpp:311-494) on given inputs: the statistics accumulated so far
// (NULL: none yet), this iteration's radii and d loss / d xys, the image size of the last forward.
// torch::manual_seed(seed) first: afterTrain draws its split samples with torch::randn.
// stats_out (nullable, 3 x N floats): the statistics AFTER the call when they survive it (no refinement
// happened); *refined = 1 when the call cleared them.
extern "C" int refm_after_train(void *mp, int step, int N, const int32_t *radii, const float *xys_grad,
                                const float *xysGradNorm, const float *visCounts, const float *max2DSize,
                                int lastHeight, int lastWidth, uint64_t seed, float *stats_out, int *refined) {
    try {
        Model &m = *static_cast<Model *>(mp);
        m.radii = torch::from_blob(const_cast<int32_t *>(radii), {N}, torch::kInt32).clone();
        m.xys = torch::zeros({N, 2}).requires_grad_();
        if (xys_grad) m.xys.mutable_grad() = tf(xys_grad, {N, 2});
        m.xysGradNorm = xysGradNorm ? tf(xysGradNorm, {N}) : torch::Tensor();
        m.visCounts = visCounts ? tf(visCounts, {N}) : torch::Tensor();
        m.max2DSize = max2DSize ? tf(max2DSize, {N}) : torch::Tensor();
        m.lastHeight = lastHeight;
        m.lastWidth = lastWidth;
        torch::manual_seed(seed);
        m.afterTrain(step);
        const bool cleared = !m.xysGradNorm.numel();
        if (refined) *refined = cleared ? 1 : 0;
        if (stats_out && !cleared) {
            out_f(m.xysGradNorm, stats_out);
            out_f(m.visCounts, stats_out + N);
            out_f(m.max2DSize, stats_out + 2 * (size_t)N);
        }
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

extern "C" int refm_save(void *mp, const char *filename, int step) {   // .ply / .splat by extension
    try {
        static_cast<Model *>(mp)->save(filename, step);
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// One iteration of opensplat.cpp:151-170 on the CPU with the reference's own Model:
//   rgb = model.forward(cam, step); loss = model.mainLoss(rgb, gt, ssimWeight); loss.backward();
//   model.optimizersStep(); model.schedulersStep(step); model.afterTrain(step);
// cam: width, height, fx, fy, cx, cy and camToWorld (4 x 4 row-major); gt [H, W, 3] at the resolution
// Model::forward renders at this step (height / getDownscaleFactor(step), model.cpp:85-92).
// rgb_out (nullable) receives the rendered image, xys_grad_out / radii_out (nullable, N rows) what
// afterTrain consumed.  Returns the loss in *loss.
extern "C" int refm_train_iteration(void *mp, int step, int width, int height, float fx, float fy, float cx,
                                    float cy, const float *camToWorld16, const float *gt, float ssimWeight,
                                    uint64_t seed, float *loss_out, float *rgb_out, float *xys_grad_out,
                                    int32_t *radii_out, int run_after_train) {
    try {
        Model &m = *static_cast<Model *>(mp);
        Camera cam(width, height, fx, fy, cx, cy, 0, 0, 0, 0, 0, tf(camToWorld16, {4, 4}), "");
        m.optimizersZeroGrad();
        torch::Tensor rgb = m.forward(cam, step);
        torch::Tensor gtT = tf(gt, {rgb.size(0), rgb.size(1), 3});
        torch::Tensor loss = m.mainLoss(rgb, gtT, ssimWeight);
        loss.backward();
        if (loss_out) *loss_out = loss.item<float>();
        if (rgb_out) out_f(rgb, rgb_out);
        if (xys_grad_out && m.xys.grad().defined()) out_f(m.xys.grad(), xys_grad_out);
        if (radii_out) {
            torch::Tensor r = m.radii.to(torch::kInt32).contiguous();
            std::memcpy(radii_out, r.data_ptr<int32_t>(), sizeof(int32_t) * r.numel());
        }
        m.optimizersStep();
        m.schedulersStep(step);
        if (run_after_train) {
            torch::manual_seed(seed);
            m.afterTrain(step);
        }
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

extern "C" int refm_downscale_factor(void *mp, int step) { return static_cast<Model *>(mp)->getDownscaleFactor(step); }
extern "C" float refm_means_lr(void *mp) {
    Model &m = *static_cast<Model *>(mp);
    return (float)static_cast<torch::optim::AdamOptions &>(m.meansOpt->param_groups()[0].options()).lr();
}
