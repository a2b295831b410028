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
