// model_fused.inl — OpenSplat's `Model` (model.hpp / model.cpp) on the FUSED MI355X operators.
//
// Included by a model.cpp patched with `integration/apply_hip_native.py --fused` (macro
// USE_HIP_NATIVE_FUSED); the patch adds four call sites, nothing else:
//
//   Model::forward      model.cpp:105-222  -> gs_fused::render          camera matrices kept on the host;
//                                                                      one SplatRender node: cat, exp,
//                                                                      normalise, view directions, SH,
//                                                                      +0.5 / clamp, sigmoid, binning,
//                                                                      compositing, clamp_max
//   Model::mainLoss     model.cpp:780-784  -> ::mainLoss               fused L1 + SSIM with its backward
//   Model::optimizersStep  :236-243        -> gs_fused::optimizers_step  six Adam updates in ONE launch
//   Model::afterTrain   model.cpp:311-494  -> gs_fused::after_train     statistics, densification and
//                                                                      culling on the device
//
// The six torch::optim::Adam objects stay what they are — the owners of exp_avg / exp_avg_sq / step and
// of the learning rates (OptimScheduler keeps working on meansOpt, --resume and the reference's own
// addToOptimizer / removeFromOptimizer too): only the arithmetic moves into kernels.  Everything is a
// template on the Model type, so this file needs no OpenSplat header itself.
//
// GS_FUSED_REFERENCE_ALPHA_RESET (macro, default off): reproduce what the reference's alpha reset
// actually does (model.cpp:464-479: `opacities` re-bound to a clamped copy the optimiser does not know
// until the next refinement re-registers it, its moments left alone) instead of the evident intent
// (clamp the registered parameter in place and zero its moments) — DESIGN.md §12.
#pragma once

#include <c10/hip/HIPCachingAllocator.h>

#include <algorithm>
#include <cmath>
#include <memory>
#include <vector>

#include "gsplat_ops.hpp"

namespace gs_fused {

using torch::indexing::Slice;

// ---- Model::forward from the camera matrices on (model.cpp:105-222) --------------------------------
// Rinv [3,3], Tinv [3,1]: world -> camera (model.cpp:100-101); T [3,1]: the camera centre in world space;
// projMat [4,4] = projectionMatrix() (NOT yet multiplied) — all HOST tensors, as Model::forward has
// them at that point.  They stay on the host: the operators take 4x4 matrices and float[3] arguments by
// value, so an iteration issues no host-to-device copy and never waits for the device (the reference
// builds viewMat / projMat on the device — three synchronising copies — and reads radii.sum() back,
// model.cpp:105-113,173).
template <class M>
inline torch::Tensor render(M &m, const torch::Tensor &Rinv, const torch::Tensor &Tinv, const torch::Tensor &T,
                            const torch::Tensor &projMat, float fx, float fy, float cx, float cy, int height,
                            int width, int step) {
    const int degreesToUse = (std::min<int>)(step / m.shDegreeInterval, m.shDegree);
    const int64_t N = m.means.size(0);
    torch::Tensor viewMat = torch::eye(4);                                   // model.cpp:105-107, on the host
    viewMat.index_put_({Slice(torch::indexing::None, 3), Slice(torch::indexing::None, 3)}, Rinv);
    viewMat.index_put_({Slice(torch::indexing::None, 3), Slice(3, 4)}, Tinv);
    // d loss / d xys, filled by the node's backward: Model::afterTrain reads it as xys.grad() (:318)
    torch::Tensor xysGrad = torch::zeros({N, 2}, m.means.options().requires_grad(false));
    auto out = SplatRender::apply(m.means, m.scales, m.quats, m.opacities, m.featuresDc, m.featuresRest,
                                  viewMat, torch::matmul(projMat, viewMat), T.transpose(0, 1).contiguous(), fx, fy,
                                  cx, cy, height, width, degreesToUse, m.backgroundColor.detach(), xysGrad);
    m.radii = out[2];
    m.xys = out[1].detach().requires_grad_();
    // model.cpp:173-174 reads radii.sum() back to return the bare background when nothing is visible — a
    // tensor without graph, on which the training loop's backward() throws.  Not mirrored: such a frame
    // renders the background here too (no Gaussian composited), with zero gradients, and costs no
    // device synchronisation on every other frame
    m.xys.mutable_grad() = xysGrad;
    return out[0];
}

// ---- Model::optimizersStep (model.cpp:236-243): six torch::optim::Adam steps as one launch ----------
namespace detail {
inline torch::optim::AdamParamState &adam_state(torch::optim::Adam *opt, const torch::Tensor &p) {
    auto &state = opt->state();
    auto key = p.unsafeGetTensorImpl();
    if (state.find(key) == state.end()) {   // first step: what torch::optim::Adam::step() creates lazily
        auto s = std::make_unique<torch::optim::AdamParamState>();
        s->step(0);
        s->exp_avg(torch::zeros_like(p, torch::MemoryFormat::Preserve));
        s->exp_avg_sq(torch::zeros_like(p, torch::MemoryFormat::Preserve));
        state[key] = std::move(s);
    }
    return static_cast<torch::optim::AdamParamState &>(*state[key]);
}
template <class M>
inline std::vector<torch::optim::Adam *> optimizers(M &m) {   // in the order of the parameter sets below
    return {m.meansOpt, m.scalesOpt, m.quatsOpt, m.opacitiesOpt, m.featuresDcOpt, m.featuresRestOpt};
}
}  // namespace detail

template <class M>
inline void optimizers_step(M &m) {
    torch::NoGradGuard noGrad;
    // one launch takes one step count.  The six optimisers of a Model step together, so there is one
    // launch — except after the reference's own alpha reset (GS_FUSED_REFERENCE_ALPHA_RESET), which
    // leaves opacitiesOpt some steps behind the others for good (its parameter is unknown to it until
    // the next refinement): groups are launched by step count
    struct Batch {
        int64_t step;
        std::vector<torch::Tensor> params, grads, expAvg, expAvgSq;
        std::vector<double> lrs;
    };
    std::vector<Batch> batches;
    for (torch::optim::Adam *opt : detail::optimizers(m)) {
        torch::Tensor p = opt->param_groups()[0].params()[0];
        if (!p.grad().defined() || p.numel() == 0) continue;   // (what Adam::step skips)
        auto &s = detail::adam_state(opt, p);
        s.step(s.step() + 1);
        auto b = std::find_if(batches.begin(), batches.end(), [&](const Batch &x) { return x.step == s.step(); });
        if (b == batches.end()) {
            batches.push_back(Batch{s.step(), {}, {}, {}, {}, {}});
            b = batches.end() - 1;
        }
        b->params.push_back(p);
        b->grads.push_back(p.grad().contiguous());
        b->expAvg.push_back(s.exp_avg());
        b->expAvgSq.push_back(s.exp_avg_sq());
        b->lrs.push_back(static_cast<torch::optim::AdamOptions &>(opt->param_groups()[0].options()).lr());
    }
    for (const Batch &b : batches) fusedAdamStep(b.params, b.grads, b.expAvg, b.expAvgSq, b.lrs, b.step);
}

// ---- Model::afterTrain (model.cpp:311-494) ----------------------------------------------------------------
template <class M>
inline void after_train(M &m, int step) {
    torch::NoGradGuard noGrad;
    if (!m.xys.grad().defined()) return;                                   // :315
    if (step < m.stopSplitAt)                                              // :317-337, one kernel, no sync
        densifyStats(m.xys.grad(), m.radii, m.lastHeight, m.lastWidth, m.xysGradNorm, m.visCounts, m.max2DSize);
    if (!(step % m.refineEvery == 0 && step > m.warmupLength)) return;    // :339
    const int resetInterval = m.resetAlphaEvery * m.refineEvery;
    const bool doDensification = step < m.stopSplitAt && step % resetInterval > m.numCameras + m.refineEvery;
    torch::Tensor *fields[6] = {&m.means, &m.scales, &m.quats, &m.opacities, &m.featuresDc, &m.featuresRest};
    auto opts = detail::optimizers(m);
    if (doDensification) {                                                 // :345-458
        std::vector<torch::Tensor> params, expAvg, expAvgSq;
        std::vector<int64_t> steps;
        for (int i = 0; i < 6; i++) {
            torch::Tensor p = opts[i]->param_groups()[0].params()[0];
            auto &s = detail::adam_state(opts[i], p);
            params.push_back(fields[i]->detach().contiguous());
            expAvg.push_back(s.exp_avg().contiguous());
            expAvgSq.push_back(s.exp_avg_sq().contiguous());
            steps.push_back(s.step());
        }
        const int64_t before = m.means.size(0);
        DensifyResult r = densify(params, expAvg, expAvgSq, m.xysGradNorm, m.visCounts, m.max2DSize, m.lastWidth,
                                  m.lastHeight, m.densifyGradThresh, m.densifySizeThresh,
                                  step < m.stopScreenSizeAt, m.splitScreenSize,
                                  step > m.refineEvery * m.resetAlphaEvery);
        for (int i = 0; i < 6; i++) {   // addToOptimizer / removeFromOptimizer (:253-309) in one go
            torch::Tensor old = opts[i]->param_groups()[0].params()[0];
            opts[i]->state().erase(old.unsafeGetTensorImpl());
            *fields[i] = r.params[i].requires_grad_();
            auto s = std::make_unique<torch::optim::AdamParamState>();
            s->step(steps[i]);
            s->exp_avg(r.expAvg[i]);
            s->exp_avg_sq(r.expAvgSq[i]);
            opts[i]->state()[fields[i]->unsafeGetTensorImpl()] = std::move(s);
            opts[i]->param_groups()[0].params()[0] = *fields[i];
        }
        std::cout << "Added " << r.added << " gaussians, culled " << r.culled << " (" << before << " -> "
                  << m.means.size(0) << ")" << std::endl;
    }
    if (step < m.stopSplitAt && step % resetInterval == m.refineEvery) {   // :464-479
        const float resetValue = 0.1f * 2.0f;
#ifdef GS_FUSED_REFERENCE_ALPHA_RESET
        // what the reference does: a clamped COPY (not a leaf, unknown to opacitiesOpt), moments untouched
        m.opacities = torch::clamp_max(m.opacities, std::log(resetValue / (1.0f - resetValue)));
#else
        // the evident intent: the registered parameter clamped in place, its moments zeroed
        torch::Tensor p = opts[3]->param_groups()[0].params()[0];
        auto &s = detail::adam_state(opts[3], p);
        torch::Tensor ea = s.exp_avg(), es = s.exp_avg_sq();
        resetOpacity(p, resetValue, ea, es);
#endif
        std::cout << "Alpha reset" << std::endl;
    }
    m.xysGradNorm = torch::Tensor();                                       // :482-484
    m.visCounts = torch::Tensor();
    m.max2DSize = torch::Tensor();
    c10::hip::HIPCachingAllocator::emptyCache();                           // :486-492
}

}  // namespace gs_fused
