// model_fused_tu.cpp — trains with OpenSplat's OWN `Model` (model.hpp + model.cpp patched by
// `integration/apply_hip_native.py --fused`) at the opensplat.cpp:151-170 level:
//
//     rgb = model.forward(cam, step); loss = model.mainLoss(rgb, gt, ssimWeight); loss.backward();
//     model.optimizersStep(); model.schedulersStep(step); model.afterTrain(step);
//
// on `--device gpu` every one of those calls lands in the fused MI355X operators (model_fused.inl:
// SplatRender, MainLoss, one-launch Adam, device-side densification), on `--device cpu` the same binary
// runs the reference's original statements (gsplat-cpu) — the patch leaves the CPU device alone.
// Test infrastructure (tests/test_gpu_model_fused.py); written for this repo, nothing copied.
//
//   model_fused_shim <case.bin> <out.bin> --device cpu|gpu
//
// Array file: repeated { u32 name_len, name, u32 ndim, i64 dims[ndim], f32 data[] }.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <string>

#include "model.hpp"

// kdtree_tensor.cpp needs nanoflann: Model's constructor gets the same brute-force stand-in as
// oracle/ref_model_shim.cpp (the test overrides the constructor's tensors anyway)
torch::Tensor PointsTensor::scales() {
    torch::NoGradGuard noGrad;   // (on the CPU device Model's `means` IS this tensor, already a leaf that requires grad)
    const int64_t n = tensor.size(0);
    torch::Tensor d = torch::cdist(tensor, tensor);
    torch::Tensor near = std::get<0>(torch::topk(d, std::min<int64_t>(4, n), -1, false));
    return near.index({torch::indexing::Slice(), torch::indexing::Slice(1, torch::indexing::None)}).sum(-1, true) / 3.0f;
}
PointsTensor::~PointsTensor() {}

static std::map<std::string, torch::Tensor> read_arrays(const std::string &path) {
    std::map<std::string, torch::Tensor> out;
    std::ifstream f(path, std::ios::binary);
    TORCH_CHECK(f.good(), "cannot open ", path);
    for (;;) {
        uint32_t nl = 0;
        if (!f.read(reinterpret_cast<char *>(&nl), 4)) break;
        std::string name(nl, '\0');
        f.read(&name[0], nl);
        uint32_t nd = 0;
        f.read(reinterpret_cast<char *>(&nd), 4);
        std::vector<int64_t> dims(nd);
        f.read(reinterpret_cast<char *>(dims.data()), 8 * nd);
        torch::Tensor t = torch::empty(dims, torch::kFloat32);
        f.read(reinterpret_cast<char *>(t.data_ptr<float>()), 4 * t.numel());
        out[name] = t;
    }
    return out;
}
static void write_array(std::ofstream &f, const std::string &name, const torch::Tensor &t_) {
    torch::Tensor t = t_.detach().to(torch::kCPU, torch::kFloat32).contiguous();
    const uint32_t nl = (uint32_t)name.size(), nd = (uint32_t)t.dim();
    f.write(reinterpret_cast<const char *>(&nl), 4);
    f.write(name.data(), nl);
    f.write(reinterpret_cast<const char *>(&nd), 4);
    for (int64_t d : t.sizes()) f.write(reinterpret_cast<const char *>(&d), 8);
    f.write(reinterpret_cast<const char *>(t.data_ptr<float>()), 4 * t.numel());
}

int main(int argc, char **argv) {
    if (argc < 5 || std::string(argv[3]) != "--device") {
        std::fprintf(stderr, "usage: model_fused_shim <case.bin> <out.bin> --device cpu|gpu\n");
        return 2;
    }
    const bool gpu = std::string(argv[4]) == "gpu";
    torch::Device device = gpu ? torch::Device(torch::kCUDA, 0) : torch::Device(torch::kCPU);
    auto in = read_arrays(argv[1]);
    auto cfg = in.at("cfg").to(torch::kInt32);
    auto ci = [&](int i) { return cfg[i].item<int>(); };
    auto cf = [&](int i) { return in.at("cfgf")[i].item<float>(); };
    const int numCameras = ci(0), numDownscales = ci(1), resolutionSchedule = ci(2), shDegree = ci(3),
              shDegreeInterval = ci(4), refineEvery = ci(5), warmupLength = ci(6), resetAlphaEvery = ci(7),
              stopScreenSizeAt = ci(8), maxSteps = ci(9), iters = ci(10), firstStep = ci(11), W = ci(12), H = ci(13);
    const float ssimWeight = cf(3);

    InputData data;
    data.scale = 1.0f;
    data.translation = torch::zeros({3});
    data.points.xyz = in.at("p0").clone();
    data.points.rgb = torch::zeros({data.points.xyz.size(0), 3}, torch::kUInt8);
    Model model(data, numCameras, numDownscales, resolutionSchedule, shDegree, shDegreeInterval, refineEvery,
                warmupLength, resetAlphaEvery, cf(0), cf(1), stopScreenSizeAt, cf(2), maxSteps, false, device);
    // the case's own starting point instead of the constructor's random quaternions
    model.means = in.at("p0").to(device).requires_grad_();
    model.scales = in.at("p1").to(device).requires_grad_();
    model.quats = in.at("p2").to(device).requires_grad_();
    model.opacities = in.at("p3").to(device).requires_grad_();
    model.featuresDc = in.at("p4").to(device).requires_grad_();
    model.featuresRest = in.at("p5").to(device).requires_grad_();
    for (const torch::Tensor *t : {&model.means, &model.scales, &model.quats, &model.opacities, &model.featuresDc,
                                   &model.featuresRest})
        TORCH_CHECK(t->is_leaf() && t->requires_grad(), "case parameters must be leaves");
    model.setupOptimizers();

    torch::Tensor cams = in.at("cams"), gts = in.at("gt").to(device);
    const int nCams = (int)cams.size(0);
    std::vector<Camera> cameras;
    for (int c = 0; c < nCams; c++) {
        auto r = cams[c];
        cameras.emplace_back(W, H, r[0].item<float>(), r[1].item<float>(), r[2].item<float>(), r[3].item<float>(),
                             0, 0, 0, 0, 0, r.slice(0, 4, 20).reshape({4, 4}).clone(), "");
    }
    torch::manual_seed(1234);   // afterTrain's split samples (device generator on the GPU)
    torch::Tensor losses = torch::zeros({iters});
    torch::Tensor counts = torch::zeros({iters});
    torch::Tensor lastRgb;
    std::vector<torch::Tensor> lossTensors;
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < iters; it++) {            // opensplat.cpp:151-170
        const int step = firstStep + it;
        Camera &cam = cameras[it % nCams];
        model.optimizersZeroGrad();
        torch::Tensor rgb = model.forward(cam, step);
        torch::Tensor gt = gts[it % nCams];
        torch::Tensor loss = model.mainLoss(rgb, gt, ssimWeight);
        loss.backward();
        model.optimizersStep();
        model.schedulersStep(step);
        model.afterTrain(step);
        lossTensors.push_back(loss.detach());   // read back after the loop: no per-iteration synchronisation
        counts[it] = (float)model.means.size(0);
        lastRgb = rgb.detach();
    }
    if (gpu) torch::cuda::synchronize();
    const double sec_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int it = 0; it < iters; it++) losses[it] = lossTensors[it].item<float>();
    const double sec = sec_;

    std::ofstream o(argv[2], std::ios::binary);
    write_array(o, "losses", losses);
    write_array(o, "counts", counts);
    write_array(o, "seconds", torch::tensor({(float)sec}));
    write_array(o, "rgb", lastRgb);
    const torch::Tensor ps[6] = {model.means, model.scales, model.quats, model.opacities, model.featuresDc,
                                 model.featuresRest};
    torch::optim::Adam *opts[6] = {model.meansOpt, model.scalesOpt, model.quatsOpt, model.opacitiesOpt,
                                   model.featuresDcOpt, model.featuresRestOpt};
    for (int i = 0; i < 6; i++) {
        write_array(o, "p" + std::to_string(i), ps[i]);
        torch::Tensor p = opts[i]->param_groups()[0].params()[0];
        auto it = opts[i]->state().find(p.unsafeGetTensorImpl());
        if (it != opts[i]->state().end()) {
            auto &s = static_cast<torch::optim::AdamParamState &>(*it->second);
            write_array(o, "m" + std::to_string(i), s.exp_avg());
            write_array(o, "v" + std::to_string(i), s.exp_avg_sq());
        }
    }
    if (model.xysGradNorm.numel()) {
        write_array(o, "xysGradNorm", model.xysGradNorm);
        write_array(o, "visCounts", model.visCounts);
        write_array(o, "max2DSize", model.max2DSize);
    }
    write_array(o, "means_lr", torch::tensor({(float)static_cast<torch::optim::AdamOptions &>(
                                   model.meansOpt->param_groups()[0].options()).lr()}));
    std::printf("{\"device\": \"%s\", \"iterations\": %d, \"seconds\": %.4f, \"final_loss\": %.6g, \"gaussians\": %d}\n",
                gpu ? "gpu" : "cpu", iters, sec, losses[iters - 1].item<float>(), (int)model.means.size(0));
    return 0;
}
