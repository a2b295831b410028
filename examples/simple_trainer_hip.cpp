// simple_trainer_hip.cpp — BASELINE config 1 (the set-up of OpenSplat's simple_trainer.cpp:79-192) written
// against THIS repository's libtorch operators: it shows that a C++ caller compiles against
// opensplat_amd/csrc/gsplat_ops.hpp with the reference's call shapes (ProjectGaussians::apply ->
// RasterizeGaussians::apply, the GPU branch of simple_trainer.cpp:173-192) and links libgsplat_torch.so
// + libgsplat_hip.so only.  Not a copy of the reference program: same experiment, own code.
//
//   simple_trainer_hip [--points N] [--width W] [--height H] [--iters K] [--lr LR] [--fused-adam]
//
// Prints "iter <i> loss <mse>" lines and a final JSON summary (iterations/s measured after warm-up).
#include <torch/torch.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "gsplat_ops.hpp"

static const double kPi = 3.14159265358979323846;

int main(int argc, char **argv) {
    gsplatCheckAbi();   // libgsplat_hip.so / libgsplat_torch.so built from the same header
    int64_t numPoints = 10000, width = 256, height = 256, iterations = 200;
    double lr = 0.01;
    bool fusedAdam = false;
    for (int i = 1; i < argc; i++) {
        auto is = [&](const char *f) { return std::strcmp(argv[i], f) == 0; };
        if (is("--points") && i + 1 < argc) numPoints = std::atoll(argv[++i]);
        else if (is("--width") && i + 1 < argc) width = std::atoll(argv[++i]);
        else if (is("--height") && i + 1 < argc) height = std::atoll(argv[++i]);
        else if (is("--iters") && i + 1 < argc) iterations = std::atoll(argv[++i]);
        else if (is("--lr") && i + 1 < argc) lr = std::atof(argv[++i]);
        else if (is("--fused-adam")) fusedAdam = true;
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (!torch::cuda::is_available()) {
        std::fprintf(stderr, "no GPU: this program has no CPU path\n");
        return 1;
    }
    const torch::Device dev(torch::kCUDA, 0);

    // target: white, top-left quadrant red, bottom-right quadrant blue
    using torch::indexing::Slice;
    torch::Tensor target = torch::ones({height, width, 3});
    target.index_put_({Slice(0, height / 2), Slice(0, width / 2)}, torch::tensor({1.0f, 0.0f, 0.0f}));
    target.index_put_({Slice(height / 2, height), Slice(width / 2, width)}, torch::tensor({0.0f, 0.0f, 1.0f}));
    target = target.to(dev);
    const double focal = 0.5 * (double)width / std::tan(0.25 * kPi);   // 90 degree horizontal fov

    // the reference's draw order on the CPU generator: means, scales, colours, then u, v, w
    torch::manual_seed(0);
    torch::Tensor means = (2.0 * (torch::rand({numPoints, 3}) - 0.5)).to(dev);
    torch::Tensor scales = torch::rand({numPoints, 3}).to(dev);
    torch::Tensor rgbs = torch::rand({numPoints, 3}).to(dev);
    torch::Tensor u = torch::rand({numPoints, 1}).to(dev), v = torch::rand({numPoints, 1}).to(dev),
                  w = torch::rand({numPoints, 1}).to(dev);
    torch::Tensor quats = torch::cat({torch::sqrt(1.0 - u) * torch::sin(2.0 * kPi * v),
                                      torch::sqrt(1.0 - u) * torch::cos(2.0 * kPi * v),
                                      torch::sqrt(u) * torch::sin(2.0 * kPi * w),
                                      torch::sqrt(u) * torch::cos(2.0 * kPi * w)}, -1);
    torch::Tensor opacities = torch::ones({numPoints, 1}, dev);
    torch::Tensor viewMat = torch::eye(4, dev);
    viewMat[2][3] = 8.0f;                                    // camera 8 units back along z
    torch::Tensor background = torch::zeros({3}, dev);
    for (torch::Tensor *t : {&means, &scales, &quats, &rgbs, &opacities}) t->requires_grad_();

    const TileBounds tileBounds = std::make_tuple((int)((width + BLOCK_X - 1) / BLOCK_X),
                                                  (int)((height + BLOCK_Y - 1) / BLOCK_Y), 1);
    torch::optim::Adam adam({rgbs, means, scales, opacities, quats}, torch::optim::AdamOptions(lr));
    FusedAdam fused({rgbs, means, scales, opacities, quats}, {lr, lr, lr, lr, lr});

    double firstLoss = 0.0, lastLoss = 0.0;
    const int64_t warm = std::min<int64_t>(10, iterations / 2);
    std::chrono::steady_clock::time_point t0;
    for (int64_t it = 0; it < iterations; it++) {
        if (it == warm) {
            torch::cuda::synchronize();
            t0 = std::chrono::steady_clock::now();
        }
        auto p = ProjectGaussians::apply(means, scales, 1.0f, quats, viewMat, viewMat, (float)focal,
                                         (float)focal, (float)(width / 2), (float)(height / 2), (int)height,
                                         (int)width, tileBounds);
        torch::Tensor img = RasterizeGaussians::apply(p[0], p[1], p[2], p[3], p[4], torch::sigmoid(rgbs),
                                                      torch::sigmoid(opacities), (int)height, (int)width,
                                                      background, p[6] /* cov2d: gsplat-cpu rectangles */);
        torch::Tensor loss = torch::mse_loss(img, target);
        if (fusedAdam) fused.zeroGrad(); else adam.zero_grad();
        loss.backward();
        if (fusedAdam) fused.step(); else adam.step();
        if (it == 0 || (it + 1) % 10 == 0 || it + 1 == iterations) {
            lastLoss = loss.item<double>();
            if (it == 0) firstLoss = lastLoss;
            std::printf("iter %lld loss %.9f\n", (long long)(it + 1), lastLoss);
        }
    }
    torch::cuda::synchronize();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("{\"points\": %lld, \"width\": %lld, \"height\": %lld, \"iterations\": %lld, \"optimizer\": \"%s\", "
                "\"first_loss\": %.9f, \"last_loss\": %.9f, \"iterations_per_s\": %.1f}\n",
                (long long)numPoints, (long long)width, (long long)height, (long long)iterations,
                fusedAdam ? "FusedAdam" : "torch::optim::Adam", firstLoss, lastLoss,
                (double)(iterations - warm) / secs);
    return 0;
}
