// TEST INFRASTRUCTURE ONLY — never linked into, imported by, or called from the product path.
//
// C-ABI shim around the *reference's own* CPU implementation (OpenSplat `rasterizer/gsplat-cpu`
// plus the three op wrappers), which `oracle/Makefile` compiles IN PLACE from /root/reference
// into `oracle/_ref/libgsplat_ref.so`.  No reference source is copied into this repository; this
// file only marshals raw host pointers into torch::Tensor views and calls:
//   project_gaussians_forward_tensor_cpu   rasterizer/gsplat-cpu/gsplat_cpu.cpp:48-131
//   rasterize_forward_tensor_cpu           rasterizer/gsplat-cpu/gsplat_cpu.cpp:137-257
//   rasterize_backward_tensor_cpu          rasterizer/gsplat-cpu/gsplat_cpu.cpp:260-376
//   compute_sh_forward_tensor_cpu          rasterizer/gsplat-cpu/gsplat_cpu.cpp:424-486
//   ProjectGaussiansCPU::apply             project_gaussians.cpp:94-123
//   RasterizeGaussiansCPU::apply           rasterize_gaussians.cpp:144-233
//   SphericalHarmonicsCPU::apply           spherical_harmonics.cpp:66-73
// It is used (a) to pin the plain-C restatement in oracle/gsplat_oracle.c, (b) to generate the
// golden fixtures under tests/golden/, (c) as bench.py's `cpu_baseline` (kind "reference").
//
// All pointers are HOST pointers to contiguous fp32 / int32 data.  Every function returns 0 on
// success and -1 if the reference threw (message via ref_last_error()).

#include <torch/torch.h>

#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "project_gaussians.hpp"
#include "rasterize_gaussians.hpp"
#include "spherical_harmonics.hpp"

namespace {

thread_local std::string g_err;

torch::Tensor f32(const float *p, std::initializer_list<int64_t> shape) {
    return torch::from_blob(const_cast<float *>(p), shape, torch::kFloat32).clone();
}

void put(const torch::Tensor &t, float *dst) {
    if (!dst) return;
    torch::Tensor c = t.detach().to(torch::kFloat32).contiguous();
    std::memcpy(dst, c.data_ptr<float>(), sizeof(float) * c.numel());
}

void put_i32(const torch::Tensor &t, int32_t *dst) {
    if (!dst) return;
    torch::Tensor c = t.detach().to(torch::kInt32).contiguous();
    std::memcpy(dst, c.data_ptr<int32_t>(), sizeof(int32_t) * c.numel());
}

struct RasterState {
    std::vector<int32_t> *px2gid = nullptr;
    int64_t pixels = 0;
};

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

#define REF_TRY try {
#define REF_CATCH                                                                                \
    }                                                                                            \
    catch (const std::exception &e) {                                                            \
        g_err = e.what();                                                                        \
        return -1;                                                                               \
    }                                                                                            \
    return 0;

extern "C" {

const char *ref_last_error() { return g_err.c_str(); }

int ref_num_threads() { return torch::get_num_threads(); }

// gsplat_cpu.cpp:48-131.  cov2d is written as N x 2 x 2, cam_depths as N (NDC z, contiguous copy);
// depth_keys_as_read (nullable): see below.
int ref_project_forward(int N, const float *means, const float *scales, float glob_scale,
                        const float *quats, const float *viewmat, const float *projmat, float fx,
                        float fy, float cx, float cy, int H, int W, float clip, float *xys,
                        int32_t *radii, float *conics, float *cov2d, float *cam_depths,
                        float *depth_keys_as_read) {
    REF_TRY
    torch::Tensor m = f32(means, {N, 3}), s = f32(scales, {N, 3}), q = f32(quats, {N, 4});
    torch::Tensor vm = f32(viewmat, {4, 4}), pm = f32(projmat, {4, 4});
    auto t = project_gaussians_forward_tensor_cpu(N, m, s, glob_scale, q, vm, pm, fx, fy, cx, cy,
                                                  H, W, clip);
    put(std::get<0>(t), xys);
    put_i32(std::get<1>(t), radii);
    put(std::get<2>(t), conics);
    put(std::get<3>(t), cov2d);
    put(std::get<4>(t), cam_depths);
    if (depth_keys_as_read) {
        // What rasterize_forward_tensor_cpu really sorts by: it takes camDepths.data_ptr()
        // (gsplat_cpu.cpp:152) of the NON-contiguous view pProj[..., 2] (:128) and indexes it with
        // unit stride (:157-158).  Reproduce that read verbatim (in bounds: a + 2 < 3N).
        const torch::Tensor &cd = std::get<4>(t);
        const float *p = static_cast<const float *>(cd.data_ptr());
        for (int a = 0; a < N; a++) depth_keys_as_read[a] = p[a];
    }
    REF_CATCH
}

// Backward of the CPU projection = libtorch autograd through gsplat_cpu.cpp:48-131 with the
// cotangents the CPU rasterizer hands back (xys and conics only; rasterize_gaussians.cpp:221-232).
int ref_project_backward(int N, const float *means, const float *scales, float glob_scale,
                         const float *quats, const float *viewmat, const float *projmat, float fx,
                         float fy, float cx, float cy, int H, int W, float clip,
                         const float *v_xys, const float *v_conics, float *v_means,
                         float *v_scales, float *v_quats) {
    REF_TRY
    torch::Tensor m = f32(means, {N, 3}).requires_grad_(true);
    torch::Tensor s = f32(scales, {N, 3}).requires_grad_(true);
    torch::Tensor q = f32(quats, {N, 4}).requires_grad_(true);
    torch::Tensor vm = f32(viewmat, {4, 4}), pm = f32(projmat, {4, 4});
    auto p = ProjectGaussiansCPU::apply(m, s, glob_scale, q, vm, pm, fx, fy, cx, cy, H, W, clip);
    auto g = torch::autograd::grad({p[0], p[2]}, {m, s, q},
                                   {f32(v_xys, {N, 2}), f32(v_conics, {N, 3})});
    put(g[0], v_means);
    put(g[1], v_scales);
    put(g[2], v_quats);
    REF_CATCH
}

// gsplat_cpu.cpp:424-486 through SphericalHarmonicsCPU::apply.  K = coeffs.size(-2).
int ref_sh_forward(int N, int K, int degrees_to_use, const float *dirs, const float *coeffs,
                   float *colors) {
    REF_TRY
    torch::Tensor c = SphericalHarmonicsCPU::apply(degrees_to_use, f32(dirs, {N, 3}),
                                                   f32(coeffs, {N, K, 3}));
    put(c, colors);
    REF_CATCH
}

int ref_sh_backward(int N, int K, int degrees_to_use, const float *dirs, const float *coeffs,
                    const float *v_colors, float *v_coeffs) {
    REF_TRY
    torch::Tensor co = f32(coeffs, {N, K, 3}).requires_grad_(true);
    torch::Tensor c = SphericalHarmonicsCPU::apply(degrees_to_use, f32(dirs, {N, 3}), co);
    auto g = torch::autograd::grad({c}, {co}, {f32(v_colors, {N, 3})});
    put(g[0], v_coeffs);
    REF_CATCH
}

// gsplat_cpu.cpp:137-257.  Returns an opaque state (the px2gid array) in *state; the caller must
// hand it to ref_rasterize_backward and/or ref_rasterize_free exactly once.
// px_counts (H*W, optional) receives the per-pixel contributor counts.
int ref_rasterize_forward(int W, int H, int N, const float *xys, const float *conics,
                          const float *colors, const float *opacities, const float *background,
                          const float *cov2d, const float *cam_depths, float *out_img,
                          float *final_Ts, int32_t *px_counts, void **state) {
    REF_TRY
    auto t = rasterize_forward_tensor_cpu(W, H, f32(xys, {N, 2}), f32(conics, {N, 3}),
                                          f32(colors, {N, 3}), f32(opacities, {N, 1}),
                                          f32(background, {3}), f32(cov2d, {N, 2, 2}),
                                          f32(cam_depths, {N}));
    put(std::get<0>(t), out_img);
    put(std::get<1>(t), final_Ts);
    RasterState *st = new RasterState();
    st->px2gid = std::get<2>(t);
    st->pixels = (int64_t)W * H;
    if (px_counts)
        for (int64_t i = 0; i < st->pixels; i++) px_counts[i] = (int32_t)st->px2gid[i].size();
    *state = st;
    REF_CATCH
}

// Copies the contributor ids (back-to-front per pixel, as the reference leaves them after the
// std::reverse at gsplat_cpu.cpp:252) into `ids`, pixels in raster order.
int ref_rasterize_contributors(void *state, int32_t *ids) {
    RasterState *st = static_cast<RasterState *>(state);
    int64_t o = 0;
    for (int64_t i = 0; i < st->pixels; i++)
        for (int32_t g : st->px2gid[i]) ids[o++] = g;
    return 0;
}

// gsplat_cpu.cpp:260-376.  v_out_alpha is all zeros, as at rasterize_gaussians.cpp:198.
int ref_rasterize_backward(int W, int H, int N, const float *xys, const float *conics,
                           const float *colors, const float *opacities, const float *background,
                           const float *cov2d, const float *cam_depths, const float *final_Ts,
                           void *state, const float *v_out, float *v_xy, float *v_conic,
                           float *v_colors, float *v_opacity) {
    REF_TRY
    RasterState *st = static_cast<RasterState *>(state);
    torch::Tensor vo = f32(v_out, {H, W, 3});
    torch::Tensor va = torch::zeros({H, W}, torch::kFloat32);
    auto t = rasterize_backward_tensor_cpu(H, W, f32(xys, {N, 2}), f32(conics, {N, 3}),
                                           f32(colors, {N, 3}), f32(opacities, {N, 1}),
                                           f32(background, {3}), f32(cov2d, {N, 2, 2}),
                                           f32(cam_depths, {N}), f32(final_Ts, {H, W}),
                                           st->px2gid, vo, va);
    put(std::get<0>(t), v_xy);
    put(std::get<1>(t), v_conic);
    put(std::get<2>(t), v_colors);
    put(std::get<3>(t), v_opacity);
    REF_CATCH
}

int ref_rasterize_free(void *state) {
    RasterState *st = static_cast<RasterState *>(state);
    if (st) {
        delete[] st->px2gid;
        delete st;
    }
    return 0;
}

// The whole hot path exactly as Model::forward's CPU branch strings it together
// (model.cpp:124-135,176-222) followed by backward with cotangent v_out:
//   ProjectGaussiansCPU -> [SphericalHarmonicsCPU -> clamp_min(+0.5, 0)] -> RasterizeGaussiansCPU.
// K == 0 means "no SH node": `coeffs` is then N x 3 colours used directly (simple_trainer.cpp:152-170)
// and v_coeffs is N x 3.  `scales` are the already-exponentiated scales, `quats` already normalised,
// `opacities` already sigmoid-ed (the caller-side torch ops are outside the op surface).
// times_ms[0] = forward wall time, times_ms[1] = backward wall time (steady_clock).
int ref_chain_fwd_bwd(int N, int K, int degrees_to_use, const float *means, const float *scales,
                      const float *quats, const float *dirs, const float *coeffs,
                      const float *opacities, const float *viewmat, const float *projmat, float fx,
                      float fy, float cx, float cy, int H, int W, const float *background,
                      const float *v_out, float *out_img, float *v_means, float *v_scales,
                      float *v_quats, float *v_coeffs, float *v_opacities, double *times_ms) {
    REF_TRY
    torch::Tensor m = f32(means, {N, 3}).requires_grad_(true);
    torch::Tensor s = f32(scales, {N, 3}).requires_grad_(true);
    torch::Tensor q = f32(quats, {N, 4}).requires_grad_(true);
    torch::Tensor o = f32(opacities, {N, 1}).requires_grad_(true);
    torch::Tensor co = (K > 0 ? f32(coeffs, {N, K, 3}) : f32(coeffs, {N, 3})).requires_grad_(true);
    torch::Tensor vm = f32(viewmat, {4, 4}), pm = f32(projmat, {4, 4});
    torch::Tensor bg = f32(background, {3});

    double t0 = now_ms();
    auto p = ProjectGaussiansCPU::apply(m, s, 1.0f, q, vm, pm, fx, fy, cx, cy, H, W);
    torch::Tensor rgbs;
    if (K > 0) {
        rgbs = SphericalHarmonicsCPU::apply(degrees_to_use, f32(dirs, {N, 3}), co);
        rgbs = torch::clamp_min(rgbs + 0.5f, 0.0f);
    } else {
        rgbs = co;
    }
    torch::Tensor img =
        RasterizeGaussiansCPU::apply(p[0], p[1], p[2], rgbs, o, p[3], p[4], H, W, bg);
    double t1 = now_ms();
    put(img, out_img);
    if (v_out) {
        double t2 = now_ms();
        img.backward(f32(v_out, {H, W, 3}));
        double t3 = now_ms();
        put(m.grad(), v_means);
        put(s.grad(), v_scales);
        put(q.grad(), v_quats);
        put(co.grad(), v_coeffs);
        put(o.grad(), v_opacities);
        if (times_ms) times_ms[1] = t3 - t2;
    }
    if (times_ms) times_ms[0] = t1 - t0;
    REF_CATCH
}

}  // extern "C"
