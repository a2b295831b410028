// bindings_hip_native.h — the LAUNCHER-LEVEL surface of OpenSplat's GPU rasterizer, restated.
//
// OpenSplat's operator files (project_gaussians.cpp, rasterize_gaussians.cpp,
// spherical_harmonics.cpp) call eight `*_tensor` functions declared in
// rasterizer/gsplat/bindings.h:26-40 (SH), :42-93 (projection), :96-109 (binning), :111-126 and
// :169-190 (compositing).  That header cannot be included outside the reference tree (it pulls
// forward.cuh / glm), so the eight prototypes are restated here with the same names, argument
// order, argument types and return order; bindings_hip_native.cpp implements them on the C ABI of
// libgsplat_hip.so.  A maintainer who wants to keep OpenSplat's operator files untouched replaces
// rasterizer/gsplat/bindings.cu by bindings_hip_native.cpp and this header.
//
// Semantics at THIS level are the GPU reference's: tiles by the radius square (the caller's own
// cumsum / torch::sort / gather glue decides what each tile sees), per-pixel arithmetic of
// gsplat-cpu.  The native operators (gsplat_ops.hpp) bin by the CPU pixel rectangle instead and
// need no global sort.
#pragma once

#include <torch/torch.h>

#include <tuple>

#ifndef CHECK_INPUT
#define CHECK_CUDA(x) TORCH_CHECK(x.is_cuda(), #x " must be a CUDA tensor")
#define CHECK_CONTIGUOUS(x) TORCH_CHECK(x.is_contiguous(), #x " must be contiguous")
#define CHECK_INPUT(x) \
    CHECK_CUDA(x);     \
    CHECK_CONTIGUOUS(x)
#endif

// bindings.h:26-32 / :34-40
torch::Tensor compute_sh_forward_tensor(unsigned num_points, unsigned degree, unsigned degrees_to_use,
                                        torch::Tensor &viewdirs, torch::Tensor &coeffs);
torch::Tensor compute_sh_backward_tensor(unsigned num_points, unsigned degree, unsigned degrees_to_use,
                                         torch::Tensor &viewdirs, torch::Tensor &v_colors);

// bindings.h:42-64 -> (cov3d, xys, depths, radii, conics, num_tiles_hit)
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
project_gaussians_forward_tensor(const int num_points, torch::Tensor &means3d, torch::Tensor &scales,
                                 const float glob_scale, torch::Tensor &quats, torch::Tensor &viewmat,
                                 torch::Tensor &projmat, const float fx, const float fy, const float cx,
                                 const float cy, const unsigned img_height, const unsigned img_width,
                                 const std::tuple<int, int, int> tile_bounds, const float clip_thresh);

// bindings.h:66-93 -> (v_cov2d, v_cov3d, v_mean3d, v_scale, v_quat)
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
project_gaussians_backward_tensor(const int num_points, torch::Tensor &means3d, torch::Tensor &scales,
                                  const float glob_scale, torch::Tensor &quats, torch::Tensor &viewmat,
                                  torch::Tensor &projmat, const float fx, const float fy, const float cx,
                                  const float cy, const unsigned img_height, const unsigned img_width,
                                  torch::Tensor &cov3d, torch::Tensor &radii, torch::Tensor &conics,
                                  torch::Tensor &v_xy, torch::Tensor &v_depth, torch::Tensor &v_conic);

// bindings.h:96-104 -> (isect_ids int64 [M], gaussian_ids int32 [M]), unsorted
std::tuple<torch::Tensor, torch::Tensor> map_gaussian_to_intersects_tensor(
    const int num_points, const int num_intersects, const torch::Tensor &xys,
    const torch::Tensor &depths, const torch::Tensor &radii, const torch::Tensor &cum_tiles_hit,
    const std::tuple<int, int, int> tile_bounds);

// bindings.h:106-109 -> tile_bins int32 [rows, 2]
torch::Tensor get_tile_bin_edges_tensor(int num_intersects, const torch::Tensor &isect_ids_sorted);

// bindings.h:111-126 -> (out_img [H,W,3], final_Ts [H,W], final_idx [H,W] int32)
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> rasterize_forward_tensor(
    const std::tuple<int, int, int> tile_bounds, const std::tuple<int, int, int> block,
    const std::tuple<int, int, int> img_size, const torch::Tensor &gaussian_ids_sorted,
    const torch::Tensor &tile_bins, const torch::Tensor &xys, const torch::Tensor &conics,
    const torch::Tensor &colors, const torch::Tensor &opacities, const torch::Tensor &background);

// bindings.h:169-190 -> (v_xy, v_conic, v_colors, v_opacity)
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> rasterize_backward_tensor(
    const unsigned img_height, const unsigned img_width, const torch::Tensor &gaussians_ids_sorted,
    const torch::Tensor &tile_bins, const torch::Tensor &xys, const torch::Tensor &conics,
    const torch::Tensor &colors, const torch::Tensor &opacities, const torch::Tensor &background,
    const torch::Tensor &final_Ts, const torch::Tensor &final_idx, const torch::Tensor &v_output,
    const torch::Tensor &v_output_alpha);
