// bindings_hip_native.cpp — replaces rasterizer/gsplat/bindings.cu: the eight `*_tensor` launchers
// OpenSplat's operator files call (bindings_hip_native.h restates their prototypes), implemented on
// the C ABI of libgsplat_hip.so.  Each function only checks its inputs like the reference
// (CHECK_INPUT -> c10::Error), allocates the outputs on the inputs' device with the caching
// allocator, and enqueues kernels on the current HIP stream.  No tensor math here.
#include "bindings_hip_native.h"

#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>

#include <cstring>

#include "../../include/gsplat_compat.h"
#include "../../include/gsplat_hip.h"

using torch::Tensor;

gs_stream_t gsplatCurrentStream();   // torch_ops.cpp (same library): current HIP stream + the once-per-process ABI check

namespace {

gs_stream_t stream() { return gsplatCurrentStream(); }   // torch_ops.cpp: runs the ABI check in front of the first call

void ok(int rc, const char *what) {
    TORCH_CHECK(rc == GS_OK, what, " failed: ", gs_strerror(rc),
                rc == GS_ERR_HIP ? std::string(" — ") + gs_last_hip_error() : std::string());
}

Tensor f32c(const Tensor &t) { return t.to(torch::kFloat32).contiguous(); }

// host POD of the scalar camera arguments; the two matrices stay on the device
GsCamera camera(float fx, float fy, float cx, float cy, unsigned H, unsigned W, float clip, float glob) {
    GsCamera c;
    std::memset(&c, 0, sizeof(c));
    c.fx = fx; c.fy = fy; c.cx = cx; c.cy = cy;
    c.img_width = (int32_t)W; c.img_height = (int32_t)H;
    c.clip_thresh = clip; c.glob_scale = glob;
    return c;
}

// The compositing kernels consume one packed 48-byte record per Gaussian (gs_pack_splats).  At this
// level the caller hands over the separate 2-D tensors on every call, so the record is rebuilt per
// call; the rectangle is derived from the conics (no cov2d at this level), every Gaussian that
// appears in a tile list has a positive radius by construction of map_gaussian_to_intersects.
Tensor pack(int W, int H, const Tensor &xys, const Tensor &conics, const Tensor &colors,
            const Tensor &opacities) {
    const int64_t N = xys.size(0);
    auto fo = xys.options().dtype(torch::kFloat32);
    Tensor packed = torch::empty({N, (int64_t)GS_SPLAT_DWORDS}, fo);
    Tensor tiles = torch::empty({N}, fo.dtype(torch::kInt32));
    Tensor radii = torch::ones({N}, fo.dtype(torch::kInt32));
    Tensor x = f32c(xys), c = f32c(conics), col = f32c(colors), o = f32c(opacities);
    ok(gs_pack_splats(W, H, (int)N, x.data_ptr<float>(), radii.data_ptr<int32_t>(), c.data_ptr<float>(),
                      col.data_ptr<float>(), o.data_ptr<float>(), nullptr, packed.data_ptr<float>(),
                      tiles.data_ptr<int32_t>(), 0u, stream()),
       "gs_pack_splats");
    return packed;
}

// coverage masks of a caller-sorted list (gs_block_masks): the compositing kernels need them beside
// the ids; the reference's contract has no slot for them, so they are rebuilt per call here
Tensor block_masks(int W, int H, const Tensor &ids, const Tensor &bins, const Tensor &packed) {
    Tensor masks = torch::empty({std::max<int64_t>(ids.numel(), 1)}, ids.options().dtype(torch::kInt16));
    ok(gs_block_masks(W, H, ids.data_ptr<int32_t>(), bins.data_ptr<int32_t>(), packed.data_ptr<float>(),
                      reinterpret_cast<uint16_t *>(masks.data_ptr<int16_t>()), stream()),
       "gs_block_masks");
    return masks;
}

}  // namespace

Tensor compute_sh_forward_tensor(unsigned num_points, unsigned degree, unsigned degrees_to_use,
                                 Tensor &viewdirs, Tensor &coeffs) {
    CHECK_INPUT(viewdirs);
    CHECK_INPUT(coeffs);
    TORCH_CHECK(coeffs.dim() == 3 && coeffs.size(0) == (int64_t)num_points && coeffs.size(2) == 3,
                "coeffs must have dimensions (N, D, 3)");
    c10::DeviceGuard guard(coeffs.device());
    Tensor colors = torch::empty({(int64_t)num_points, 3}, coeffs.options());
    ok(gs_sh_forward((int)num_points, (int)coeffs.size(1), (int)degrees_to_use,
                     viewdirs.data_ptr<float>(), coeffs.data_ptr<float>(), colors.data_ptr<float>(),
                     stream()),
       "gs_sh_forward");
    return colors;
}

Tensor compute_sh_backward_tensor(unsigned num_points, unsigned degree, unsigned degrees_to_use,
                                  Tensor &viewdirs, Tensor &v_colors) {
    CHECK_INPUT(viewdirs);
    CHECK_INPUT(v_colors);
    TORCH_CHECK(v_colors.dim() == 2 && v_colors.size(0) == (int64_t)num_points && v_colors.size(1) == 3,
                "v_colors must have dimensions (N, 3)");
    c10::DeviceGuard guard(v_colors.device());
    const int64_t K = (int64_t)(degree + 1) * (degree + 1);   // num_sh_bases(degree)
    Tensor v_coeffs = torch::empty({(int64_t)num_points, K, 3}, v_colors.options());
    ok(gs_sh_backward((int)num_points, (int)K, (int)degrees_to_use, viewdirs.data_ptr<float>(),
                      v_colors.data_ptr<float>(), v_coeffs.data_ptr<float>(), stream()),
       "gs_sh_backward");
    return v_coeffs;
}

std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> project_gaussians_forward_tensor(
    const int num_points, Tensor &means3d, Tensor &scales, const float glob_scale, Tensor &quats,
    Tensor &viewmat, Tensor &projmat, const float fx, const float fy, const float cx, const float cy,
    const unsigned img_height, const unsigned img_width, const std::tuple<int, int, int> tile_bounds,
    const float clip_thresh) {
    CHECK_CUDA(means3d);
    c10::DeviceGuard guard(means3d.device());
    const int64_t N = num_points;
    auto fo = means3d.options().dtype(torch::kFloat32);
    auto io = fo.dtype(torch::kInt32);
    Tensor m = f32c(means3d), s = f32c(scales), q = f32c(quats), vm = f32c(viewmat), pm = f32c(projmat);
    Tensor cov3d = torch::empty({N, 6}, fo), xys = torch::empty({N, 2}, fo), depths = torch::empty({N}, fo);
    Tensor radii = torch::empty({N}, io), conics = torch::empty({N, 3}, fo), tiles = torch::empty({N}, io);
    Tensor cov2d = torch::empty({N, 3}, fo);
    GsCamera cam = camera(fx, fy, cx, cy, img_height, img_width, clip_thresh, glob_scale);
    ok(gs_project_forward(&cam, vm.data_ptr<float>(), pm.data_ptr<float>(), (int)N, m.data_ptr<float>(),
                          s.data_ptr<float>(), q.data_ptr<float>(), xys.data_ptr<float>(),
                          depths.data_ptr<float>(), radii.data_ptr<int32_t>(), conics.data_ptr<float>(),
                          tiles.data_ptr<int32_t>(), cov3d.data_ptr<float>(), cov2d.data_ptr<float>(),
                          stream()),
       "gs_project_forward");
    // the binning contract of this level counts tiles by the radius square (forward.cu:86-94)
    ok(gs_compat_tiles_hit((int)N, xys.data_ptr<float>(), radii.data_ptr<int32_t>(),
                           std::get<0>(tile_bounds), std::get<1>(tile_bounds), tiles.data_ptr<int32_t>(),
                           stream()),
       "gs_compat_tiles_hit");
    return std::make_tuple(cov3d, xys, depths, radii, conics, tiles);
}

std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> project_gaussians_backward_tensor(
    const int num_points, Tensor &means3d, Tensor &scales, const float glob_scale, Tensor &quats,
    Tensor &viewmat, Tensor &projmat, const float fx, const float fy, const float cx, const float cy,
    const unsigned img_height, const unsigned img_width, Tensor &cov3d, Tensor &radii, Tensor &conics,
    Tensor &v_xy, Tensor &v_depth, Tensor &v_conic) {
    CHECK_CUDA(means3d);
    c10::DeviceGuard guard(means3d.device());
    const int64_t N = num_points;
    auto fo = means3d.options().dtype(torch::kFloat32);
    Tensor m = f32c(means3d), s = f32c(scales), q = f32c(quats), vm = f32c(viewmat), pm = f32c(projmat);
    Tensor r = radii.to(torch::kInt32).contiguous();
    Tensor vxy = f32c(v_xy), vc = f32c(v_conic);
    Tensor vd = v_depth.defined() && v_depth.numel() == N ? f32c(v_depth) : Tensor();
    // the reference's scratch outputs: not needed by the caller's backward (project_gaussians.cpp:77-92)
    Tensor v_cov2d = torch::zeros({N, 3}, fo), v_cov3d = torch::zeros({N, 6}, fo);
    Tensor v_mean = torch::empty({N, 3}, fo), v_scale = torch::empty({N, 3}, fo), v_quat = torch::empty({N, 4}, fo);
    GsCamera cam = camera(fx, fy, cx, cy, img_height, img_width, 0.01f, glob_scale);
    ok(gs_project_backward(&cam, vm.data_ptr<float>(), pm.data_ptr<float>(), (int)N, m.data_ptr<float>(),
                           s.data_ptr<float>(), q.data_ptr<float>(), r.data_ptr<int32_t>(),
                           vxy.data_ptr<float>(), vd.defined() ? vd.data_ptr<float>() : nullptr,
                           vc.data_ptr<float>(), v_mean.data_ptr<float>(), v_scale.data_ptr<float>(),
                           v_quat.data_ptr<float>(), stream()),
       "gs_project_backward");
    return std::make_tuple(v_cov2d, v_cov3d, v_mean, v_scale, v_quat);
}

std::tuple<Tensor, Tensor> map_gaussian_to_intersects_tensor(
    const int num_points, const int num_intersects, const Tensor &xys, const Tensor &depths,
    const Tensor &radii, const Tensor &cum_tiles_hit, const std::tuple<int, int, int> tile_bounds) {
    CHECK_INPUT(xys);
    CHECK_INPUT(depths);
    CHECK_INPUT(radii);
    CHECK_INPUT(cum_tiles_hit);
    c10::DeviceGuard guard(xys.device());
    Tensor ids = torch::zeros({(int64_t)num_intersects}, xys.options().dtype(torch::kInt64));
    Tensor gids = torch::zeros({(int64_t)num_intersects}, xys.options().dtype(torch::kInt32));
    Tensor cum = cum_tiles_hit.to(torch::kInt32).contiguous();
    Tensor r = radii.to(torch::kInt32).contiguous();
    ok(gs_compat_map_intersects(num_points, xys.data_ptr<float>(), depths.data_ptr<float>(),
                                r.data_ptr<int32_t>(), cum.data_ptr<int32_t>(), std::get<0>(tile_bounds),
                                std::get<1>(tile_bounds), ids.data_ptr<int64_t>(),
                                gids.data_ptr<int32_t>(), stream()),
       "gs_compat_map_intersects");
    return std::make_tuple(ids, gids);
}

Tensor get_tile_bin_edges_tensor(int num_intersects, const Tensor &isect_ids_sorted) {
    CHECK_INPUT(isect_ids_sorted);
    c10::DeviceGuard guard(isect_ids_sorted.device());
    // the reference allocates [num_intersects, 2] and indexes it by tile id (bindings.cu:324-326):
    // a frame with fewer intersections than tiles would write out of bounds there; 65536 extra rows
    // cover every image up to 4096 x 4096 pixels
    Tensor bins = torch::zeros({(int64_t)num_intersects + 65536, 2},
                               isect_ids_sorted.options().dtype(torch::kInt32));
    ok(gs_compat_tile_bin_edges(num_intersects, isect_ids_sorted.data_ptr<int64_t>(),
                                bins.data_ptr<int32_t>(), bins.size(0), stream()),
       "gs_compat_tile_bin_edges");
    return bins;
}

std::tuple<Tensor, Tensor, Tensor> rasterize_forward_tensor(
    const std::tuple<int, int, int> tile_bounds, const std::tuple<int, int, int> block,
    const std::tuple<int, int, int> img_size, const Tensor &gaussian_ids_sorted, const Tensor &tile_bins,
    const Tensor &xys, const Tensor &conics, const Tensor &colors, const Tensor &opacities,
    const Tensor &background) {
    CHECK_INPUT(gaussian_ids_sorted);
    CHECK_INPUT(tile_bins);
    CHECK_INPUT(xys);
    CHECK_INPUT(conics);
    CHECK_INPUT(colors);
    CHECK_INPUT(opacities);
    CHECK_INPUT(background);
    TORCH_CHECK(std::get<0>(block) == GS_TILE && std::get<1>(block) == GS_TILE, "block must be 16 x 16");
    TORCH_CHECK(colors.dim() == 2 && colors.size(1) == 3, "colors must have 3 channels");
    c10::DeviceGuard guard(xys.device());
    const int W = std::get<0>(img_size), H = std::get<1>(img_size);
    const int tiles = std::get<0>(tile_bounds) * std::get<1>(tile_bounds);
    TORCH_CHECK(tile_bins.size(0) >= tiles, "tile_bins has fewer rows than tiles");
    auto fo = xys.options().dtype(torch::kFloat32);
    Tensor packed = pack(W, H, xys, conics, colors, opacities);
    Tensor ids = gaussian_ids_sorted.to(torch::kInt32).contiguous();
    Tensor bins = tile_bins.to(torch::kInt32).contiguous();
    Tensor bg = f32c(background);
    Tensor img = torch::empty({H, W, 3}, fo), Ts = torch::empty({H, W}, fo);
    Tensor idx = torch::empty({H, W}, fo.dtype(torch::kInt32));
    Tensor masks = block_masks(W, H, ids, bins, packed);
    ok(gs_rasterize_forward(W, H, ids.data_ptr<int32_t>(),
                            reinterpret_cast<const uint16_t *>(masks.data_ptr<int16_t>()),
                            bins.data_ptr<int32_t>(),
                            packed.data_ptr<float>(), bg.data_ptr<float>(), img.data_ptr<float>(),
                            Ts.data_ptr<float>(), idx.data_ptr<int32_t>(), nullptr, nullptr, nullptr, 0u,
                            stream()),
       "gs_rasterize_forward");
    return std::make_tuple(img, Ts, idx);
}

std::tuple<Tensor, Tensor, Tensor, Tensor> rasterize_backward_tensor(
    const unsigned img_height, const unsigned img_width, const Tensor &gaussians_ids_sorted,
    const Tensor &tile_bins, const Tensor &xys, const Tensor &conics, const Tensor &colors,
    const Tensor &opacities, const Tensor &background, const Tensor &final_Ts, const Tensor &final_idx,
    const Tensor &v_output, const Tensor &v_output_alpha) {
    CHECK_INPUT(xys);
    CHECK_INPUT(colors);
    TORCH_CHECK(xys.dim() == 2 && xys.size(1) == 2, "xys must have dimensions (N, 2)");
    TORCH_CHECK(colors.dim() == 2 && colors.size(1) == 3, "colors must have dimensions (N, 3)");
    c10::DeviceGuard guard(xys.device());
    const int W = (int)img_width, H = (int)img_height;
    const int64_t N = xys.size(0);
    auto fo = xys.options().dtype(torch::kFloat32);
    Tensor packed = pack(W, H, xys, conics, colors, opacities);
    Tensor ids = gaussians_ids_sorted.to(torch::kInt32).contiguous();
    Tensor bins = tile_bins.to(torch::kInt32).contiguous();
    Tensor bg = f32c(background), fT = f32c(final_Ts), vo = f32c(v_output);
    Tensor fi = final_idx.to(torch::kInt32).contiguous();
    Tensor voa = v_output_alpha.defined() && v_output_alpha.numel() == (int64_t)W * H ? f32c(v_output_alpha) : Tensor();
    Tensor v_xy = torch::empty({N, 2}, fo), v_conic = torch::empty({N, 3}, fo);
    Tensor v_colors = torch::empty({N, 3}, fo), v_opacity = torch::empty({N, 1}, fo);
    const size_t ws_bytes = gs_rasterize_backward_workspace_bytes((int)N);
    Tensor ws = torch::empty({(int64_t)ws_bytes + 64}, fo.dtype(torch::kUInt8));
    char *wp = reinterpret_cast<char *>(ws.data_ptr<uint8_t>());
    wp += (64 - (reinterpret_cast<uintptr_t>(wp) & 63u)) & 63u;
    Tensor masks = block_masks(W, H, ids, bins, packed);
    ok(gs_rasterize_backward(W, H, (int)N, ids.data_ptr<int32_t>(),
                             reinterpret_cast<const uint16_t *>(masks.data_ptr<int16_t>()),
                             bins.data_ptr<int32_t>(),
                             packed.data_ptr<float>(), bg.data_ptr<float>(), fT.data_ptr<float>(),
                             fi.data_ptr<int32_t>(), vo.data_ptr<float>(),
                             voa.defined() ? voa.data_ptr<float>() : nullptr, nullptr,
                             v_xy.data_ptr<float>(), v_conic.data_ptr<float>(), v_colors.data_ptr<float>(),
                             v_opacity.data_ptr<float>(), wp, ws_bytes, nullptr, nullptr, 0u, stream()),
       "gs_rasterize_backward");
    return std::make_tuple(v_xy, v_conic, v_colors, v_opacity);
}

// ---- Python face of the launcher level (tests drive the eight functions through torch.ops) --------
#include <torch/library.h>

namespace {
using TL = std::vector<Tensor>;
TL op_l_project_fwd(Tensor means, Tensor scales, double glob, Tensor quats, Tensor vm, Tensor pm, double fx,
                    double fy, double cx, double cy, int64_t H, int64_t W, int64_t tiles_x, int64_t tiles_y,
                    double clip) {
    auto t = project_gaussians_forward_tensor((int)means.size(0), means, scales, (float)glob, quats, vm, pm,
                                              (float)fx, (float)fy, (float)cx, (float)cy, (unsigned)H,
                                              (unsigned)W, std::make_tuple((int)tiles_x, (int)tiles_y, 1),
                                              (float)clip);
    return {std::get<0>(t), std::get<1>(t), std::get<2>(t), std::get<3>(t), std::get<4>(t), std::get<5>(t)};
}
TL op_l_project_bwd(Tensor means, Tensor scales, double glob, Tensor quats, Tensor vm, Tensor pm, double fx,
                    double fy, double cx, double cy, int64_t H, int64_t W, Tensor cov3d, Tensor radii,
                    Tensor conics, Tensor v_xy, Tensor v_depth, Tensor v_conic) {
    auto t = project_gaussians_backward_tensor((int)means.size(0), means, scales, (float)glob, quats, vm, pm,
                                               (float)fx, (float)fy, (float)cx, (float)cy, (unsigned)H,
                                               (unsigned)W, cov3d, radii, conics, v_xy, v_depth, v_conic);
    return {std::get<0>(t), std::get<1>(t), std::get<2>(t), std::get<3>(t), std::get<4>(t)};
}
Tensor op_l_sh_fwd(int64_t degree, int64_t degrees_to_use, Tensor dirs, Tensor coeffs) {
    return compute_sh_forward_tensor((unsigned)coeffs.size(0), (unsigned)degree, (unsigned)degrees_to_use, dirs, coeffs);
}
Tensor op_l_sh_bwd(int64_t degree, int64_t degrees_to_use, Tensor dirs, Tensor v_colors) {
    return compute_sh_backward_tensor((unsigned)v_colors.size(0), (unsigned)degree, (unsigned)degrees_to_use, dirs, v_colors);
}
TL op_l_map(int64_t num_intersects, Tensor xys, Tensor depths, Tensor radii, Tensor cum, int64_t tiles_x,
            int64_t tiles_y) {
    auto t = map_gaussian_to_intersects_tensor((int)xys.size(0), (int)num_intersects, xys, depths, radii, cum,
                                               std::make_tuple((int)tiles_x, (int)tiles_y, 1));
    return {std::get<0>(t), std::get<1>(t)};
}
Tensor op_l_edges(int64_t num_intersects, Tensor ids_sorted) {
    return get_tile_bin_edges_tensor((int)num_intersects, ids_sorted);
}
TL op_l_rast_fwd(int64_t tiles_x, int64_t tiles_y, int64_t W, int64_t H, Tensor ids, Tensor bins, Tensor xys,
                 Tensor conics, Tensor colors, Tensor opac, Tensor bg) {
    auto t = rasterize_forward_tensor(std::make_tuple((int)tiles_x, (int)tiles_y, 1), std::make_tuple(16, 16, 1),
                                      std::make_tuple((int)W, (int)H, 1), ids, bins, xys, conics, colors, opac, bg);
    return {std::get<0>(t), std::get<1>(t), std::get<2>(t)};
}
TL op_l_rast_bwd(int64_t H, int64_t W, Tensor ids, Tensor bins, Tensor xys, Tensor conics, Tensor colors,
                 Tensor opac, Tensor bg, Tensor final_Ts, Tensor final_idx, Tensor v_out, Tensor v_out_alpha) {
    auto t = rasterize_backward_tensor((unsigned)H, (unsigned)W, ids, bins, xys, conics, colors, opac, bg, final_Ts,
                                       final_idx, v_out, v_out_alpha);
    return {std::get<0>(t), std::get<1>(t), std::get<2>(t), std::get<3>(t)};
}
}  // namespace

TORCH_LIBRARY_FRAGMENT(opensplat_amd, m) {
    m.def("launcher_project_gaussians_forward(Tensor means, Tensor scales, float glob_scale, Tensor quats, "
          "Tensor viewmat, Tensor projmat, float fx, float fy, float cx, float cy, int img_height, "
          "int img_width, int tiles_x, int tiles_y, float clip_thresh) -> Tensor[]", &op_l_project_fwd);
    m.def("launcher_project_gaussians_backward(Tensor means, Tensor scales, float glob_scale, Tensor quats, "
          "Tensor viewmat, Tensor projmat, float fx, float fy, float cx, float cy, int img_height, "
          "int img_width, Tensor cov3d, Tensor radii, Tensor conics, Tensor v_xy, Tensor v_depth, "
          "Tensor v_conic) -> Tensor[]", &op_l_project_bwd);
    m.def("launcher_compute_sh_forward(int degree, int degrees_to_use, Tensor viewdirs, Tensor coeffs) -> Tensor",
          &op_l_sh_fwd);
    m.def("launcher_compute_sh_backward(int degree, int degrees_to_use, Tensor viewdirs, Tensor v_colors) -> Tensor",
          &op_l_sh_bwd);
    m.def("launcher_map_gaussian_to_intersects(int num_intersects, Tensor xys, Tensor depths, Tensor radii, "
          "Tensor cum_tiles_hit, int tiles_x, int tiles_y) -> Tensor[]", &op_l_map);
    m.def("launcher_get_tile_bin_edges(int num_intersects, Tensor isect_ids_sorted) -> Tensor", &op_l_edges);
    m.def("launcher_rasterize_forward(int tiles_x, int tiles_y, int img_width, int img_height, "
          "Tensor gaussian_ids_sorted, Tensor tile_bins, Tensor xys, Tensor conics, Tensor colors, "
          "Tensor opacities, Tensor background) -> Tensor[]", &op_l_rast_fwd);
    m.def("launcher_rasterize_backward(int img_height, int img_width, Tensor gaussian_ids_sorted, "
          "Tensor tile_bins, Tensor xys, Tensor conics, Tensor colors, Tensor opacities, Tensor background, "
          "Tensor final_Ts, Tensor final_idx, Tensor v_output, Tensor v_output_alpha) -> Tensor[]",
          &op_l_rast_bwd);
}
