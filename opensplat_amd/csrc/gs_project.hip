// gs_project.hip — 3-D -> 2-D projection of Gaussians and its VJP, one lane per Gaussian.
//
// Replaces project_gaussians_forward_kernel / project_gaussians_backward_kernel
// (reference rasterizer/gsplat/forward.cu:19-103, backward.cu:357-542).  The arithmetic follows
// the CPU oracle's batched formulation (rasterizer/gsplat-cpu/gsplat_cpu.cpp:48-131), including
// its det / w clamps and pixel-centre formula; the near-plane cull and the view-space depth /
// cov3d / tile-count outputs are the GPU operator surface's (forward.cu:49-52,95-99).
//
// Roofline: pure streaming, HBM-bound.  Forward reads 40 B and writes 84 B per Gaussian, backward
// reads 64 B and writes 40 B; the camera lives in kernel arguments (SGPRs).  Loads/stores are
// per-lane contiguous 12/16-byte vectors over a dense index range, i.e. every fetched line is
// fully consumed by the wave that touches it.
#include "gs_gaussian.h"

namespace gs {

__global__ void __launch_bounds__(256)
k_project_forward(CamArgs cam, const float *__restrict__ vm_dev, const float *__restrict__ pm_dev,
                  int N, const float *__restrict__ means,
                  const float *__restrict__ scales, const float *__restrict__ quats,
                  float *__restrict__ xys, float *__restrict__ depths, int32_t *__restrict__ radii,
                  float *__restrict__ conics, int32_t *__restrict__ num_tiles_hit,
                  float *__restrict__ cov3d, float *__restrict__ cov2d) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    load_device_matrices(cam, vm_dev, pm_dev);
    float mean[3] = {means[3 * n], means[3 * n + 1], means[3 * n + 2]};
    float scale[3] = {scales[3 * n], scales[3 * n + 1], scales[3 * n + 2]};
    if (cam.flags & GS_CAM_LOG_SCALES) {  // fused torch::exp(scales), model.cpp:148
#pragma unroll
        for (int j = 0; j < 3; j++) scale[j] = expf(scale[j]);
    }
    const float4_u q4 = reinterpret_cast<const float4_u *>(quats)[n];  // may be a slice: 4-B aligned
    float quat[4] = {q4.x, q4.y, q4.z, q4.w};
    Proj o;
    project_one(cam, mean, scale, quat, o);

    ProjOut r;
    project_outputs(cam, o, r);
    const float u = r.u, v = r.v, conic0 = r.conic[0], conic1 = r.conic[1], conic2 = r.conic[2];
    const int rad = r.radius, tiles = r.tiles;
    xys[2 * n + 0] = u;
    xys[2 * n + 1] = v;
    depths[n] = o.p[2];
    radii[n] = rad;
    conics[3 * n + 0] = conic0;
    conics[3 * n + 1] = conic1;
    conics[3 * n + 2] = conic2;
    num_tiles_hit[n] = tiles;
    cov3d[6 * n + 0] = o.S3[0];
    cov3d[6 * n + 1] = o.S3[1];
    cov3d[6 * n + 2] = o.S3[2];
    cov3d[6 * n + 3] = o.S3[4];
    cov3d[6 * n + 4] = o.S3[5];
    cov3d[6 * n + 5] = o.S3[8];
    cov2d[3 * n + 0] = o.a;
    cov2d[3 * n + 1] = o.b;
    cov2d[3 * n + 2] = o.c;
}

// VJP; derivation in DESIGN.md §"Projection backward".  Equals libtorch autograd through
// gsplat_cpu.cpp:48-131 for cotangents on xys and conics (plus the GPU surface's v_depth).
__global__ void __launch_bounds__(256)
k_project_backward(CamArgs cam, const float *__restrict__ vm_dev,
                   const float *__restrict__ pm_dev, int N, const float *__restrict__ means,
                   const float *__restrict__ scales, const float *__restrict__ quats,
                   const int32_t *__restrict__ radii, const float *__restrict__ v_xy,
                   const float *__restrict__ v_depth, const float *__restrict__ v_conic,
                   float *__restrict__ v_means, float *__restrict__ v_scales,
                   float *__restrict__ v_quats) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    load_device_matrices(cam, vm_dev, pm_dev);
    float4_u *vq4 = reinterpret_cast<float4_u *>(v_quats);
    if (radii[n] <= 0) {  // backward.cu:380-382: culled Gaussians get no gradient
        v_means[3 * n] = v_means[3 * n + 1] = v_means[3 * n + 2] = 0.0f;
        v_scales[3 * n] = v_scales[3 * n + 1] = v_scales[3 * n + 2] = 0.0f;
        vq4[n] = (float4_u)(0.0f);
        return;
    }
    float mean[3] = {means[3 * n], means[3 * n + 1], means[3 * n + 2]};
    float scale[3] = {scales[3 * n], scales[3 * n + 1], scales[3 * n + 2]};
    if (cam.flags & GS_CAM_LOG_SCALES) {
#pragma unroll
        for (int j = 0; j < 3; j++) scale[j] = expf(scale[j]);
    }
    const float4_u q4 = reinterpret_cast<const float4_u *>(quats)[n];  // may be a slice: 4-B aligned
    float quat[4] = {q4.x, q4.y, q4.z, q4.w};
    Proj o;
    project_one(cam, mean, scale, quat, o);
    ProjGrad g;
    project_backward_one(cam, o, scale, v_xy[2 * n + 0], v_xy[2 * n + 1], v_conic[3 * n + 0],
                         v_conic[3 * n + 1], v_conic[3 * n + 2], v_depth ? v_depth[n] : 0.0f, g);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        v_means[3 * n + j] = g.v_mean[j];
        v_scales[3 * n + j] = g.v_scale[j];
    }
    float4_u vq;
    vq.x = g.v_quat[0]; vq.y = g.v_quat[1]; vq.z = g.v_quat[2]; vq.w = g.v_quat[3];
    vq4[n] = vq;
}

}  // namespace gs

extern "C" int gs_project_forward(const GsCamera *cam, const float *viewmat_dev,
                                  const float *projmat_dev, int N, const float *means,
                                  const float *scales, const float *quats, float *xys,
                                  float *depths, int32_t *radii, float *conics,
                                  int32_t *num_tiles_hit, float *cov3d, float *cov2d,
                                  gs_stream_t stream) {
    if (!cam || N < 0) return GS_ERR_INVALID_ARGUMENT;
    if (N == 0) return GS_OK;
    if (!means || !scales || !quats || !xys || !depths || !radii || !conics || !num_tiles_hit ||
        !cov3d || !cov2d)
        return GS_ERR_INVALID_ARGUMENT;
    if (cam->img_width <= 0 || cam->img_height <= 0) return GS_ERR_INVALID_ARGUMENT;
    if (cam->img_width > 65535 || cam->img_height > 65535) return GS_ERR_UNSUPPORTED;
    gs::CamArgs a = gs::make_cam(cam);
    int blocks = (N + 255) / 256;
    GS_LAUNCH(gs::k_project_forward, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a,
                       viewmat_dev, projmat_dev, N, means, scales, quats, xys, depths, radii, conics, num_tiles_hit, cov3d,
                       cov2d);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" int gs_project_backward(const GsCamera *cam, const float *viewmat_dev,
                                   const float *projmat_dev, int N, const float *means,
                                   const float *scales, const float *quats, const int32_t *radii,
                                   const float *v_xy, const float *v_depth, const float *v_conic,
                                   float *v_means, float *v_scales, float *v_quats,
                                   gs_stream_t stream) {
    if (!cam || N < 0) return GS_ERR_INVALID_ARGUMENT;
    if (N == 0) return GS_OK;
    if (!means || !scales || !quats || !radii || !v_xy || !v_conic || !v_means || !v_scales ||
        !v_quats)
        return GS_ERR_INVALID_ARGUMENT;
    gs::CamArgs a = gs::make_cam(cam);
    int blocks = (N + 255) / 256;
    GS_LAUNCH(gs::k_project_backward, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a,
                       viewmat_dev, projmat_dev, N, means, scales, quats, radii, v_xy, v_depth, v_conic, v_means, v_scales,
                       v_quats);
    GS_LAUNCH_CHECK();
    return GS_OK;
}
