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
#include "gs_device.h"

namespace gs {

struct CamArgs {
    float vm[12];  // rows 0..2 of viewmat
    float pm[16];
    float fx, fy, cx, cy;
    int W, H;
    float clip, glob;
    uint32_t flags;
};

static CamArgs make_cam(const GsCamera *c) {
    CamArgs a;
    for (int i = 0; i < 12; i++) a.vm[i] = c->viewmat[i];
    for (int i = 0; i < 16; i++) a.pm[i] = c->projmat[i];
    a.fx = c->fx; a.fy = c->fy; a.cx = c->cx; a.cy = c->cy;
    a.W = c->img_width; a.H = c->img_height;
    a.clip = c->clip_thresh; a.glob = c->glob_scale;
    a.flags = c->flags;
    return a;
}

// The operator surface hands the two matrices over as device tensors (model.cpp:93-113 builds
// them on the device); reading them here through wave-uniform (scalar) loads avoids the
// device->host copy + sync a host-side camera struct would need.
__device__ __forceinline__ void load_device_matrices(CamArgs &cam, const float *__restrict__ vm_dev,
                                                     const float *__restrict__ pm_dev) {
    if (vm_dev) {
#pragma unroll
        for (int i = 0; i < 12; i++) cam.vm[i] = vm_dev[i];
    }
    if (pm_dev) {
#pragma unroll
        for (int i = 0; i < 16; i++) cam.pm[i] = pm_dev[i];
    }
}

// Everything forward and backward both need about one Gaussian.
struct Proj {
    float p[3];       // view-space position
    float Rq[9];      // rotation of the normalised quaternion
    float M[9];       // Rq * glob * scale
    float S3[9];      // M M^T
    float t0, t1;     // FOV-clamped view x, y
    int clx, cly;     // active clamp side (-1, 0, +1)
    float rz, rz2;
    float T[6];       // J * Rview, 2x3
    float a, b, c;    // cov2d incl. blur
    float det_raw, det;
    float ph[4], rw;
    float qn;         // |quat| (clamped at 1e-12)
    float u[4];       // normalised quat
};

__device__ __forceinline__ void project_one(const CamArgs &cam, const float *mean,
                                            const float *scale, const float *quat, Proj &o) {
    const float *vm = cam.vm;
#pragma unroll
    for (int i = 0; i < 3; i++)
        o.p[i] = vm[4 * i + 0] * mean[0] + vm[4 * i + 1] * mean[1] + vm[4 * i + 2] * mean[2] +
                 vm[4 * i + 3];
    // quatToRot, gsplat_cpu.cpp:16-40 (F.normalize eps = 1e-12)
    float n = sqrtf(quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2] + quat[3] * quat[3]);
    n = fmaxf(n, 1e-12f);
    o.qn = n;
    float w = quat[0] / n, x = quat[1] / n, y = quat[2] / n, z = quat[3] / n;
    o.u[0] = w; o.u[1] = x; o.u[2] = y; o.u[3] = z;
    o.Rq[0] = 1.0f - 2.0f * (y * y + z * z);
    o.Rq[1] = 2.0f * (x * y - w * z);
    o.Rq[2] = 2.0f * (x * z + w * y);
    o.Rq[3] = 2.0f * (x * y + w * z);
    o.Rq[4] = 1.0f - 2.0f * (x * x + z * z);
    o.Rq[5] = 2.0f * (y * z - w * x);
    o.Rq[6] = 2.0f * (x * z - w * y);
    o.Rq[7] = 2.0f * (y * z + w * x);
    o.Rq[8] = 1.0f - 2.0f * (x * x + y * y);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) o.M[3 * i + j] = o.Rq[3 * i + j] * cam.glob * scale[j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            o.S3[3 * i + j] = o.M[3 * i + 0] * o.M[3 * j + 0] + o.M[3 * i + 1] * o.M[3 * j + 1] +
                              o.M[3 * i + 2] * o.M[3 * j + 2];
    // project_cov3d_ewa, gsplat_cpu.cpp:64-99
    float fovx = 0.5f * (float)cam.W / cam.fx;
    float fovy = 0.5f * (float)cam.H / cam.fy;
    float limx = 1.3f * fovx, limy = 1.3f * fovy;
    float xz = o.p[0] / o.p[2], yz = o.p[1] / o.p[2];
    o.clx = (xz > limx) ? 1 : ((xz < -limx) ? -1 : 0);
    o.cly = (yz > limy) ? 1 : ((yz < -limy) ? -1 : 0);
    o.t0 = o.p[2] * fminf(limx, fmaxf(-limx, xz));
    o.t1 = o.p[2] * fminf(limy, fmaxf(-limy, yz));
    o.rz = 1.0f / o.p[2];
    o.rz2 = o.rz * o.rz;
    float J00 = cam.fx * o.rz, J02 = -cam.fx * o.t0 * o.rz2;
    float J11 = cam.fy * o.rz, J12 = -cam.fy * o.t1 * o.rz2;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        o.T[j] = J00 * vm[j] + J02 * vm[8 + j];
        o.T[3 + j] = J11 * vm[4 + j] + J12 * vm[8 + j];
    }
    float CT[6];  // S3 * T^T
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
            CT[2 * i + j] = o.S3[3 * i + 0] * o.T[3 * j + 0] + o.S3[3 * i + 1] * o.T[3 * j + 1] +
                            o.S3[3 * i + 2] * o.T[3 * j + 2];
    o.a = (o.T[0] * CT[0] + o.T[1] * CT[2] + o.T[2] * CT[4]) + 0.3f;
    o.b = o.T[0] * CT[1] + o.T[1] * CT[3] + o.T[2] * CT[5];
    o.c = (o.T[3] * CT[1] + o.T[4] * CT[3] + o.T[5] * CT[5]) + 0.3f;
    o.det_raw = o.a * o.c - o.b * o.b;
    o.det = fmaxf(o.det_raw, 1e-6f);
    // project_pix, gsplat_cpu.cpp:119-122
#pragma unroll
    for (int i = 0; i < 4; i++)
        o.ph[i] = cam.pm[4 * i + 0] * mean[0] + cam.pm[4 * i + 1] * mean[1] +
                  cam.pm[4 * i + 2] * mean[2] + cam.pm[4 * i + 3];
    o.rw = 1.0f / fmaxf(o.ph[3], 1e-6f);
}

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
    const float4 q4 = reinterpret_cast<const float4 *>(quats)[n];
    float quat[4] = {q4.x, q4.y, q4.z, q4.w};
    Proj o;
    project_one(cam, mean, scale, quat, o);

    float conic0 = o.c / o.det, conic1 = -o.b / o.det, conic2 = o.a / o.det;
    float bb = (o.a + o.c) / 2.0f;
    float sq = sqrtf(fmaxf(bb * bb - o.det, 0.1f));
    float radius = ceilf(3.0f * sqrtf(fmaxf(bb + sq, bb - sq)));
    float px = o.ph[0] * o.rw, py = o.ph[1] * o.rw;
    // CPU pixel-centre formula (gsplat_cpu.cpp:123-124) plus the principal-point offset the GPU
    // path honours (helpers.cuh:13-15); the offset is exactly 0 when cx == W/2, cy == H/2.
    float u = 0.5f * ((px + 1.0f) * (float)cam.W - 1.0f) + (cam.cx - 0.5f * (float)cam.W);
    float v = 0.5f * ((py + 1.0f) * (float)cam.H - 1.0f) + (cam.cy - 0.5f * (float)cam.H);

    bool visible = o.p[2] > cam.clip;  // clip_near_plane, helpers.cuh:225-233
    int rad = visible ? (int)fminf(radius, 2.0e9f) : 0;
    int tiles = 0;
    if (visible) {
        PixRect r = pixel_rect(u, v, o.a, o.c, cam.W, cam.H);
        tiles = rect_tiles(r);
    }
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
    float4 *vq4 = reinterpret_cast<float4 *>(v_quats);
    if (radii[n] <= 0) {  // backward.cu:380-382: culled Gaussians get no gradient
        v_means[3 * n] = v_means[3 * n + 1] = v_means[3 * n + 2] = 0.0f;
        v_scales[3 * n] = v_scales[3 * n + 1] = v_scales[3 * n + 2] = 0.0f;
        vq4[n] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    float mean[3] = {means[3 * n], means[3 * n + 1], means[3 * n + 2]};
    float scale[3] = {scales[3 * n], scales[3 * n + 1], scales[3 * n + 2]};
    if (cam.flags & GS_CAM_LOG_SCALES) {
#pragma unroll
        for (int j = 0; j < 3; j++) scale[j] = expf(scale[j]);
    }
    const float4 q4 = reinterpret_cast<const float4 *>(quats)[n];
    float quat[4] = {q4.x, q4.y, q4.z, q4.w};
    Proj o;
    project_one(cam, mean, scale, quat, o);
    const float *vm = cam.vm;
    const float *pm = cam.pm;

    float vmean[3];
    // xys <- pHom
    float vpx = 0.5f * (float)cam.W * v_xy[2 * n + 0];
    float vpy = 0.5f * (float)cam.H * v_xy[2 * n + 1];
    float vh0 = vpx * o.rw, vh1 = vpy * o.rw;
    float vrw = vpx * o.ph[0] + vpy * o.ph[1];
    float vh3 = (o.ph[3] >= 1e-6f) ? -o.rw * o.rw * vrw : 0.0f;
#pragma unroll
    for (int j = 0; j < 3; j++) vmean[j] = pm[j] * vh0 + pm[4 + j] * vh1 + pm[12 + j] * vh3;

    // conic <- cov2d
    float A = o.c / o.det, B = -o.b / o.det, C = o.a / o.det;
    float vA = v_conic[3 * n + 0], vB = v_conic[3 * n + 1], vC = v_conic[3 * n + 2];
    float va, vb, vc;
    if (o.det_raw > 1e-6f) {
        va = -A * A * vA - A * B * vB - B * B * vC;
        vb = -2.0f * A * B * vA - (A * C + B * B) * vB - 2.0f * B * C * vC;
        vc = -B * B * vA - B * C * vB - C * C * vC;
    } else {
        va = vC / o.det;
        vb = -vB / o.det;
        vc = vA / o.det;
    }
    float S00 = 2.0f * va, S01 = vb, S11 = 2.0f * vc;  // V + V^T
    // TC = T * S3 (2x3); vT = S * TC; ST = S * T; G = T^T * ST; vM = G * M
    float TC[6], vT[6], ST[6];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            TC[3 * i + j] = o.T[3 * i + 0] * o.S3[j] + o.T[3 * i + 1] * o.S3[3 + j] +
                            o.T[3 * i + 2] * o.S3[6 + j];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        vT[j] = S00 * TC[j] + S01 * TC[3 + j];
        vT[3 + j] = S01 * TC[j] + S11 * TC[3 + j];
        ST[j] = S00 * o.T[j] + S01 * o.T[3 + j];
        ST[3 + j] = S01 * o.T[j] + S11 * o.T[3 + j];
    }
    float G[9], vM[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) G[3 * i + j] = o.T[i] * ST[j] + o.T[3 + i] * ST[3 + j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            vM[3 * i + j] = G[3 * i + 0] * o.M[j] + G[3 * i + 1] * o.M[3 + j] + G[3 * i + 2] * o.M[6 + j];
    float vR[9];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            acc += o.Rq[3 * i + j] * vM[3 * i + j];
            vR[3 * i + j] = vM[3 * i + j] * cam.glob * scale[j];
        }
        // d exp(ls) / d ls = exp(ls) when the input was a log-scale
        v_scales[3 * n + j] = cam.glob * acc * ((cam.flags & GS_CAM_LOG_SCALES) ? scale[j] : 1.0f);
    }
    float w = o.u[0], x = o.u[1], y = o.u[2], z = o.u[3];
    float vu0 = 2.0f * (-z * vR[1] + y * vR[2] + z * vR[3] - x * vR[5] - y * vR[6] + x * vR[7]);
    float vu1 = 2.0f * (y * vR[1] + z * vR[2] + y * vR[3] - 2.0f * x * vR[4] - w * vR[5] +
                        z * vR[6] + w * vR[7] - 2.0f * x * vR[8]);
    float vu2 = 2.0f * (-2.0f * y * vR[0] + x * vR[1] + w * vR[2] + x * vR[3] + z * vR[5] -
                        w * vR[6] + z * vR[7] - 2.0f * y * vR[8]);
    float vu3 = 2.0f * (-2.0f * z * vR[0] - w * vR[1] + x * vR[2] + w * vR[3] - 2.0f * z * vR[4] +
                        y * vR[5] + x * vR[6] + y * vR[7]);
    float dotuv = w * vu0 + x * vu1 + y * vu2 + z * vu3;
    vq4[n] = make_float4((vu0 - w * dotuv) / o.qn, (vu1 - x * dotuv) / o.qn,
                         (vu2 - y * dotuv) / o.qn, (vu3 - z * dotuv) / o.qn);

    // T = J * Rview: vJ = vT * Rview^T (only J00, J02, J11, J12 are live)
    float vJ00 = vT[0] * vm[0] + vT[1] * vm[1] + vT[2] * vm[2];
    float vJ02 = vT[0] * vm[8] + vT[1] * vm[9] + vT[2] * vm[10];
    float vJ11 = vT[3] * vm[4] + vT[4] * vm[5] + vT[5] * vm[6];
    float vJ12 = vT[3] * vm[8] + vT[4] * vm[9] + vT[5] * vm[10];
    float vrz = cam.fx * vJ00 + cam.fy * vJ11 +
                2.0f * o.rz * (-cam.fx * o.t0 * vJ02 - cam.fy * o.t1 * vJ12);
    float vt0 = -cam.fx * o.rz2 * vJ02;
    float vt1 = -cam.fy * o.rz2 * vJ12;
    float vp0 = 0.0f, vp1 = 0.0f, vp2 = -o.rz2 * vrz;
    float limx = 1.3f * (0.5f * (float)cam.W / cam.fx);
    float limy = 1.3f * (0.5f * (float)cam.H / cam.fy);
    if (o.clx == 0) vp0 = vt0; else vp2 += (o.clx > 0 ? limx : -limx) * vt0;
    if (o.cly == 0) vp1 = vt1; else vp2 += (o.cly > 0 ? limy : -limy) * vt1;
    if (v_depth) vp2 += v_depth[n];
#pragma unroll
    for (int j = 0; j < 3; j++)
        v_means[3 * n + j] = vmean[j] + (vm[j] * vp0 + vm[4 + j] * vp1 + vm[8 + j] * vp2);
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
    hipLaunchKernelGGL(gs::k_project_forward, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a,
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
    hipLaunchKernelGGL(gs::k_project_backward, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a,
                       viewmat_dev, projmat_dev, N, means, scales, quats, radii, v_xy, v_depth, v_conic, v_means, v_scales,
                       v_quats);
    GS_LAUNCH_CHECK();
    return GS_OK;
}
