// gs_gaussian.h — the per-Gaussian device functions shared by the stage kernels (gs_project.hip,
// gs_sh.hip, gs_bin.hip) and by the fused per-Gaussian kernels (gs_fused.hip): projection and its
// VJP, SH basis, view direction, the packed compositing record.  One definition each, so that the
// fused and the operator-granular paths compute identical bits.
#pragma once

#include "gs_device.h"

namespace gs {

// 16-byte vector access through a pointer that is only 4-byte aligned (rows of 45 floats, slices of
// a flat parameter buffer): gfx950 global loads / stores allow it, the type tells the compiler.
typedef float float4_u __attribute__((ext_vector_type(4), aligned(4)));

struct CamArgs {
    float vm[12];  // rows 0..2 of viewmat
    float pm[16];
    float fx, fy, cx, cy;
    int W, H;
    float clip, glob;
    uint32_t flags;
};

static inline CamArgs make_cam(const GsCamera *c) {
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

// Outputs of the forward projection of one Gaussian (operator surface of ProjectGaussians).
struct ProjOut {
    float u, v;          // pixel centre
    float conic[3];
    int radius, tiles;   // GPU-surface radius / CPU-rectangle tile count (0 when near-plane culled)
};

__device__ __forceinline__ void project_outputs(const CamArgs &cam, const Proj &o, ProjOut &r) {
    r.conic[0] = o.c / o.det;
    r.conic[1] = -o.b / o.det;
    r.conic[2] = o.a / o.det;
    float bb = (o.a + o.c) / 2.0f;
    float sq = sqrtf(fmaxf(bb * bb - o.det, 0.1f));
    float radius = ceilf(3.0f * sqrtf(fmaxf(bb + sq, bb - sq)));
    float px = o.ph[0] * o.rw, py = o.ph[1] * o.rw;
    // CPU pixel-centre formula (gsplat_cpu.cpp:123-124) plus the principal-point offset the GPU
    // path honours (helpers.cuh:13-15); the offset is exactly 0 when cx == W/2, cy == H/2.
    r.u = 0.5f * ((px + 1.0f) * (float)cam.W - 1.0f) + (cam.cx - 0.5f * (float)cam.W);
    r.v = 0.5f * ((py + 1.0f) * (float)cam.H - 1.0f) + (cam.cy - 0.5f * (float)cam.H);
    bool visible = o.p[2] > cam.clip;  // clip_near_plane, helpers.cuh:225-233
    r.radius = visible ? (int)fminf(radius, 2.0e9f) : 0;
    r.tiles = 0;
    if (visible) {
        PixRect pr = pixel_rect(r.u, r.v, o.a, o.c, cam.W, cam.H);
        r.tiles = rect_tiles(pr);
    }
}

// VJP of the projection of one Gaussian; derivation in DESIGN.md §6.  `scale` is the scale the
// forward used (after exp() when the input held log-scales); v_scale is w.r.t. the INPUT.
struct ProjGrad { float v_mean[3], v_scale[3], v_quat[4]; };

__device__ __forceinline__ void project_backward_one(const CamArgs &cam, const Proj &o,
                                                     const float *scale, float vxy0, float vxy1,
                                                     float vA, float vB, float vC, float v_depth,
                                                     ProjGrad &g) {
    const float *vm = cam.vm;
    const float *pm = cam.pm;

    float vmean[3];
    // xys <- pHom
    float vpx = 0.5f * (float)cam.W * vxy0;
    float vpy = 0.5f * (float)cam.H * vxy1;
    float vh0 = vpx * o.rw, vh1 = vpy * o.rw;
    float vrw = vpx * o.ph[0] + vpy * o.ph[1];
    float vh3 = (o.ph[3] >= 1e-6f) ? -o.rw * o.rw * vrw : 0.0f;
#pragma unroll
    for (int j = 0; j < 3; j++) vmean[j] = pm[j] * vh0 + pm[4 + j] * vh1 + pm[12 + j] * vh3;

    // conic <- cov2d
    float A = o.c / o.det, B = -o.b / o.det, C = o.a / o.det;
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
        g.v_scale[j] = cam.glob * acc * ((cam.flags & GS_CAM_LOG_SCALES) ? scale[j] : 1.0f);
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
    g.v_quat[0] = (vu0 - w * dotuv) / o.qn;
    g.v_quat[1] = (vu1 - x * dotuv) / o.qn;
    g.v_quat[2] = (vu2 - y * dotuv) / o.qn;
    g.v_quat[3] = (vu3 - z * dotuv) / o.qn;

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
    vp2 += v_depth;
#pragma unroll
    for (int j = 0; j < 3; j++)
        g.v_mean[j] = vmean[j] + (vm[j] * vp0 + vm[4 + j] * vp1 + vm[8 + j] * vp2);
}

// The packed 48-byte record of one Gaussian (layout and sigma_max / rectangle logic: gs_bin.hip,
// "pack").  Returns the number of tiles the tightened rectangle touches.
__device__ __forceinline__ int pack_one(int W, int H, float x, float y, float A, float B, float C,
                                        bool have_cov, float cxx, float cyy, float opacity_in,
                                        int radius, float c0, float c1, float c2, uint32_t flags,
                                        float4 &p0, float4 &p1, float4 &p2) {
    const float det = A * C - B * B;
    if (!have_cov) {
        // conic = cov2d^-1  ->  cov2d = conic^-1: xx = C / det, yy = A / det
        cxx = C / det;
        cyy = A / det;
    }
    PixRect r = pixel_rect(x, y, cxx, cyy, W, H);
    float opac = opacity_in;
    if (flags & GS_FLAG_LOGIT_OPACITY) opac = 1.0f / (1.0f + expf(-opac));  // torch::sigmoid, model.cpp:215
    // conservative w.r.t. rounding of the log, the exp and the product opacity*exp(-sigma)
    float smax = (opac > 0.0f) ? (logf(255.0f * opac) + 2.0e-3f) : -1.0f;
    uint32_t binding = 1u;
    // trust the ellipse box only for a well-conditioned, positive-definite conic
    if (smax >= 0.0f && A > 0.0f && C > 0.0f && det > 1.0e-4f * (A * C) && det < 3.0e38f) {
        const float k2 = 2.0f * smax / det;
        const float hx = sqrtf(k2 * C) * 1.001f + 1.0e-3f;
        const float hy = sqrtf(k2 * A) * 1.001f + 1.0e-3f;
        PixRect e;
        e.x0 = max(0, f2i_sat(ceilf(x - hx)));
        e.x1 = min(W, f2i_sat(floorf(x + hx)) + 1);
        e.y0 = max(0, f2i_sat(ceilf(y - hy)));
        e.y1 = min(H, f2i_sat(floorf(y + hy)) + 1);
        PixRect t;
        t.x0 = max(r.x0, e.x0); t.x1 = min(r.x1, e.x1);
        t.y0 = max(r.y0, e.y0); t.y1 = min(r.y1, e.y1);
        binding = (t.x0 != e.x0 || t.x1 != e.x1 || t.y0 != e.y0 || t.y1 != e.y1) ? 1u : 0u;
        r = t;
    }
    int tiles = (radius > 0 && smax >= 0.0f) ? rect_tiles(r) : 0;
    if (tiles == 0) r.x0 = r.x1 = r.y0 = r.y1 = 0;
    uint32_t rx = (uint32_t)r.x0 | ((uint32_t)r.x1 << 16);
    uint32_t ry = (uint32_t)r.y0 | ((uint32_t)r.y1 << 16);
    smax = __uint_as_float((__float_as_uint(smax) & ~1u) | binding);
    p0 = make_float4(x, y, A, B);
    p1 = make_float4(C, opac, smax, __uint_as_float(rx));
    p2 = make_float4(c0, c1, c2, __uint_as_float(ry));
    return tiles;
}

__device__ __forceinline__ void sh_basis(int nb, float x, float y, float z, float *r) {
    // gsplat_cpu.cpp:436-483; r[] has 25 slots, entries >= nb stay 0
#pragma unroll
    for (int i = 0; i < 25; i++) r[i] = 0.0f;
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    r[0] = C0;
    if (nb <= 1) return;
    r[1] = C1 * -y;
    r[2] = C1 * z;
    r[3] = C1 * -x;
    if (nb <= 4) return;
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    r[4] = 1.0925484305920792f * xy;
    r[5] = -1.0925484305920792f * yz;
    r[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
    r[7] = -1.0925484305920792f * xz;
    r[8] = 0.5462742152960396f * (xx - yy);
    if (nb <= 9) return;
    r[9] = -0.5900435899266435f * y * (3.0f * xx - yy);
    r[10] = 2.890611442640554f * xy * z;
    r[11] = -0.4570457994644658f * y * (4.0f * zz - xx - yy);
    r[12] = 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
    r[13] = -0.4570457994644658f * x * (4.0f * zz - xx - yy);
    r[14] = 1.445305721320277f * z * (xx - yy);
    r[15] = -0.5900435899266435f * x * (xx - 3.0f * yy);
    if (nb <= 16) return;
    r[16] = 2.5033429417967046f * xy * (xx - yy);
    r[17] = -1.7701307697799304f * yz * (3.0f * xx - yy);
    r[18] = 0.9461746957575601f * xy * (7.0f * zz - 1.0f);
    r[19] = -0.6690465435572892f * yz * (7.0f * zz - 3.0f);
    r[20] = 0.10578554691520431f * (zz * (35.0f * zz - 30.0f) + 3.0f);
    r[21] = -0.6690465435572892f * xz * (7.0f * zz - 3.0f);
    r[22] = 0.47308734787878004f * (xx - yy) * (7.0f * zz - 1.0f);
    r[23] = -1.7701307697799304f * xz * (xx - 3.0f * yy);
    r[24] = 0.6258357354491761f * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy));
}

__host__ __device__ inline int num_bases(int degree) {  // gsplat_cpu.cpp:409-422
    return degree == 0 ? 1 : degree == 1 ? 4 : degree == 2 ? 9 : degree == 3 ? 16 : 25;
}

// 4 waves per block (2 for K = 25 so the padded slabs stay under 64 KiB of dynamic LDS);
// each wave owns 64 consecutive Gaussians.

__device__ __forceinline__ void view_dir(const float *__restrict__ means, int64_t g, float cx,
                                         float cy, float cz, float &x, float &y, float &z) {
    // (means - T) / ||means - T||, model.cpp:176-177 (torch::norm, no epsilon)
    x = means[3 * g] - cx;
    y = means[3 * g + 1] - cy;
    z = means[3 * g + 2] - cz;
    const float n = sqrtf(x * x + y * y + z * z);
    x /= n; y /= n; z /= n;
}

// Body of the K = 16 split-coefficient SH forward, four lanes per Gaussian (t = 4 * Gaussian + lane of
// the quad; comments: gs_sh.hip, k_sh_forward_fused16_quad).  colors (nullable) receives
// clamp_min(rgb + 0.5, 0) at colors[color_stride * g + 0..2] — stride 3 for a colour tensor, 12 with
// colors = packed + 8 to drop it straight into the packed record.
__device__ __forceinline__ void sh_forward16_quad_body(int64_t t, int N, int nb, const float *__restrict__ means,
                                                       float cx, float cy, float cz,
                                                       const float *__restrict__ cp_dev,
                                                       const float *__restrict__ dc,
                                                       const float *__restrict__ rest,
                                                       float *__restrict__ colors, int color_stride,
                                                       float *__restrict__ rgb_raw) {
    const int64_t g = t >> 2;
    const int q = (int)(t & 3);
    if (g >= N) return;  // whole quads drop out together
    // the mean first: its consumer (the view direction) is waited for with the rows still in flight
    const float mx = means[3 * g], my = means[3 * g + 1], mz = means[3 * g + 2];
    const float *row = rest + g * 45 + 12 * q;
    // {A.r A.g A.b B.r | B.g B.b C.r C.g | C.b D.r D.g D.b}; lane 3: {.. | .. | C.g* C.b* C.b D?} -> c.w = float 44
    const float4_u a = *reinterpret_cast<const float4_u *>(row);
    const float4_u b = *reinterpret_cast<const float4_u *>(row + 4);
    const float4_u c = *reinterpret_cast<const float4_u *>(row + (q == 3 ? 5 : 8));
    const float d0 = dc[3 * g], d1 = dc[3 * g + 1], d2 = dc[3 * g + 2];
    if (cp_dev) { cx = cp_dev[0]; cy = cp_dev[1]; cz = cp_dev[2]; }
    float x = mx - cx, y = my - cy, z = mz - cz;       // view_dir(), on the values loaded above
    const float nrm = sqrtf(x * x + y * y + z * z);
    x /= nrm; y /= nrm; z /= nrm;
    float r[25];
    sh_basis(nb, x, y, z, r);
    const bool l3 = q == 3;
    const float rA = q == 0 ? r[1] : q == 1 ? r[5] : q == 2 ? r[9] : r[13];
    const float rB = q == 0 ? r[2] : q == 1 ? r[6] : q == 2 ? r[10] : r[14];
    const float rC = q == 0 ? r[3] : q == 1 ? r[7] : q == 2 ? r[11] : r[15];
    const float rD = q == 0 ? r[4] : q == 1 ? r[8] : q == 2 ? r[12] : r[0];
    // third coefficient's blue: float 8 of the lane's span (c.x), for lane 3 float 44 = c.w;
    // fourth coefficient: floats 9 .. 11 (c.y c.z c.w), for lane 3 the dc term
    const float Cb = l3 ? c.w : c.x;
    const float D0 = l3 ? d0 : c.y, D1 = l3 ? d1 : c.z, D2 = l3 ? d2 : c.w;
    float c0 = rA * a.x + rB * a.w + rC * b.z + rD * D0;
    float c1 = rA * a.y + rB * b.x + rC * b.w + rD * D1;
    float c2 = rA * a.z + rB * b.y + rC * Cb + rD * D2;
    c0 += dpp_f<0xB1>(c0); c1 += dpp_f<0xB1>(c1); c2 += dpp_f<0xB1>(c2);  // quad_perm [1,0,3,2]
    c0 += dpp_f<0x4E>(c0); c1 += dpp_f<0x4E>(c1); c2 += dpp_f<0x4E>(c2);  // quad_perm [2,3,0,1]
    if (q == 0) {
        rgb_raw[3 * g + 0] = c0;
        rgb_raw[3 * g + 1] = c1;
        rgb_raw[3 * g + 2] = c2;
        if (colors) {
            colors[color_stride * g + 0] = fmaxf(c0 + 0.5f, 0.0f);
            colors[color_stride * g + 1] = fmaxf(c1 + 0.5f, 0.0f);
            colors[color_stride * g + 2] = fmaxf(c2 + 0.5f, 0.0f);
        }
    }
}

// Split-coefficient SH kernels (features_dc + features_rest): LDS slab geometry per K.
template <int K>
struct ShSplit {
    static constexpr int ROW = 3 * (K - 1);
    static constexpr int ROWP = (ROW == 0) ? 1 : (ROW | 1);
    static constexpr int kBlock = (K > 16) ? 128 : 256;
};

static inline int deg_from_bases(int K) {  // spherical_harmonics.cpp:3-16, but strict
    switch (K) {
    case 1: return 0;
    case 4: return 1;
    case 9: return 2;
    case 16: return 3;
    case 25: return 4;
    default: return -1;
    }
}


// gs_sh.hip: the K = 16 split-coefficient SH forward, four lanes per Gaussian, no LDS (colors may be
// NULL: only the raw rgb is written).
int launch_sh_forward_fused16_quad(int N, int nb, const float *means, const float *cam_pos,
                                   const float *features_dc, const float *features_rest,
                                   float *colors, float *rgb_raw, hipStream_t s);

}  // namespace gs
