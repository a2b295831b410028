/*
 * gsplat_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded, fp32 restatement of OpenSplat's CPU rasterizer
 * (reference: /root/reference/rasterizer/gsplat-cpu/gsplat_cpu.cpp, VERSION 1.1.5), written from
 * the reference's behaviour — no reference source is copied.  It is the checker the HIP kernels
 * are compared against in tests/, in __graft_entry__.smoke() and in bench.py's cpu_baseline leg.
 * Nothing on the product path may import, link or call it.
 *
 * Parity status: PINNED.  Every function here is checked against the reference's own code,
 * compiled in place into oracle/_ref/libgsplat_ref.so (oracle/Makefile, oracle/ref_shim.cpp), by
 * tests/test_oracle_vs_reference.py, and against the golden vectors under tests/golden/ that
 * were generated from that build by tests/golden/make_golden.py.  The reference itself ships no
 * tests or golden vectors.
 *
 * The compositing loops (orc_rasterize_*) are bit-for-bit restatements: same operation order,
 * no FMA contraction (built with -ffp-contract=off), libm expf.  The projection and SH functions
 * restate batched torch ops whose internal summation order is unspecified, so they agree with
 * the reference to fp32 round-off (a few ulp), not bitwise.
 *
 * All pointers are host pointers to contiguous row-major fp32 / int32 arrays.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* quaternion (w,x,y,z) -> rotation matrix; gsplat_cpu.cpp:16-40 (normalises with F.normalize,
 * eps 1e-12) */
static void quat_to_rot(const float *q, float R[9]) {
    float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < 1e-12f) n = 1e-12f;
    float w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
    R[0] = 1.0f - 2.0f * (y * y + z * z);
    R[1] = 2.0f * (x * y - w * z);
    R[2] = 2.0f * (x * z + w * y);
    R[3] = 2.0f * (x * y + w * z);
    R[4] = 1.0f - 2.0f * (x * x + z * z);
    R[5] = 2.0f * (y * z - w * x);
    R[6] = 2.0f * (x * z - w * y);
    R[7] = 2.0f * (y * z + w * x);
    R[8] = 1.0f - 2.0f * (x * x + y * y);
}

static float clampf(float v, float lo, float hi) { return fminf(hi, fmaxf(lo, v)); }

/* Intermediate quantities of one Gaussian's projection, shared by forward and backward. */
typedef struct {
    float p[3];      /* view-space position                       gsplat_cpu.cpp:68-70 */
    float Rq[9];     /* rotation from quat                        :74                   */
    float M[9];      /* Rq * glob_scale * scales                  :75                   */
    float cov3d[9];  /* M M^T                                     :76                   */
    float t[3];      /* FOV-clamped view position                 :79-85                */
    int clampx, clampy; /* -1 / 0 / +1: which side of the clamp is active */
    float rz, rz2;
    float J[6];      /* 2x3                                       :89-92                */
    float T[6];      /* J * Rclip (2x3)                           :94                   */
    float a, b, c;   /* cov2d entries incl. blur                  :95-99                */
    float b10;       /* cov2d[1][0] (equals b up to round-off)                          */
    float det_raw, det;
    float phom[4], rw;
} ProjTmp;

static void project_one(const float *mean, const float *scale, float glob_scale, const float *quat,
                        const float *vm, const float *pm, float fx, float fy, int H, int W,
                        ProjTmp *o) {
    float fovx = 0.5f * (float)W / fx; /* :64-65 */
    float fovy = 0.5f * (float)H / fy;
    for (int i = 0; i < 3; i++)
        o->p[i] = vm[4 * i + 0] * mean[0] + vm[4 * i + 1] * mean[1] + vm[4 * i + 2] * mean[2] +
                  vm[4 * i + 3];
    quat_to_rot(quat, o->Rq);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) o->M[3 * i + j] = o->Rq[3 * i + j] * glob_scale * scale[j];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            o->cov3d[3 * i + j] = o->M[3 * i + 0] * o->M[3 * j + 0] +
                                  o->M[3 * i + 1] * o->M[3 * j + 1] +
                                  o->M[3 * i + 2] * o->M[3 * j + 2];
    float limx = 1.3f * fovx, limy = 1.3f * fovy;
    float xz = o->p[0] / o->p[2], yz = o->p[1] / o->p[2];
    o->clampx = (xz > limx) ? 1 : ((xz < -limx) ? -1 : 0);
    o->clampy = (yz > limy) ? 1 : ((yz < -limy) ? -1 : 0);
    o->t[0] = o->p[2] * clampf(xz, -limx, limx);
    o->t[1] = o->p[2] * clampf(yz, -limy, limy);
    o->t[2] = o->p[2];
    o->rz = 1.0f / o->t[2];
    o->rz2 = o->rz * o->rz;
    o->J[0] = fx * o->rz;
    o->J[1] = 0.0f;
    o->J[2] = -fx * o->t[0] * o->rz2;
    o->J[3] = 0.0f;
    o->J[4] = fy * o->rz;
    o->J[5] = -fy * o->t[1] * o->rz2;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++)
            o->T[3 * i + j] = o->J[3 * i + 0] * vm[0 + j] + o->J[3 * i + 1] * vm[4 + j] +
                              o->J[3 * i + 2] * vm[8 + j];
    float CT[6]; /* cov3d * T^T, 3x2 */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 2; j++)
            CT[2 * i + j] = o->cov3d[3 * i + 0] * o->T[3 * j + 0] +
                            o->cov3d[3 * i + 1] * o->T[3 * j + 1] +
                            o->cov3d[3 * i + 2] * o->T[3 * j + 2];
    float c2[4];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            c2[2 * i + j] =
                o->T[3 * i + 0] * CT[0 + j] + o->T[3 * i + 1] * CT[2 + j] + o->T[3 * i + 2] * CT[4 + j];
    o->a = c2[0] + 0.3f;
    o->b = c2[1];
    o->b10 = c2[2];
    o->c = c2[3] + 0.3f;
    o->det_raw = o->a * o->c - o->b * o->b; /* :103 */
    o->det = fmaxf(o->det_raw, 1e-6f);      /* :104 */
    for (int i = 0; i < 4; i++)
        o->phom[i] = pm[4 * i + 0] * mean[0] + pm[4 * i + 1] * mean[1] + pm[4 * i + 2] * mean[2] +
                     pm[4 * i + 3] * 1.0f;
    o->rw = 1.0f / fmaxf(o->phom[3], 1e-6f); /* :121 */
}

/* gsplat_cpu.cpp:48-131.  cov2d: N x 2 x 2.  depths_view / cov3d6 (N x 6, upper triangle) are
 * extra outputs (may be NULL) that the GPU operator surface exposes (forward.cu:95, :462-469). */
int orc_project_forward(int N, const float *means, const float *scales, float glob_scale,
                        const float *quats, const float *viewmat, const float *projmat, float fx,
                        float fy, float cx, float cy, int H, int W, float clip, float *xys,
                        int32_t *radii, float *conics, float *cov2d, float *cam_depths,
                        float *depths_view, float *cov3d6, float *depth_keys_as_read) {
    (void)cx; (void)cy; (void)clip; /* the CPU path ignores them, :58-62,:71,:123-124 */
    /* REFERENCE QUIRK (documented in DESIGN.md, P11): camDepths is the strided view
     * pProj[..., 2] (:128) but the CPU rasterizer indexes its data_ptr() with unit stride
     * (:152,:157-158), so the key it actually sorts Gaussian a by is element a+2 of the
     * flattened [N,3] pProj array, not Gaussian a's depth.  depth_keys_as_read (nullable)
     * returns those keys so tests can reproduce the reference's end-to-end chain exactly. */
    float *pproj_flat = depth_keys_as_read ? (float *)malloc(sizeof(float) * 3 * (size_t)(N + 1)) : 0;
    for (int n = 0; n < N; n++) {
        ProjTmp t;
        project_one(means + 3 * n, scales + 3 * n, glob_scale, quats + 4 * n, viewmat, projmat, fx,
                    fy, H, W, &t);
        conics[3 * n + 0] = t.c / t.det;
        conics[3 * n + 1] = -t.b / t.det;
        conics[3 * n + 2] = t.a / t.det;
        float bb = (t.a + t.c) / 2.0f;
        float sq = sqrtf(fmaxf(bb * bb - t.det, 0.1f));
        float v1 = bb + sq, v2 = bb - sq;
        float radius = ceilf(3.0f * sqrtf(fmaxf(v1, v2)));
        radii[n] = (int32_t)radius;
        float px = t.phom[0] * t.rw, py = t.phom[1] * t.rw, pz = t.phom[2] * t.rw;
        xys[2 * n + 0] = 0.5f * ((px + 1.0f) * (float)W - 1.0f);
        xys[2 * n + 1] = 0.5f * ((py + 1.0f) * (float)H - 1.0f);
        cov2d[4 * n + 0] = t.a;
        cov2d[4 * n + 1] = t.b;
        cov2d[4 * n + 2] = t.b10;
        cov2d[4 * n + 3] = t.c;
        cam_depths[n] = pz;
        if (pproj_flat) {
            pproj_flat[3 * n + 0] = px;
            pproj_flat[3 * n + 1] = py;
            pproj_flat[3 * n + 2] = pz;
        }
        if (depths_view) depths_view[n] = t.p[2];
        if (cov3d6) {
            cov3d6[6 * n + 0] = t.cov3d[0];
            cov3d6[6 * n + 1] = t.cov3d[1];
            cov3d6[6 * n + 2] = t.cov3d[2];
            cov3d6[6 * n + 3] = t.cov3d[4];
            cov3d6[6 * n + 4] = t.cov3d[5];
            cov3d6[6 * n + 5] = t.cov3d[8];
        }
    }
    if (pproj_flat) {
        for (int n = 0; n < N; n++) depth_keys_as_read[n] = pproj_flat[n + 2];
        free(pproj_flat);
    }
    return 0;
}

/* The two behaviours the product takes from the reference's GPU path rather than from gsplat-cpu
 * (DESIGN.md P2, P3), restated so that they can be checked; applied IN PLACE to the outputs of
 * orc_project_forward:
 *   near-plane cull  rasterizer/gsplat/forward.cu:49-52 + helpers.cuh:225-233: p_view.z <= clip_thresh
 *                    -> radii = 0, num_tiles_hit = 0, the Gaussian takes no part in the frame
 *                    (visible[n] = 0; the caller leaves it out of the compositing, its gradients are 0);
 *   principal point  helpers.cuh:13-15,112-122: pixel = 0.5 W x_ndc + cx - 0.5, which is gsplat-cpu's
 *                    0.5 ((x_ndc + 1) W - 1) (gsplat_cpu.cpp:123-124; it ignores cx, cy) moved by
 *                    (cx - W/2): xys += (cx - W/2, cy - H/2), the offset being exactly 0 for a centred
 *                    principal point.
 * xys_gpu_formula (nullable) receives helpers.cuh's own expression, rw = 1 / (w + 1e-6) included, so
 * that a test can bound the distance between the two forms. */
int orc_project_gpu_semantics(int N, const float *means, const float *viewmat, const float *projmat,
                              float cx, float cy, int H, int W, float clip, float *xys,
                              int32_t *radii, int32_t *visible, float *xys_gpu_formula) {
    const float *vm = viewmat, *pm = projmat;
    for (int n = 0; n < N; n++) {
        const float *m = means + 3 * n;
        float pz = vm[8] * m[0] + vm[9] * m[1] + vm[10] * m[2] + vm[11];   /* transform_4x3, row 2 */
        int vis = !(pz <= clip);
        visible[n] = vis;
        if (!vis) radii[n] = 0;
        xys[2 * n + 0] = xys[2 * n + 0] + (cx - 0.5f * (float)W);
        xys[2 * n + 1] = xys[2 * n + 1] + (cy - 0.5f * (float)H);
        if (xys_gpu_formula) {
            float h[4];
            for (int i = 0; i < 4; i++)
                h[i] = pm[4 * i + 0] * m[0] + pm[4 * i + 1] * m[1] + pm[4 * i + 2] * m[2] + pm[4 * i + 3];
            float rw = 1.0f / (h[3] + 1e-6f);
            xys_gpu_formula[2 * n + 0] = 0.5f * (float)W * (h[0] * rw) + cx - 0.5f;
            xys_gpu_formula[2 * n + 1] = 0.5f * (float)H * (h[1] * rw) + cy - 0.5f;
        }
    }
    return 0;
}

/* VJP of orc_project_forward w.r.t. (means, scales, quats) for cotangents on xys and conics —
 * what libtorch autograd computes through gsplat_cpu.cpp:48-131 (the CPU path has no hand-written
 * backward; project_gaussians.cpp:94-123 is a plain differentiable function).  v_depth_view
 * (nullable) is the extra view-space-depth cotangent of the GPU surface (backward.cu:395-398). */
int orc_project_backward(int N, const float *means, const float *scales, float glob_scale,
                         const float *quats, const float *viewmat, const float *projmat, float fx,
                         float fy, float cx, float cy, int H, int W, float clip,
                         const float *v_xys, const float *v_conics, const float *v_depth_view,
                         float *v_means, float *v_scales, float *v_quats) {
    (void)cx; (void)cy; (void)clip;
    const float *vm = viewmat, *pm = projmat;
    for (int n = 0; n < N; n++) {
        ProjTmp t;
        const float *q = quats + 4 * n, *s = scales + 3 * n;
        project_one(means + 3 * n, s, glob_scale, q, vm, pm, fx, fy, H, W, &t);
        float vmean[3] = {0, 0, 0};

        /* xys <- pHom (:119-125) */
        float vpx = 0.5f * (float)W * v_xys[2 * n + 0];
        float vpy = 0.5f * (float)H * v_xys[2 * n + 1];
        float vh[4] = {vpx * t.rw, vpy * t.rw, 0.0f, 0.0f};
        float vrw = vpx * t.phom[0] + vpy * t.phom[1];
        if (t.phom[3] > 1e-6f) vh[3] = -t.rw * t.rw * vrw;
        for (int j = 0; j < 3; j++)
            vmean[j] += pm[0 + j] * vh[0] + pm[4 + j] * vh[1] + pm[8 + j] * vh[2] + pm[12 + j] * vh[3];

        /* conic <- cov2d (:103-109) */
        float A = t.c / t.det, B = -t.b / t.det, C = t.a / t.det;
        float vA = v_conics[3 * n + 0], vB = v_conics[3 * n + 1], vC = v_conics[3 * n + 2];
        float va, vb, vc;
        if (t.det_raw > 1e-6f) {
            va = -A * A * vA - A * B * vB - B * B * vC;
            vb = -2.0f * A * B * vA - (A * C + B * B) * vB - 2.0f * B * C * vC;
            vc = -B * B * vA - B * C * vB - C * C * vC;
        } else { /* det clamped: conic = (c, -b, a) / eps */
            va = vC / t.det;
            vb = -vB / t.det;
            vc = vA / t.det;
        }
        /* cov2d = T cov3d T^T : v_T = (V + V^T) T cov3d, v_cov3d(sym) = T^T V T */
        float S2[4] = {2.0f * va, vb, vb, 2.0f * vc}; /* V + V^T */
        float TC[6];                                  /* T * cov3d, 2x3 */
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++)
                TC[3 * i + j] = t.T[3 * i + 0] * t.cov3d[0 + j] + t.T[3 * i + 1] * t.cov3d[3 + j] +
                                t.T[3 * i + 2] * t.cov3d[6 + j];
        float vT[6];
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++) vT[3 * i + j] = S2[2 * i + 0] * TC[j] + S2[2 * i + 1] * TC[3 + j];
        /* G = T^T (V+V^T) T  (3x3 symmetric) so that v_M = G M */
        float ST[6];
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++) ST[3 * i + j] = S2[2 * i + 0] * t.T[j] + S2[2 * i + 1] * t.T[3 + j];
        float G[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) G[3 * i + j] = t.T[i] * ST[j] + t.T[3 + i] * ST[3 + j];
        float vM[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                vM[3 * i + j] = G[3 * i + 0] * t.M[0 + j] + G[3 * i + 1] * t.M[3 + j] + G[3 * i + 2] * t.M[6 + j];
        /* M = Rq * g * s (:75) */
        float vR[9];
        for (int j = 0; j < 3; j++) {
            float acc = 0.0f;
            for (int i = 0; i < 3; i++) {
                acc += t.Rq[3 * i + j] * vM[3 * i + j];
                vR[3 * i + j] = vM[3 * i + j] * glob_scale * s[j];
            }
            v_scales[3 * n + j] = glob_scale * acc;
        }
        /* Rq <- normalised quat u = (w,x,y,z) */
        float nrm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        if (nrm < 1e-12f) nrm = 1e-12f;
        float w = q[0] / nrm, x = q[1] / nrm, y = q[2] / nrm, z = q[3] / nrm;
        float vu[4];
        vu[0] = 2.0f * (-z * vR[1] + y * vR[2] + z * vR[3] - x * vR[5] - y * vR[6] + x * vR[7]);
        vu[1] = 2.0f * (y * vR[1] + z * vR[2] + y * vR[3] - 2.0f * x * vR[4] - w * vR[5] + z * vR[6] +
                        w * vR[7] - 2.0f * x * vR[8]);
        vu[2] = 2.0f * (-2.0f * y * vR[0] + x * vR[1] + w * vR[2] + x * vR[3] + z * vR[5] - w * vR[6] +
                        z * vR[7] - 2.0f * y * vR[8]);
        vu[3] = 2.0f * (-2.0f * z * vR[0] - w * vR[1] + x * vR[2] + w * vR[3] - 2.0f * z * vR[4] +
                        y * vR[5] + x * vR[6] + y * vR[7]);
        float dotuv = w * vu[0] + x * vu[1] + y * vu[2] + z * vu[3];
        float u[4] = {w, x, y, z};
        for (int k = 0; k < 4; k++) v_quats[4 * n + k] = (vu[k] - u[k] * dotuv) / nrm;

        /* T = J Rclip (:94): v_J = v_T Rclip^T */
        float vJ[6];
        for (int i = 0; i < 2; i++)
            for (int k = 0; k < 3; k++)
                vJ[3 * i + k] = vT[3 * i + 0] * vm[4 * k + 0] + vT[3 * i + 1] * vm[4 * k + 1] +
                                vT[3 * i + 2] * vm[4 * k + 2];
        float vrz = fx * vJ[0] + fy * vJ[4] +
                    2.0f * t.rz * (-fx * t.t[0] * vJ[2] - fy * t.t[1] * vJ[5]);
        float vtx = -fx * t.rz2 * vJ[2];
        float vty = -fy * t.rz2 * vJ[5];
        float vp[3] = {0.0f, 0.0f, -t.rz2 * vrz};
        float limx = 1.3f * (0.5f * (float)W / fx), limy = 1.3f * (0.5f * (float)H / fy);
        if (t.clampx == 0) vp[0] += vtx; else vp[2] += (t.clampx > 0 ? limx : -limx) * vtx;
        if (t.clampy == 0) vp[1] += vty; else vp[2] += (t.clampy > 0 ? limy : -limy) * vty;
        if (v_depth_view) vp[2] += v_depth_view[n];
        for (int j = 0; j < 3; j++)
            vmean[j] += vm[0 + j] * vp[0] + vm[4 + j] * vp[1] + vm[8 + j] * vp[2];
        for (int j = 0; j < 3; j++) v_means[3 * n + j] = vmean[j];
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Spherical harmonics; constants and basis as gsplat_cpu.cpp:379-407, :424-486. */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                              -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                              0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                              -0.5900435899266435f};
static const float SH_C4[] = {2.5033429417967046f,  -1.7701307697799304f, 0.9461746957575601f,
                              -0.6690465435572892f, 0.10578554691520431f, -0.6690465435572892f,
                              0.47308734787878004f, -1.7701307697799304f, 0.6258357354491761f};

int orc_num_sh_bases(int degree) { /* :409-422 */
    switch (degree) {
    case 0: return 1;
    case 1: return 4;
    case 2: return 9;
    case 3: return 16;
    default: return 25;
    }
}

static void sh_basis(int num_bases, const float *d, float *r /* 25 */) {
    for (int i = 0; i < 25; i++) r[i] = 0.0f;
    r[0] = SH_C0;
    if (num_bases <= 1) return;
    float x = d[0], y = d[1], z = d[2];
    r[1] = SH_C1 * -y;
    r[2] = SH_C1 * z;
    r[3] = SH_C1 * -x;
    if (num_bases <= 4) return;
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    r[4] = SH_C2[0] * xy;
    r[5] = SH_C2[1] * yz;
    r[6] = SH_C2[2] * (2.0f * zz - xx - yy);
    r[7] = SH_C2[3] * xz;
    r[8] = SH_C2[4] * (xx - yy);
    if (num_bases <= 9) return;
    r[9] = SH_C3[0] * y * (3.0f * xx - yy);
    r[10] = SH_C3[1] * xy * z;
    r[11] = SH_C3[2] * y * (4.0f * zz - xx - yy);
    r[12] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
    r[13] = SH_C3[4] * x * (4.0f * zz - xx - yy);
    r[14] = SH_C3[5] * z * (xx - yy);
    r[15] = SH_C3[6] * x * (xx - 3.0f * yy);
    if (num_bases <= 16) return;
    r[16] = SH_C4[0] * xy * (xx - yy);
    r[17] = SH_C4[1] * yz * (3.0f * xx - yy);
    r[18] = SH_C4[2] * xy * (7.0f * zz - 1.0f);
    r[19] = SH_C4[3] * yz * (7.0f * zz - 3.0f);
    r[20] = SH_C4[4] * (zz * (35.0f * zz - 30.0f) + 3.0f);
    r[21] = SH_C4[5] * xz * (7.0f * zz - 3.0f);
    r[22] = SH_C4[6] * (xx - yy) * (7.0f * zz - 1.0f);
    r[23] = SH_C4[7] * xz * (xx - 3.0f * yy);
    r[24] = SH_C4[8] * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy));
}

/* colors[n,c] = sum_b basis_b(dir_n) * coeffs[n,b,c]; bases >= numShBases(degrees_to_use) are 0. */
int orc_sh_forward(int N, int K, int degrees_to_use, const float *dirs, const float *coeffs,
                   float *colors) {
    int nb = orc_num_sh_bases(degrees_to_use);
    if (nb > K) nb = K;
    for (int n = 0; n < N; n++) {
        float r[25];
        sh_basis(nb, dirs + 3 * n, r);
        for (int c = 0; c < 3; c++) {
            float acc = 0.0f;
            for (int b = 0; b < K; b++) acc += r[b] * coeffs[(size_t)n * K * 3 + 3 * b + c];
            colors[3 * n + c] = acc;
        }
    }
    return 0;
}

/* v_coeffs[n,b,c] = basis_b * v_colors[n,c] (autograd of the product-sum at :485). */
int orc_sh_backward(int N, int K, int degrees_to_use, const float *dirs, const float *v_colors,
                    float *v_coeffs) {
    int nb = orc_num_sh_bases(degrees_to_use);
    if (nb > K) nb = K;
    for (int n = 0; n < N; n++) {
        float r[25];
        sh_basis(nb, dirs + 3 * n, r);
        for (int b = 0; b < K; b++)
            for (int c = 0; c < 3; c++)
                v_coeffs[(size_t)n * K * 3 + 3 * b + c] = r[b] * v_colors[3 * n + c];
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Compositing; gsplat_cpu.cpp:137-257 (forward) and :260-376 (backward). */
typedef struct {
    int64_t pixels;
    int64_t total;
    int64_t *offsets; /* pixels + 1 */
    int32_t *ids;     /* per pixel, back-to-front (after the reverse at :252) */
} OrcRaster;

static const float *g_sort_key;
static int cmp_depth(const void *a, const void *b) {
    int32_t ia = *(const int32_t *)a, ib = *(const int32_t *)b;
    float da = g_sort_key[ia], db = g_sort_key[ib];
    if (da < db) return -1;
    if (da > db) return 1;
    return (ia > ib) - (ia < ib); /* the reference's std::sort (:155-159) is unstable; ties are
                                     broken by index here — fixtures avoid depth ties */
}

/* Pixel rectangle of one Gaussian; gsplat_cpu.cpp:167-168,201-204.  rows [r0,r1), cols [c0,c1). */
void orc_pixel_rect(float gX, float gY, float cov_xx, float cov_yy, int W, int H, int *r0, int *r1,
                    int *c0, int *c1) {
    float sqx = 3.0f * sqrtf(cov_xx), sqy = 3.0f * sqrtf(cov_yy);
    int minx = (int)floorf(gY - sqy) - 2;
    int maxx = (int)ceilf(gY + sqy) + 2;
    int miny = (int)floorf(gX - sqx) - 2;
    int maxy = (int)ceilf(gX + sqx) + 2;
    *r0 = minx > 0 ? minx : 0;
    *r1 = maxx < H ? maxx : H;
    *c0 = miny > 0 ? miny : 0;
    *c1 = maxy < W ? maxy : W;
}

/* Window variant: only the pixels of rows [wy0,wy1) x cols [wx0,wx1) are evaluated (every pixel's
 * recurrence is independent of every other pixel's, so a window is the full image restricted);
 * pixels outside keep T = 1, colour = background, no contributors.  Lets the parity tests check
 * crops of BASELINE-size frames in full-image coordinates (no translation round-off). */
static void *rasterize_forward_impl(int W, int H, int N, const float *xys, const float *conics,
                                    const float *colors, const float *opacities,
                                    const float *background, const float *cov2d /* N x 2 x 2 */,
                                    const float *cam_depths, float *out_img, float *final_Ts,
                                    int32_t *px_counts, int wx0, int wy0, int wx1, int wy1,
                                    const int32_t *tile_rect /* N x 4 or NULL */) {
    int64_t P = (int64_t)W * H;
    if (wx0 < 0) wx0 = 0;
    if (wy0 < 0) wy0 = 0;
    if (wx1 > W) wx1 = W;
    if (wy1 > H) wy1 = H;
    int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    for (int i = 0; i < N; i++) order[i] = i;
    g_sort_key = cam_depths;
    qsort(order, (size_t)N, sizeof(int32_t), cmp_depth);

    unsigned char *done = (unsigned char *)calloc((size_t)P, 1);
    int32_t *counts = (int32_t *)calloc((size_t)P, sizeof(int32_t));
    for (int64_t i = 0; i < P; i++) final_Ts[i] = 1.0f;
    memset(out_img, 0, sizeof(float) * 3 * (size_t)P);

    /* append log of (pixel, gaussian) in processing order */
    size_t cap = 1 << 20, len = 0;
    int32_t *log_pix = (int32_t *)malloc(cap * sizeof(int32_t));
    int32_t *log_gid = (int32_t *)malloc(cap * sizeof(int32_t));

    const float alphaThresh = 1.0f / 255.0f;
    for (int idx = 0; idx < N; idx++) {
        int32_t g = order[idx];
        float A = conics[3 * g + 0], B = conics[3 * g + 1], C = conics[3 * g + 2];
        float gX = xys[2 * g + 0], gY = xys[2 * g + 1];
        int r0, r1, c0, c1;
        orc_pixel_rect(gX, gY, cov2d[4 * g + 0], cov2d[4 * g + 3], W, H, &r0, &r1, &c0, &c1);
        if (r0 < wy0) r0 = wy0;
        if (r1 > wy1) r1 = wy1;
        if (c0 < wx0) c0 = wx0;
        if (c1 > wx1) c1 = wx1;
        if (tile_rect) {
            /* the caller's binning contract: the Gaussian is in the lists of tiles [tx0,tx1) x [ty0,ty1)
             * only, and a pixel only walks the list of its own 16 x 16 tile (forward.cu:256-283) */
            const int32_t *t = tile_rect + 4 * (size_t)g;
            if (c0 < 16 * t[0]) c0 = 16 * t[0];
            if (c1 > 16 * t[1]) c1 = 16 * t[1];
            if (r0 < 16 * t[2]) r0 = 16 * t[2];
            if (r1 > 16 * t[3]) r1 = 16 * t[3];
        }
        for (int i = r0; i < r1; i++) {
            for (int j = c0; j < c1; j++) {
                int64_t pix = (int64_t)i * W + j;
                if (done[pix]) continue;
                float xCam = gX - (float)j;
                float yCam = gY - (float)i;
                float sigma = 0.5f * (A * xCam * xCam + C * yCam * yCam) + B * xCam * yCam;
                if (sigma < 0.0f) continue;
                float alpha = fminf(0.999f, opacities[g] * expf(-sigma));
                if (alpha < alphaThresh) continue;
                float T = final_Ts[pix];
                float nextT = T * (1.0f - alpha);
                if (nextT <= 1e-4f) {
                    done[pix] = 1;
                    continue;
                }
                float vis = alpha * T;
                out_img[3 * pix + 0] += vis * colors[3 * g + 0];
                out_img[3 * pix + 1] += vis * colors[3 * g + 1];
                out_img[3 * pix + 2] += vis * colors[3 * g + 2];
                final_Ts[pix] = nextT;
                if (len == cap) {
                    cap *= 2;
                    log_pix = (int32_t *)realloc(log_pix, cap * sizeof(int32_t));
                    log_gid = (int32_t *)realloc(log_gid, cap * sizeof(int32_t));
                }
                log_pix[len] = (int32_t)pix;
                log_gid[len] = g;
                len++;
                counts[pix]++;
            }
        }
    }
    for (int64_t pix = 0; pix < P; pix++) {
        float T = final_Ts[pix];
        out_img[3 * pix + 0] += T * background[0];
        out_img[3 * pix + 1] += T * background[1];
        out_img[3 * pix + 2] += T * background[2];
    }

    OrcRaster *st = (OrcRaster *)malloc(sizeof(OrcRaster));
    st->pixels = P;
    st->total = (int64_t)len;
    st->offsets = (int64_t *)malloc(sizeof(int64_t) * (size_t)(P + 1));
    st->ids = (int32_t *)malloc(sizeof(int32_t) * (len > 0 ? len : 1));
    st->offsets[0] = 0;
    for (int64_t pix = 0; pix < P; pix++) st->offsets[pix + 1] = st->offsets[pix] + counts[pix];
    /* fill back-to-front: walk the log forwards, write each pixel's slots from its end */
    int64_t *cursor = (int64_t *)malloc(sizeof(int64_t) * (size_t)P);
    for (int64_t pix = 0; pix < P; pix++) cursor[pix] = st->offsets[pix + 1];
    for (size_t k = 0; k < len; k++) st->ids[--cursor[log_pix[k]]] = log_gid[k];
    if (px_counts) memcpy(px_counts, counts, sizeof(int32_t) * (size_t)P);
    free(cursor);
    free(log_pix);
    free(log_gid);
    free(counts);
    free(done);
    free(order);
    return st;
}

void *orc_rasterize_forward_window(int W, int H, int N, const float *xys, const float *conics,
                                   const float *colors, const float *opacities,
                                   const float *background, const float *cov2d /* N x 2 x 2 */,
                                   const float *cam_depths, float *out_img, float *final_Ts,
                                   int32_t *px_counts, int wx0, int wy0, int wx1, int wy1) {
    return rasterize_forward_impl(W, H, N, xys, conics, colors, opacities, background, cov2d, cam_depths,
                                  out_img, final_Ts, px_counts, wx0, wy0, wx1, wy1, NULL);
}

/* The same recurrence under a CALLER-SUPPLIED binning contract: tile_rect[g] = {tx0, tx1, ty0, ty1} are the
 * 16 x 16 tiles whose lists hold Gaussian g (rasterizer/gsplat/forward.cu:86-94 assigns the tiles under the
 * square of half-width `radius` around the centre, helpers.cuh:17-49; map_gaussian_to_intersects :107-143
 * writes one list entry per such tile), and a pixel walks the list of its own tile only (:256-283).  Per
 * pixel the decisions stay gsplat-cpu's (rectangle of gsplat_cpu.cpp:167-168,201-204, sigma, alpha, T): what
 * the launcher-level functions of this repo compute when they are handed the reference's tile lists. */
void *orc_rasterize_forward_tiles(int W, int H, int N, const float *xys, const float *conics,
                                  const float *colors, const float *opacities, const float *background,
                                  const float *cov2d, const float *cam_depths, const int32_t *tile_rect,
                                  float *out_img, float *final_Ts, int32_t *px_counts) {
    return rasterize_forward_impl(W, H, N, xys, conics, colors, opacities, background, cov2d, cam_depths,
                                  out_img, final_Ts, px_counts, 0, 0, W, H, tile_rect);
}

void *orc_rasterize_forward(int W, int H, int N, const float *xys, const float *conics,
                            const float *colors, const float *opacities, const float *background,
                            const float *cov2d /* N x 2 x 2 */, const float *cam_depths,
                            float *out_img, float *final_Ts, int32_t *px_counts) {
    return orc_rasterize_forward_window(W, H, N, xys, conics, colors, opacities, background, cov2d,
                                        cam_depths, out_img, final_Ts, px_counts, 0, 0, W, H);
}

int64_t orc_rasterize_total(void *state) { return ((OrcRaster *)state)->total; }

int orc_rasterize_contributors(void *state, int32_t *ids) {
    OrcRaster *st = (OrcRaster *)state;
    memcpy(ids, st->ids, sizeof(int32_t) * (size_t)st->total);
    return 0;
}

int orc_rasterize_free(void *state) {
    OrcRaster *st = (OrcRaster *)state;
    if (st) {
        free(st->offsets);
        free(st->ids);
        free(st);
    }
    return 0;
}

/* gsplat_cpu.cpp:260-376.  v_out_alpha may be NULL (zeros, rasterize_gaussians.cpp:198).
 * Outputs are zero-filled here, then accumulated in pixel raster order. */
int orc_rasterize_backward_window(int W, int H, int N, const float *xys, const float *conics,
                                  const float *colors, const float *opacities,
                                  const float *background, const float *final_Ts, void *state,
                                  const float *v_out, const float *v_out_alpha, float *v_xy,
                                  float *v_conic, float *v_colors, float *v_opacity, int wx0,
                                  int wy0, int wx1, int wy1) {
    if (wx0 < 0) wx0 = 0;
    if (wy0 < 0) wy0 = 0;
    if (wx1 > W) wx1 = W;
    if (wy1 > H) wy1 = H;
    OrcRaster *st = (OrcRaster *)state;
    memset(v_xy, 0, sizeof(float) * 2 * (size_t)N);
    memset(v_conic, 0, sizeof(float) * 3 * (size_t)N);
    memset(v_colors, 0, sizeof(float) * 3 * (size_t)N);
    memset(v_opacity, 0, sizeof(float) * (size_t)N);
    const float bgX = background[0], bgY = background[1], bgZ = background[2];
    const float alphaThresh = 1.0f / 255.0f;
    for (int i = wy0; i < wy1; i++) {
        for (int j = wx0; j < wx1; j++) {
            int64_t pix = (int64_t)i * W + j;
            float Tfinal = final_Ts[pix];
            float T = Tfinal;
            float buffer[3] = {0.0f, 0.0f, 0.0f};
            float vo0 = v_out[3 * pix + 0], vo1 = v_out[3 * pix + 1], vo2 = v_out[3 * pix + 2];
            float voa = v_out_alpha ? v_out_alpha[pix] : 0.0f;
            for (int64_t k = st->offsets[pix]; k < st->offsets[pix + 1]; k++) {
                int32_t g = st->ids[k];
                float A = conics[3 * g + 0], B = conics[3 * g + 1], C = conics[3 * g + 2];
                float gX = xys[2 * g + 0], gY = xys[2 * g + 1];
                float xCam = gX - (float)j;
                float yCam = gY - (float)i;
                float sigma = 0.5f * (A * xCam * xCam + C * yCam * yCam) + B * xCam * yCam;
                if (sigma < 0.0f) continue;
                float vis = expf(-sigma);
                float alpha = fminf(0.99f, opacities[g] * vis);
                if (alpha < alphaThresh) continue;
                float ra = 1.0f / (1.0f - alpha);
                T *= ra;
                float fac = alpha * T;
                v_colors[3 * g + 0] += fac * vo0;
                v_colors[3 * g + 1] += fac * vo1;
                v_colors[3 * g + 2] += fac * vo2;
                float v_alpha = ((colors[3 * g + 0] * T - buffer[0] * ra) * vo0) +
                                ((colors[3 * g + 1] * T - buffer[1] * ra) * vo1) +
                                ((colors[3 * g + 2] * T - buffer[2] * ra) * vo2) +
                                (Tfinal * ra * voa) + (-Tfinal * ra * bgX * vo0) +
                                (-Tfinal * ra * bgY * vo1) + (-Tfinal * ra * bgZ * vo2);
                buffer[0] += colors[3 * g + 0] * fac;
                buffer[1] += colors[3 * g + 1] * fac;
                buffer[2] += colors[3 * g + 2] * fac;
                float v_sigma = -opacities[g] * vis * v_alpha;
                v_conic[3 * g + 0] += 0.5f * v_sigma * xCam * xCam;
                v_conic[3 * g + 1] += 0.5f * v_sigma * xCam * yCam;
                v_conic[3 * g + 2] += 0.5f * v_sigma * yCam * yCam;
                v_xy[2 * g + 0] += v_sigma * (A * xCam + B * yCam);
                v_xy[2 * g + 1] += v_sigma * (B * xCam + C * yCam);
                v_opacity[g] += vis * v_alpha;
            }
        }
    }
    return 0;
}

int orc_rasterize_backward(int W, int H, int N, const float *xys, const float *conics,
                           const float *colors, const float *opacities, const float *background,
                           const float *final_Ts, void *state, const float *v_out,
                           const float *v_out_alpha, float *v_xy, float *v_conic, float *v_colors,
                           float *v_opacity) {
    return orc_rasterize_backward_window(W, H, N, xys, conics, colors, opacities, background,
                                         final_Ts, state, v_out, v_out_alpha, v_xy, v_conic,
                                         v_colors, v_opacity, 0, 0, W, H);
}

/* libm expf, exported so tests can check the device expf against the host's bit for bit. */
float orc_expf(float x) { return expf(x); }
void orc_expf_array(int64_t n, const float *x, float *y) {
    for (int64_t i = 0; i < n; i++) y[i] = expf(x[i]);
}
