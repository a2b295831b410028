// gs_fused.hip — the per-Gaussian stages of the path as ONE kernel per direction
// (gs_gaussian_forward / gs_gaussian_backward, include/gsplat_hip.h "Fused per-Gaussian stages").
//
// Forward  = gs_project_forward + gs_sh_forward_fused + gs_pack_splats:
//   reads  means 12 + scales 12 + quats 16 + opacity 4 + SH 12K               (232 B at K = 16)
//   writes packed 48 + depth 4 + radius 4 + raw rgb 12 (+ xys 8)               ( 68 B)
//   instead of 84 + (12K + 12) + 24 read and 84 + 24 + 52 written by the three stage kernels: the
//   2-D intermediates (xys, conics, cov2d, cov3d, colours, tile counts) stay in registers.
// Backward = record split + gs_sh_backward_fused + gs_project_backward:
//   reads  the 64-byte gradient record, parameters 44, radius 4, raw rgb 12    (124 B)
//   writes v_SH 12K + v_means 12 + v_scales 12 + v_quats 16 + v_opacity 4 (and, on request,
//   zeroes the record so that the next frame can skip its memset — slower at C2, see the header).
// One lane per Gaussian; the higher-band SH rows of a wave's 64 Gaussians move through an LDS slab
// of odd row stride with coalesced 16-byte global accesses (as in gs_sh.hip).  Both kernels are
// HBM-streaming; at K = 16 the forward is one launch of two kinds of LDS-free workgroups instead
// (k_sh_project_pack16).
// Device functions are shared with the stage kernels (gs_gaussian.h): identical results.
#include "gs_gaussian.h"

namespace gs {

#ifndef GS_GRAD_REC
#define GS_GRAD_REC 16
#endif
constexpr int kRec = GS_GRAD_REC;  // floats per gradient record (gs_raster.hip: kGradRec)

template <int K>
__global__ void __launch_bounds__(ShSplit<K>::kBlock)
k_gaussian_forward(CamArgs cam, const float *__restrict__ vm_dev, const float *__restrict__ pm_dev,
                   int N, int nb, const float *__restrict__ means, const float *__restrict__ scales,
                   const float *__restrict__ quats, const float *__restrict__ opacities,
                   const float *__restrict__ dc, const float *__restrict__ rest, float cx, float cy,
                   float cz, const float *__restrict__ cp_dev, float4 *__restrict__ packed,
                   float *__restrict__ depths, int32_t *__restrict__ radii,
                   float *__restrict__ rgb_raw, float *__restrict__ xys, uint32_t flags) {
    constexpr int ROW = ShSplit<K>::ROW, ROWP = ShSplit<K>::ROWP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (cp_dev) { cx = cp_dev[0]; cy = cp_dev[1]; cz = cp_dev[2]; }
    load_device_matrices(cam, vm_dev, pm_dev);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *slab = smem + wave * (64 * ROWP);
    const int64_t g0 = ((int64_t)blockIdx.x * (ShSplit<K>::kBlock / 64) + wave) * 64;
    const int cnt = g0 < N ? min(64, (int)(N - g0)) : 0;
    // The wave's slab of higher-band coefficients is fetched into registers FIRST (coalesced 16-byte
    // loads, fixed trip count), the projection arithmetic of this lane's Gaussian runs while those
    // loads are in flight, and only then do the values go through LDS to their rows.
    constexpr int kSlabIters = (64 * ROW + 255) / 256;
    float4 buf[kSlabIters > 0 ? kSlabIters : 1];
    const int total = cnt * ROW;
    if constexpr (ROW > 0) {
        const float *src = rest + g0 * ROW;  // 16-B aligned: 64 * ROW * 4 bytes per wave
#pragma unroll
        for (int it = 0; it < kSlabIters; it++) {
            const int i = (it * 64 + lane) * 4;
            buf[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i + 3 < total) {
                const float4_u v = *reinterpret_cast<const float4_u *>(src + i);
                buf[it] = make_float4(v.x, v.y, v.z, v.w);
            } else if (i < total) {   // ragged end of the last wave
                float e[4] = {0.f, 0.f, 0.f, 0.f};
                for (int k = 0; k < 4 && i + k < total; k++) e[k] = src[i + k];
                buf[it] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    }
    const int64_t g = g0 + min(lane, max(cnt - 1, 0));  // idle lanes shadow the last Gaussian
    const bool active = lane < cnt;

    // ---- projection (gs_project_forward) ---------------------------------------------------------
    Proj o;
    ProjOut po;
    if (cnt > 0) {
        float mean[3] = {means[3 * g], means[3 * g + 1], means[3 * g + 2]};
        float scale[3] = {scales[3 * g], scales[3 * g + 1], scales[3 * g + 2]};
        if (cam.flags & GS_CAM_LOG_SCALES) {
#pragma unroll
            for (int j = 0; j < 3; j++) scale[j] = expf(scale[j]);
        }
        const float4_u q4 = reinterpret_cast<const float4_u *>(quats)[g];
        float quat[4] = {q4.x, q4.y, q4.z, q4.w};
        project_one(cam, mean, scale, quat, o);
        project_outputs(cam, o, po);
    }
    if constexpr (ROW > 0) {
#pragma unroll
        for (int it = 0; it < kSlabIters; it++) {
            const int i = (it * 64 + lane) * 4;
            const float e[4] = {buf[it].x, buf[it].y, buf[it].z, buf[it].w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int idx = i + k;
                if (idx < total) slab[(idx / ROW) * ROWP + (idx % ROW)] = e[k];
            }
        }
    }
    __syncthreads();
    if (!active) return;

    // ---- SH colour (gs_sh_forward_fused) ---------------------------------------------------------
    float x, y, z;
    view_dir(means, g, cx, cy, cz, x, y, z);
    float r[25];
    sh_basis(nb, x, y, z, r);
    float c0 = r[0] * dc[3 * g], c1 = r[0] * dc[3 * g + 1], c2 = r[0] * dc[3 * g + 2];
    const float *row = slab + lane * ROWP;
#pragma unroll
    for (int b = 1; b < K; b++) {
        c0 += r[b] * row[3 * (b - 1) + 0];
        c1 += r[b] * row[3 * (b - 1) + 1];
        c2 += r[b] * row[3 * (b - 1) + 2];
    }
    rgb_raw[3 * g + 0] = c0;
    rgb_raw[3 * g + 1] = c1;
    rgb_raw[3 * g + 2] = c2;

    depths[g] = o.p[2];
    radii[g] = po.radius;
    if (xys) {
        xys[2 * g + 0] = po.u;
        xys[2 * g + 1] = po.v;
    }

    // ---- packed compositing record (gs_pack_splats) -------------------------------------------------
    float4 p0, p1, p2;
    pack_one(cam.W, cam.H, po.u, po.v, po.conic[0], po.conic[1], po.conic[2], true, o.a, o.c,
             opacities[g], po.radius, fmaxf(c0 + 0.5f, 0.0f), fmaxf(c1 + 0.5f, 0.0f),
             fmaxf(c2 + 0.5f, 0.0f), flags, p0, p1, p2);
    packed[3 * g + 0] = p0;
    packed[3 * g + 1] = p1;
    packed[3 * g + 2] = p2;
}

// rgb_raw == nullptr: the colour part of the record (p2.xyz) is somebody else's to write — the SH
// lanes of k_sh_project_pack16 — and only p2.w (the rectangle's rows) is stored
__device__ __forceinline__ void project_pack_body(int g, CamArgs &cam, const float *__restrict__ vm_dev,
                                                  const float *__restrict__ pm_dev, int N,
                                                  const float *__restrict__ means,
                                                  const float *__restrict__ scales,
                                                  const float *__restrict__ quats,
                                                  const float *__restrict__ opacities,
                                                  const float *__restrict__ rgb_raw,
                                                  float4 *__restrict__ packed, float *__restrict__ depths,
                                                  int32_t *__restrict__ radii, float *__restrict__ xys,
                                                  uint32_t flags) {
    if (g >= N) return;
    // every input of the lane is requested before the first use: the projection's few hundred
    // instructions then run on top of ONE memory round trip (the first version fetched the quaternion
    // after the scales had arrived and the colour / opacity after the projection: three in a row)
    float mean[3] = {means[3 * g], means[3 * g + 1], means[3 * g + 2]};
    float scale[3] = {scales[3 * g], scales[3 * g + 1], scales[3 * g + 2]};
    const float4_u q4 = reinterpret_cast<const float4_u *>(quats)[g];
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    if (rgb_raw) { c0 = rgb_raw[3 * g + 0]; c1 = rgb_raw[3 * g + 1]; c2 = rgb_raw[3 * g + 2]; }
    const float opac = opacities[g];
    load_device_matrices(cam, vm_dev, pm_dev);
    if (cam.flags & GS_CAM_LOG_SCALES) {
#pragma unroll
        for (int j = 0; j < 3; j++) scale[j] = expf(scale[j]);
    }
    float quat[4] = {q4.x, q4.y, q4.z, q4.w};
    Proj o;
    project_one(cam, mean, scale, quat, o);
    ProjOut po;
    project_outputs(cam, o, po);
    depths[g] = o.p[2];
    radii[g] = po.radius;
    if (xys) {
        xys[2 * g + 0] = po.u;
        xys[2 * g + 1] = po.v;
    }
    float4 p0, p1, p2;
    pack_one(cam.W, cam.H, po.u, po.v, po.conic[0], po.conic[1], po.conic[2], true, o.a, o.c,
             opac, po.radius, fmaxf(c0 + 0.5f, 0.0f), fmaxf(c1 + 0.5f, 0.0f),
             fmaxf(c2 + 0.5f, 0.0f), flags, p0, p1, p2);
    packed[3 * (size_t)g + 0] = p0;
    packed[3 * (size_t)g + 1] = p1;
    if (rgb_raw) packed[3 * (size_t)g + 2] = p2;
    else reinterpret_cast<float *>(packed)[12 * (size_t)g + 11] = p2.w;
}

// K = 16: the SH rows are 180 bytes; one lane per Gaussian needs them in an 11.5 KB-per-wave LDS slab,
// which caps the occupancy at three waves per SIMD and leaves the kernel above waiting on memory
// (measured 103 us at N = 1 M).  Faster: the four-lanes-per-Gaussian SH kernel of gs_sh.hip (no
// LDS) writes the raw rgb, and this LDS-free kernel does projection + packed record.
__global__ void __launch_bounds__(256)
k_project_pack(CamArgs cam, const float *__restrict__ vm_dev, const float *__restrict__ pm_dev, int N,
               const float *__restrict__ means, const float *__restrict__ scales,
               const float *__restrict__ quats, const float *__restrict__ opacities,
               const float *__restrict__ rgb_raw, float4 *__restrict__ packed,
               float *__restrict__ depths, int32_t *__restrict__ radii, float *__restrict__ xys,
               uint32_t flags) {
    project_pack_body(blockIdx.x * blockDim.x + threadIdx.x, cam, vm_dev, pm_dev, N, means, scales, quats,
                      opacities, rgb_raw, packed, depths, radii, xys, flags);
}

// Both of them in ONE launch (gs_gaussian_forward at K = 16): of every five consecutive workgroups
// four run the SH forward (256 threads = 64 Gaussians each, HBM-streaming) and the fifth the
// projection + packed record of the same 256 Gaussians (VALU-bound), so that the two kinds sit on the
// CUs side by side instead of one kernel after the other.  Nothing depends on anything: the SH
// lanes put clamp_min(rgb + 0.5, 0) into bytes 32..43 of the packed record themselves, the
// projection lanes write the other 36 bytes.
__global__ void __launch_bounds__(256)
k_sh_project_pack16(CamArgs cam, const float *__restrict__ vm_dev, const float *__restrict__ pm_dev, int N,
                    int nb, const float *__restrict__ means, const float *__restrict__ scales,
                    const float *__restrict__ quats, const float *__restrict__ opacities,
                    const float *__restrict__ dc, const float *__restrict__ rest, float cx, float cy,
                    float cz, const float *__restrict__ cp_dev, float4 *__restrict__ packed,
                    float *__restrict__ depths, int32_t *__restrict__ radii, float *__restrict__ rgb_raw,
                    float *__restrict__ xys, uint32_t flags) {
    const int grp = blockIdx.x / 5, r = blockIdx.x - 5 * grp;
    if (r < 4) {
        sh_forward16_quad_body((int64_t)(4 * grp + r) * 256 + threadIdx.x, N, nb, means, cx, cy, cz, cp_dev, dc, rest,
                               reinterpret_cast<float *>(packed) + 8, 12, rgb_raw);
    } else {
        project_pack_body(grp * 256 + (int)threadIdx.x, cam, vm_dev, pm_dev, N, means, scales, quats, opacities,
                          nullptr, packed, depths, radii, xys, flags);
    }
}

template <int K>
__global__ void __launch_bounds__(ShSplit<K>::kBlock)
k_gaussian_backward(CamArgs cam, const float *__restrict__ vm_dev, const float *__restrict__ pm_dev,
                    int N, int nb, const float *__restrict__ means, const float *__restrict__ scales,
                    const float *__restrict__ quats, const float *__restrict__ opacities, float cx,
                    float cy, float cz, const float *__restrict__ cp_dev,
                    const int32_t *__restrict__ radii, const float *__restrict__ rgb_raw,
                    float4 *__restrict__ records, float *__restrict__ v_means,
                    float *__restrict__ v_scales, float *__restrict__ v_quats,
                    float *__restrict__ v_opacity, float *__restrict__ v_dc,
                    float *__restrict__ v_rest, float *__restrict__ v_xy, uint32_t flags) {
    constexpr int ROW = ShSplit<K>::ROW, ROWP = ShSplit<K>::ROWP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (cp_dev) { cx = cp_dev[0]; cy = cp_dev[1]; cz = cp_dev[2]; }
    load_device_matrices(cam, vm_dev, pm_dev);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *slab = smem + wave * (64 * ROWP);
    // GS_FLAG_ACCUMULATE_GRADS: the six parameter gradients are ADDED to what the output tensors
    // hold (several cameras per optimiser step on one rank, one all-reduce for all of them)
    const bool accum = (flags & GS_FLAG_ACCUMULATE_GRADS) != 0u;
    auto put = [accum](float *p, float v) { *p = accum ? *p + v : v; };
    const int64_t g0 = ((int64_t)blockIdx.x * (ShSplit<K>::kBlock / 64) + wave) * 64;
    const int cnt = g0 < N ? min(64, (int)(N - g0)) : 0;
    if (lane < cnt) {
        const int64_t g = g0 + lane;
        // ---- every input of the lane is requested before the first use (one memory round trip under
        //      the arithmetic instead of a chain of seven: record, opacity, mean, rgb, radius, mean /
        //      scale, quaternion) ----
        // the gradient record of gs_rasterize_backward: {vx vy vA vB | vC vr vg vb | vo - - -}
        const float4 ra = records[(kRec / 4) * g + 0], rb = records[(kRec / 4) * g + 1];
        float vo = reinterpret_cast<const float *>(records)[kRec * (size_t)g + 8];
        float mean[3] = {means[3 * g], means[3 * g + 1], means[3 * g + 2]};
        const float raw0 = rgb_raw[3 * g + 0], raw1 = rgb_raw[3 * g + 1], raw2 = rgb_raw[3 * g + 2];
        const int32_t radius = radii[g];
        const float logit = opacities[g];
        float scale[3] = {scales[3 * g], scales[3 * g + 1], scales[3 * g + 2]};
        const float4_u q4 = reinterpret_cast<const float4_u *>(quats)[g];
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        if (flags & GS_FLAG_RECORDS_ZEROED) {  // keep the "records are zero between frames" invariant
            records[(kRec / 4) * g + 0] = zero;
            records[(kRec / 4) * g + 1] = zero;
            records[(kRec / 4) * g + 2] = zero;
        }
        if (flags & GS_FLAG_LOGIT_OPACITY) {  // d sigmoid: s (1 - s), model.cpp:215
            const float sg = 1.0f / (1.0f + expf(-logit));
            vo *= sg * (1.0f - sg);
        }
        put(v_opacity + g, vo);
        if (v_xy) {
            v_xy[2 * g + 0] = ra.x;
            v_xy[2 * g + 1] = ra.y;
        }

        // ---- SH backward (gs_sh_backward_fused) ------------------------------------------------------
        // view_dir(): (mean - T) / ||mean - T||, model.cpp:176-177, on the values loaded above
        float x = mean[0] - cx, y = mean[1] - cy, z = mean[2] - cz;
        const float nrm = sqrtf(x * x + y * y + z * z);
        x /= nrm; y /= nrm; z /= nrm;
        float r[25];
        sh_basis(nb, x, y, z, r);
        const float v0 = (raw0 + 0.5f >= 0.0f) ? rb.y : 0.0f;
        const float v1 = (raw1 + 0.5f >= 0.0f) ? rb.z : 0.0f;
        const float v2 = (raw2 + 0.5f >= 0.0f) ? rb.w : 0.0f;
        if (flags & GS_FLAG_EMIT_VCOLOR) {
            // factored gradient exchange: hand out the colour cotangent behind the clamp mask itself
            // (never accumulated: every camera has its own view direction); the SH gradients are
            // formed from all cameras' cotangents by gs_sh_backward_cameras
            v_dc[3 * g + 0] = v0;
            v_dc[3 * g + 1] = v1;
            v_dc[3 * g + 2] = v2;
        } else {
            put(v_dc + 3 * g + 0, r[0] * v0);
            put(v_dc + 3 * g + 1, r[0] * v1);
            put(v_dc + 3 * g + 2, r[0] * v2);
        }
        float *row = slab + lane * ROWP;
#pragma unroll
        for (int b = 1; b < K; b++) {
            row[3 * (b - 1) + 0] = r[b] * v0;
            row[3 * (b - 1) + 1] = r[b] * v1;
            row[3 * (b - 1) + 2] = r[b] * v2;
        }

        // ---- projection backward (gs_project_backward) -----------------------------------------------
        float4_u *vq4 = reinterpret_cast<float4_u *>(v_quats);
        if (radius <= 0) {  // culled Gaussians get no gradient (backward.cu:380-382)
            if (!accum) {
                v_means[3 * g] = v_means[3 * g + 1] = v_means[3 * g + 2] = 0.0f;
                v_scales[3 * g] = v_scales[3 * g + 1] = v_scales[3 * g + 2] = 0.0f;
                vq4[g] = (float4_u)(0.0f);
            }
        } else {
            if (cam.flags & GS_CAM_LOG_SCALES) {
#pragma unroll
                for (int j = 0; j < 3; j++) scale[j] = expf(scale[j]);
            }
            float quat[4] = {q4.x, q4.y, q4.z, q4.w};
            Proj o;
            project_one(cam, mean, scale, quat, o);
            ProjGrad pg;
            project_backward_one(cam, o, scale, ra.x, ra.y, ra.z, ra.w, rb.x, 0.0f, pg);
#pragma unroll
            for (int j = 0; j < 3; j++) {
                put(v_means + 3 * g + j, pg.v_mean[j]);
                put(v_scales + 3 * g + j, pg.v_scale[j]);
            }
            float4_u vq;
            vq.x = pg.v_quat[0]; vq.y = pg.v_quat[1]; vq.z = pg.v_quat[2]; vq.w = pg.v_quat[3];
            if (accum) vq = vq + vq4[g];
            vq4[g] = vq;
        }
    }
    __syncthreads();
    if constexpr (ROW > 0) {
        float *dst = v_rest + g0 * ROW;
        const int total = cnt * ROW;
        for (int i = lane * 4; i < total; i += 64 * 4) {
            if (i + 3 < total) {
                float e[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int idx = i + k;
                    e[k] = slab[(idx / ROW) * ROWP + (idx % ROW)];
                }
                float4_u v;
                v.x = e[0]; v.y = e[1]; v.z = e[2]; v.w = e[3];
                if (accum) {
                    v = v + *reinterpret_cast<float4_u *>(dst + i);
                    *reinterpret_cast<float4_u *>(dst + i) = v;
                } else {
                    // 180 B per Gaussian that nothing on the path reads back: streamed past the
                    // caches, so that their write-back does not sit on the NEXT kernel's reads (the
                    // SH forward of the following step: 53 -> 40 us in the timed loop; this kernel
                    // +5 us; the training iteration, whose Adam step reads them at once, unchanged)
                    // (four scalar builtins: the back end merges them into one global_store_dwordx4 nt;
                    // on the 4-byte-aligned vector type the hint is dropped)
                    __builtin_nontemporal_store(v.x, dst + i);
                    __builtin_nontemporal_store(v.y, dst + i + 1);
                    __builtin_nontemporal_store(v.z, dst + i + 2);
                    __builtin_nontemporal_store(v.w, dst + i + 3);
                }
            } else {
                for (int idx = i; idx < total; idx++)
                    put(dst + idx, slab[(idx / ROW) * ROWP + (idx % ROW)]);
            }
        }
    }
}

// SH gradients from the colour cotangents of SEVERAL cameras (the factored gradient exchange of the
// camera-per-rank path): v_sh[n][b] = sum over cameras c of basis_b(dir(n, c)) * v_c(n, c).
// The SH gradient of one camera is the outer product of a basis vector every rank can evaluate
// (it depends on the Gaussian's mean and the camera centre only) and three floats per Gaussian, so
// the ranks exchange the three floats (all-gather, 12 B per Gaussian and camera) instead of
// all-reducing 12 K bytes per Gaussian.  One lane per Gaussian, the sums in registers, cameras in
// index order (deterministic, identical on every rank); rows out through the LDS slab like
// k_gaussian_backward.  HBM-streaming: reads 12 + 12 * cameras, writes 12 K bytes per Gaussian.
template <int K>
__global__ void __launch_bounds__(ShSplit<K>::kBlock)
k_sh_backward_cameras(int N, int nb, int n_cams, const float *__restrict__ means,
                      const float *__restrict__ cam_pos, size_t cam_stride,
                      const float *__restrict__ v_colors, size_t v_stride, float *__restrict__ v_dc,
                      float *__restrict__ v_rest, uint32_t flags) {
    constexpr int ROW = ShSplit<K>::ROW, ROWP = ShSplit<K>::ROWP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *slab = smem + wave * (64 * ROWP);
    const bool accum = (flags & GS_FLAG_ACCUMULATE_GRADS) != 0u;
    auto put = [accum](float *p, float v) { *p = accum ? *p + v : v; };
    const int64_t g0 = ((int64_t)blockIdx.x * (ShSplit<K>::kBlock / 64) + wave) * 64;
    const int cnt = g0 < N ? min(64, (int)(N - g0)) : 0;
    if (lane < cnt) {
        const int64_t g = g0 + lane;
        float acc[3 * K];
#pragma unroll
        for (int i = 0; i < 3 * K; i++) acc[i] = 0.0f;
        const float mx = means[3 * g], my = means[3 * g + 1], mz = means[3 * g + 2];
        float n0 = 0.0f, n1 = 0.0f, n2 = 0.0f;   // the next camera's cotangent, requested one iteration ahead
        if (n_cams > 0) {
            const float *vc = v_colors + 3 * g;
            n0 = vc[0]; n1 = vc[1]; n2 = vc[2];
        }
        for (int c = 0; c < n_cams; c++) {
            const float v0 = n0, v1 = n1, v2 = n2;
            if (c + 1 < n_cams) {
                const float *vc = v_colors + (size_t)(c + 1) * v_stride + 3 * g;
                n0 = vc[0]; n1 = vc[1]; n2 = vc[2];
            }
            if (v0 == 0.0f && v1 == 0.0f && v2 == 0.0f) continue;  // culled / unseen from camera c
            const float *cp = cam_pos + (size_t)c * cam_stride;
            float x = mx - cp[0], y = my - cp[1], z = mz - cp[2];   // view_dir()
            const float nrm = sqrtf(x * x + y * y + z * z);
            x /= nrm; y /= nrm; z /= nrm;
            float r[25];
            sh_basis(nb, x, y, z, r);
#pragma unroll
            for (int b = 0; b < K; b++) {
                acc[3 * b + 0] = acc[3 * b + 0] + r[b] * v0;
                acc[3 * b + 1] = acc[3 * b + 1] + r[b] * v1;
                acc[3 * b + 2] = acc[3 * b + 2] + r[b] * v2;
            }
        }
        put(v_dc + 3 * g + 0, acc[0]);
        put(v_dc + 3 * g + 1, acc[1]);
        put(v_dc + 3 * g + 2, acc[2]);
        float *row = slab + lane * ROWP;
#pragma unroll
        for (int i = 0; i < ROW; i++) row[i] = acc[3 + i];
    }
    __syncthreads();
    if constexpr (ROW > 0) {
        float *dst = v_rest + g0 * ROW;
        const int total = cnt * ROW;
        for (int i = lane * 4; i < total; i += 64 * 4) {
            if (i + 3 < total) {
                float e[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int idx = i + k;
                    e[k] = slab[(idx / ROW) * ROWP + (idx % ROW)];
                }
                float4_u v;
                v.x = e[0]; v.y = e[1]; v.z = e[2]; v.w = e[3];
                if (accum) {
                    v = v + *reinterpret_cast<float4_u *>(dst + i);
                    *reinterpret_cast<float4_u *>(dst + i) = v;
                } else {
                    // 180 B per Gaussian that nothing on the path reads back: streamed past the
                    // caches, so that their write-back does not sit on the NEXT kernel's reads (the
                    // SH forward of the following step: 53 -> 40 us in the timed loop; this kernel
                    // +5 us; the training iteration, whose Adam step reads them at once, unchanged)
                    // (four scalar builtins: the back end merges them into one global_store_dwordx4 nt;
                    // on the 4-byte-aligned vector type the hint is dropped)
                    __builtin_nontemporal_store(v.x, dst + i);
                    __builtin_nontemporal_store(v.y, dst + i + 1);
                    __builtin_nontemporal_store(v.z, dst + i + 2);
                    __builtin_nontemporal_store(v.w, dst + i + 3);
                }
            } else {
                for (int idx = i; idx < total; idx++)
                    put(dst + idx, slab[(idx / ROW) * ROWP + (idx % ROW)]);
            }
        }
    }
}

template <int K>
static int launch_sh_backward_cameras(int N, int nb, int n_cams, const float *means,
                                      const float *cam_pos, size_t cam_stride, const float *v_colors,
                                      size_t v_stride, float *v_dc, float *v_rest, uint32_t flags,
                                      hipStream_t s) {
    constexpr int BLK = ShSplit<K>::kBlock;
    const size_t lds = (size_t)BLK * ShSplit<K>::ROWP * sizeof(float);
    GS_LAUNCH(k_sh_backward_cameras<K>, dim3((N + BLK - 1) / BLK), dim3(BLK), lds, s, N, nb,
                       n_cams, means, cam_pos, cam_stride, v_colors, v_stride, v_dc, v_rest, flags);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

template <int K>
static int launch_gaussian_forward(const CamArgs &cam, const float *vm_dev, const float *pm_dev, int N,
                                   int nb, const float *means, const float *scales,
                                   const float *quats, const float *opacities, const float *dc,
                                   const float *rest, const float *cp, float *packed, float *depths,
                                   int32_t *radii, float *rgb_raw, float *xys, uint32_t flags,
                                   hipStream_t s) {
    constexpr int BLK = ShSplit<K>::kBlock;
    const size_t lds = (size_t)BLK * ShSplit<K>::ROWP * sizeof(float);
    const bool dev = on_device(cp);
    GS_LAUNCH(k_gaussian_forward<K>, dim3((N + BLK - 1) / BLK), dim3(BLK), lds, s, cam, vm_dev,
                       pm_dev, N, nb, means, scales, quats, opacities, dc, rest, dev ? 0.f : cp[0],
                       dev ? 0.f : cp[1], dev ? 0.f : cp[2], dev ? cp : nullptr,
                       reinterpret_cast<float4 *>(packed), depths, radii, rgb_raw, xys, flags);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

template <int K>
static int launch_gaussian_backward(const CamArgs &cam, const float *vm_dev, const float *pm_dev, int N,
                                    int nb, const float *means, const float *scales,
                                    const float *quats, const float *opacities, const float *cp,
                                    const int32_t *radii, const float *rgb_raw, void *records,
                                    float *v_means, float *v_scales, float *v_quats, float *v_opacity,
                                    float *v_dc, float *v_rest, float *v_xy, uint32_t flags,
                                    hipStream_t s) {
    constexpr int BLK = ShSplit<K>::kBlock;
    const size_t lds = (size_t)BLK * ShSplit<K>::ROWP * sizeof(float);
    const bool dev = on_device(cp);
    GS_LAUNCH(k_gaussian_backward<K>, dim3((N + BLK - 1) / BLK), dim3(BLK), lds, s, cam, vm_dev,
                       pm_dev, N, nb, means, scales, quats, opacities, dev ? 0.f : cp[0],
                       dev ? 0.f : cp[1], dev ? 0.f : cp[2], dev ? cp : nullptr, radii, rgb_raw,
                       reinterpret_cast<float4 *>(records), v_means, v_scales, v_quats, v_opacity, v_dc,
                       v_rest, v_xy, flags);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

}  // namespace gs

extern "C" int gs_gaussian_forward(const GsCamera *cam, const float *viewmat_dev,
                                   const float *projmat_dev, int N, int K, int degrees_to_use,
                                   const float *means, const float *scales, const float *quats,
                                   const float *opacities, const float *features_dc,
                                   const float *features_rest, const float *cam_pos, float *packed,
                                   float *depths, int32_t *radii, float *rgb_raw, float *xys,
                                   uint32_t flags, gs_stream_t stream) {
    GS_TRACE("gs_gaussian_forward");
    const int deg = gs::deg_from_bases(K);
    if (!cam || N < 0 || deg < 0 || degrees_to_use < 0 || degrees_to_use > deg)
        return GS_ERR_INVALID_ARGUMENT;
    if (N == 0) return GS_OK;
    if (!means || !scales || !quats || !opacities || !features_dc || (K > 1 && !features_rest) ||
        !cam_pos || !packed || !depths || !radii || !rgb_raw)
        return GS_ERR_INVALID_ARGUMENT;
    if (cam->img_width <= 0 || cam->img_height <= 0) return GS_ERR_INVALID_ARGUMENT;
    if (cam->img_width > 65535 || cam->img_height > 65535) return GS_ERR_UNSUPPORTED;
    if ((uintptr_t)packed & 15u) return GS_ERR_INVALID_ARGUMENT;  // the record is three aligned float4s
    // (quats / features_rest may be slices of a flat parameter buffer: 4-byte alignment suffices)
    const gs::CamArgs a = gs::make_cam(cam);
    const int nb = gs::num_bases(degrees_to_use);
    hipStream_t s = (hipStream_t)stream;
#define GS_FWD(KK)                                                                                    \
    return gs::launch_gaussian_forward<KK>(a, viewmat_dev, projmat_dev, N, nb, means, scales, quats,  \
                                           opacities, features_dc, features_rest, cam_pos, packed,    \
                                           depths, radii, rgb_raw, xys, flags, s)
    switch (K) {
    case 1: GS_FWD(1);
    case 4: GS_FWD(4);
    case 9: GS_FWD(9);
    case 16: {  // SH forward and projection + record as one launch of two kinds of workgroups
        const bool dev = gs::on_device(cam_pos);
        GS_LAUNCH(gs::k_sh_project_pack16, dim3(5 * ((N + 255) / 256)), dim3(256), 0, s, a,
                           viewmat_dev, projmat_dev, N, nb, means, scales, quats, opacities, features_dc,
                           features_rest, dev ? 0.f : cam_pos[0], dev ? 0.f : cam_pos[1],
                           dev ? 0.f : cam_pos[2], dev ? cam_pos : nullptr,
                           reinterpret_cast<float4 *>(packed), depths, radii, rgb_raw, xys, flags);
        GS_LAUNCH_CHECK();
        return GS_OK;
    }
    default: GS_FWD(25);
    }
#undef GS_FWD
}

extern "C" int gs_gaussian_backward(const GsCamera *cam, const float *viewmat_dev,
                                    const float *projmat_dev, int N, int K, int degrees_to_use,
                                    const float *means, const float *scales, const float *quats,
                                    const float *opacities, const float *cam_pos,
                                    const int32_t *radii, const float *rgb_raw, void *records,
                                    size_t records_bytes, float *v_means, float *v_scales,
                                    float *v_quats, float *v_opacity, float *v_dc, float *v_rest,
                                    float *v_xy, uint32_t flags, gs_stream_t stream) {
    GS_TRACE("gs_gaussian_backward");
    const int deg = gs::deg_from_bases(K);
    if (!cam || N < 0 || deg < 0 || degrees_to_use < 0 || degrees_to_use > deg)
        return GS_ERR_INVALID_ARGUMENT;
    if (N == 0) return GS_OK;
    const bool emit = (flags & GS_FLAG_EMIT_VCOLOR) != 0u;   // v_dc <- colour cotangent, v_rest untouched
    if (!means || !scales || !quats || !opacities || !cam_pos || !radii || !rgb_raw || !records ||
        !v_means || !v_scales || !v_quats || !v_opacity || !v_dc || (K > 1 && !v_rest && !emit))
        return GS_ERR_INVALID_ARGUMENT;
    if (records_bytes < (size_t)N * gs::kRec * sizeof(float)) return GS_ERR_WORKSPACE;
    if ((uintptr_t)records & 63u) return GS_ERR_INVALID_ARGUMENT;
    const gs::CamArgs a = gs::make_cam(cam);
    const int nb = gs::num_bases(degrees_to_use);
    hipStream_t s = (hipStream_t)stream;
#define GS_BWD(KK)                                                                                    \
    return gs::launch_gaussian_backward<KK>(a, viewmat_dev, projmat_dev, N, nb, means, scales, quats, \
                                            opacities, cam_pos, radii, rgb_raw, records, v_means,     \
                                            v_scales, v_quats, v_opacity, v_dc, v_rest, v_xy, flags, s)
    switch (emit ? 1 : K) {   // (the K = 1 instantiation has no SH rows to move)
    case 1: GS_BWD(1);
    case 4: GS_BWD(4);
    case 9: GS_BWD(9);
    case 16: GS_BWD(16);
    default: GS_BWD(25);
    }
#undef GS_BWD
}

extern "C" int gs_sh_backward_cameras(int N, int K, int degrees_to_use, int n_cams, const float *means,
                                      const float *cam_pos_dev, size_t cam_pos_stride,
                                      const float *v_colors, size_t v_colors_stride, float *v_dc,
                                      float *v_rest, uint32_t flags, gs_stream_t stream) {
    GS_TRACE("gs_sh_backward_cameras");
    const int deg = gs::deg_from_bases(K);
    if (N < 0 || deg < 0 || degrees_to_use < 0 || degrees_to_use > deg || n_cams < 0)
        return GS_ERR_INVALID_ARGUMENT;
    if (N == 0) return GS_OK;
    if (!means || (n_cams > 0 && (!cam_pos_dev || !v_colors)) || !v_dc || (K > 1 && !v_rest))
        return GS_ERR_INVALID_ARGUMENT;
    if (cam_pos_stride < 3 || v_colors_stride < (size_t)N * 3) return GS_ERR_INVALID_ARGUMENT;
    const int nb = gs::num_bases(degrees_to_use);
    hipStream_t s = (hipStream_t)stream;
#define GS_SHC(KK)                                                                                   \
    return gs::launch_sh_backward_cameras<KK>(N, nb, n_cams, means, cam_pos_dev, cam_pos_stride,     \
                                              v_colors, v_colors_stride, v_dc, v_rest, flags, s)
    switch (K) {
    case 1: GS_SHC(1);
    case 4: GS_SHC(4);
    case 9: GS_SHC(9);
    case 16: GS_SHC(16);
    default: GS_SHC(25);
    }
#undef GS_SHC
}
