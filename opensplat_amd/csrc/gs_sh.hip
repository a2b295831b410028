// gs_sh.hip — view-dependent colour from spherical-harmonic coefficients, and its VJP.
//
// Replaces compute_sh_forward_kernel / compute_sh_backward_kernel (reference
// rasterizer/gsplat/sh.cuh:52-260); basis functions and constants as the CPU oracle
// (rasterizer/gsplat-cpu/gsplat_cpu.cpp:379-407, :424-486).
//
// Roofline: HBM streaming.  Forward reads 12 + 12K B and writes 12 B per Gaussian (204 B at
// K = 16), backward reads 24 B and writes 12K B.  The coefficient tensor is [N, K, 3] fp32 — an
// AoS layout fixed by the operator surface — so a lane-per-Gaussian mapping would stride lanes
// by 12K bytes.  Instead each wave moves its 64 Gaussians' coefficients as one contiguous
// 64*12K-byte slab with fully coalesced 16-byte loads through LDS (rows padded by one dword so
// the per-lane row walk is bank-conflict free), then each lane reduces its own row.
#include "gs_gaussian.h"

namespace gs {

template <int K>
struct ShCfg {
    static constexpr int kBlock = (K > 16) ? 128 : 256;
};

// K is a template parameter so the row length (3K dwords, +1 pad) is static.
template <int K>
__global__ void __launch_bounds__(ShCfg<K>::kBlock)
k_sh_forward(int N, int nb, const float *__restrict__ dirs, const float *__restrict__ coeffs,
             float *__restrict__ colors) {
    constexpr int ROW = 3 * K;       // dwords per Gaussian
    constexpr int ROWP = ROW | 1;    // odd stride -> conflict-free per-lane row walk
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *slab = smem + wave * (64 * ROWP);
    const int64_t g0 = ((int64_t)blockIdx.x * (ShCfg<K>::kBlock / 64) + wave) * 64;
    const int cnt = g0 < N ? min(64, (int)(N - g0)) : 0;  // 0: wave has no Gaussians (tail)
    const float *src = coeffs + g0 * ROW;
    const int total = cnt * ROW;  // dwords in this wave's slab (src is 16-B aligned: 64*ROW*4*g)
    // coalesced 16-byte loads, scattered into the padded rows
    for (int i = lane * 4; i < total; i += 64 * 4) {
        if (i + 3 < total) {
            float4 v = *reinterpret_cast<const float4 *>(src + i);
            float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int idx = i + k;
                slab[(idx / ROW) * ROWP + (idx % ROW)] = e[k];
            }
        } else {
            for (int idx = i; idx < total; idx++) slab[(idx / ROW) * ROWP + (idx % ROW)] = src[idx];
        }
    }
    __syncthreads();  // slabs are wave-private; the barrier only orders LDS writes -> reads
    if (lane >= cnt) return;
    const int64_t g = g0 + lane;
    float r[25];
    sh_basis(nb, dirs[3 * g], dirs[3 * g + 1], dirs[3 * g + 2], r);
    const float *row = slab + lane * ROWP;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int b = 0; b < K; b++) {
        c0 += r[b] * row[3 * b + 0];
        c1 += r[b] * row[3 * b + 1];
        c2 += r[b] * row[3 * b + 2];
    }
    colors[3 * g + 0] = c0;
    colors[3 * g + 1] = c1;
    colors[3 * g + 2] = c2;
}

template <int K>
__global__ void __launch_bounds__(ShCfg<K>::kBlock)
k_sh_backward(int N, int nb, const float *__restrict__ dirs, const float *__restrict__ v_colors,
              float *__restrict__ v_coeffs) {
    constexpr int ROW = 3 * K;
    constexpr int ROWP = ROW | 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *slab = smem + wave * (64 * ROWP);
    const int64_t g0 = ((int64_t)blockIdx.x * (ShCfg<K>::kBlock / 64) + wave) * 64;
    const int cnt = g0 < N ? min(64, (int)(N - g0)) : 0;
    if (lane < cnt) {
        const int64_t g = g0 + lane;
        float r[25];
        sh_basis(nb, dirs[3 * g], dirs[3 * g + 1], dirs[3 * g + 2], r);
        float v0 = v_colors[3 * g], v1 = v_colors[3 * g + 1], v2 = v_colors[3 * g + 2];
        float *row = slab + lane * ROWP;
#pragma unroll
        for (int b = 0; b < K; b++) {
            row[3 * b + 0] = r[b] * v0;
            row[3 * b + 1] = r[b] * v1;
            row[3 * b + 2] = r[b] * v2;
        }
    }
    __syncthreads();
    float *dst = v_coeffs + g0 * ROW;
    const int total = cnt * ROW;
    for (int i = lane * 4; i < total; i += 64 * 4) {
        if (i + 3 < total) {
            float e[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int idx = i + k;
                e[k] = slab[(idx / ROW) * ROWP + (idx % ROW)];
            }
            *reinterpret_cast<float4 *>(dst + i) = make_float4(e[0], e[1], e[2], e[3]);
        } else {
            for (int idx = i; idx < total; idx++) dst[idx] = slab[(idx / ROW) * ROWP + (idx % ROW)];
        }
    }
}

// K = 16 (degree 3, the configuration OpenSplat trains at): FOUR lanes per Gaussian, each loads its
// 48 contiguous bytes (bases 4q..4q+3 x rgb) as three float4 — a wave covers 16 Gaussians = 3 KiB of
// contiguous coefficients, every fetched byte is used, no LDS staging — and the four partial dot
// products are combined with two DPP quad permutes.
__global__ void __launch_bounds__(256)
k_sh_forward16_quad(int N, int nb, const float *__restrict__ dirs,
                    const float *__restrict__ coeffs, float *__restrict__ colors) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t g = t >> 2;
    const int q = (int)(t & 3);
    if (g >= N) return;  // whole quads drop out together
    float r[25];
    sh_basis(nb, dirs[3 * g], dirs[3 * g + 1], dirs[3 * g + 2], r);
    const float4 *p = reinterpret_cast<const float4 *>(coeffs + g * 48 + q * 12);
    const float4 a = p[0], b = p[1], c = p[2];
    const float r0 = q == 0 ? r[0] : q == 1 ? r[4] : q == 2 ? r[8] : r[12];
    const float r1 = q == 0 ? r[1] : q == 1 ? r[5] : q == 2 ? r[9] : r[13];
    const float r2 = q == 0 ? r[2] : q == 1 ? r[6] : q == 2 ? r[10] : r[14];
    const float r3 = q == 0 ? r[3] : q == 1 ? r[7] : q == 2 ? r[11] : r[15];
    float c0 = r0 * a.x + r1 * a.w + r2 * b.z + r3 * c.y;
    float c1 = r0 * a.y + r1 * b.x + r2 * b.w + r3 * c.z;
    float c2 = r0 * a.z + r1 * b.y + r2 * c.x + r3 * c.w;
    c0 += dpp_f<0xB1>(c0); c1 += dpp_f<0xB1>(c1); c2 += dpp_f<0xB1>(c2);  // quad_perm [1,0,3,2]
    c0 += dpp_f<0x4E>(c0); c1 += dpp_f<0x4E>(c1); c2 += dpp_f<0x4E>(c2);  // quad_perm [2,3,0,1]
    if (q == 0) {
        colors[3 * g + 0] = c0;
        colors[3 * g + 1] = c1;
        colors[3 * g + 2] = c2;
    }
}

template <int K>
static int launch_fwd(int N, int nb, const float *dirs, const float *coeffs, float *colors,
                      hipStream_t s) {
    constexpr int ROWP = (3 * K) | 1;
    constexpr int BLK = ShCfg<K>::kBlock;  // == Gaussians per block
    size_t lds = (size_t)BLK * ROWP * sizeof(float);
    int blocks = (N + BLK - 1) / BLK;
    GS_LAUNCH(k_sh_forward<K>, dim3(blocks), dim3(BLK), lds, s, N, nb, dirs, coeffs,
                       colors);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

template <int K>
static int launch_bwd(int N, int nb, const float *dirs, const float *v_colors, float *v_coeffs,
                      hipStream_t s) {
    constexpr int ROWP = (3 * K) | 1;
    constexpr int BLK = ShCfg<K>::kBlock;
    size_t lds = (size_t)BLK * ROWP * sizeof(float);
    int blocks = (N + BLK - 1) / BLK;
    GS_LAUNCH(k_sh_backward<K>, dim3(blocks), dim3(BLK), lds, s, N, nb, dirs,
                       v_colors, v_coeffs);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

// ---- fused-glue variants ---------------------------------------------------
// Coefficients arrive as OpenSplat stores them — features_dc[N,3] and features_rest[N,K-1,3]
// (model.hpp) — so the per-step torch::cat copy (model.cpp:114, 192 MB at C2) disappears; the view
// direction normalize(mean - cam_pos) (model.cpp:176-177) is computed in registers; the output is
// clamp_min(rgb + 0.5, 0) (model.cpp:192) plus the raw rgb the backward needs for the clamp mask.
// Layout trick as in k_sh_forward: a wave moves the 64 * 12(K-1)-byte slab of its 64 Gaussians'
// higher-band coefficients with coalesced 16-byte loads through LDS rows of odd stride.

template <int K>
__global__ void __launch_bounds__(ShSplit<K>::kBlock)
k_sh_forward_fused(int N, int nb, const float *__restrict__ means, float cx, float cy, float cz,
                   const float *__restrict__ cp_dev, const float *__restrict__ dc, const float *__restrict__ rest,
                   float *__restrict__ colors, float *__restrict__ rgb_raw) {
    constexpr int ROW = ShSplit<K>::ROW, ROWP = ShSplit<K>::ROWP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (cp_dev) { cx = cp_dev[0]; cy = cp_dev[1]; cz = cp_dev[2]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *slab = smem + wave * (64 * ROWP);
    const int64_t g0 = ((int64_t)blockIdx.x * (ShSplit<K>::kBlock / 64) + wave) * 64;
    const int cnt = g0 < N ? min(64, (int)(N - g0)) : 0;
    if constexpr (ROW > 0) {
        const float *src = rest + g0 * ROW;  // 16-B aligned: 64 * ROW * 4 bytes per wave
        const int total = cnt * ROW;
        for (int i = lane * 4; i < total; i += 64 * 4) {
            if (i + 3 < total) {
                float4 v = *reinterpret_cast<const float4 *>(src + i);
                float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int idx = i + k;
                    slab[(idx / ROW) * ROWP + (idx % ROW)] = e[k];
                }
            } else {
                for (int idx = i; idx < total; idx++) slab[(idx / ROW) * ROWP + (idx % ROW)] = src[idx];
            }
        }
    }
    __syncthreads();
    if (lane >= cnt) return;
    const int64_t g = g0 + lane;
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
    colors[3 * g + 0] = fmaxf(c0 + 0.5f, 0.0f);
    colors[3 * g + 1] = fmaxf(c1 + 0.5f, 0.0f);
    colors[3 * g + 2] = fmaxf(c2 + 0.5f, 0.0f);
}

template <int K>
__global__ void __launch_bounds__(ShSplit<K>::kBlock)
k_sh_backward_fused(int N, int nb, const float *__restrict__ means, float cx, float cy, float cz,
                    const float *__restrict__ cp_dev, const float *__restrict__ rgb_raw, const float *__restrict__ v_colors,
                    float *__restrict__ v_dc, float *__restrict__ v_rest) {
    constexpr int ROW = ShSplit<K>::ROW, ROWP = ShSplit<K>::ROWP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *slab = smem + wave * (64 * ROWP);
    const int64_t g0 = ((int64_t)blockIdx.x * (ShSplit<K>::kBlock / 64) + wave) * 64;
    const int cnt = g0 < N ? min(64, (int)(N - g0)) : 0;
    if (cp_dev) { cx = cp_dev[0]; cy = cp_dev[1]; cz = cp_dev[2]; }
    if (lane < cnt) {
        const int64_t g = g0 + lane;
        float x, y, z;
        view_dir(means, g, cx, cy, cz, x, y, z);
        float r[25];
        sh_basis(nb, x, y, z, r);
        // clamp_min(rgb + 0.5, 0) backward: gradient passes where rgb + 0.5 >= 0 (torch's mask)
        const float v0 = (rgb_raw[3 * g + 0] + 0.5f >= 0.0f) ? v_colors[3 * g + 0] : 0.0f;
        const float v1 = (rgb_raw[3 * g + 1] + 0.5f >= 0.0f) ? v_colors[3 * g + 1] : 0.0f;
        const float v2 = (rgb_raw[3 * g + 2] + 0.5f >= 0.0f) ? v_colors[3 * g + 2] : 0.0f;
        v_dc[3 * g + 0] = r[0] * v0;
        v_dc[3 * g + 1] = r[0] * v1;
        v_dc[3 * g + 2] = r[0] * v2;
        float *row = slab + lane * ROWP;
#pragma unroll
        for (int b = 1; b < K; b++) {
            row[3 * (b - 1) + 0] = r[b] * v0;
            row[3 * (b - 1) + 1] = r[b] * v1;
            row[3 * (b - 1) + 2] = r[b] * v2;
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
                *reinterpret_cast<float4 *>(dst + i) = make_float4(e[0], e[1], e[2], e[3]);
            } else {
                for (int idx = i; idx < total; idx++) dst[idx] = slab[(idx / ROW) * ROWP + (idx % ROW)];
            }
        }
    }
}

// K = 16 fused forward, four lanes per Gaussian.  Lane q < 3 combines the coefficients 4q+1 .. 4q+4
// = floats 12q .. 12q+11 of the Gaussian's features_rest row; lane 3 takes coefficients 13, 14, 15
// (floats 36 .. 44) and the dc term.  Every lane issues the SAME three 16-byte loads — lane 3's third
// one starts at float 41 instead of 44, so that it ends with the row and never leaves the tensor —
// plus dc and the mean, all before any arithmetic: the first version chose between two differently
// shaped load sets in a divergent branch, which the compiler could only place behind the view
// direction's divisions, i.e. two dependent memory round trips per wave (3.9 TB/s; a plain streaming
// read of the same 216 MB reaches 6.0, scripts/ubench/stream_bw.hip).  Rows of features_rest are
// 180 bytes, so the 16-byte loads are only 4-byte aligned — gfx950 global loads allow that.
__global__ void __launch_bounds__(256)
k_sh_forward_fused16_quad(int N, int nb, const float *__restrict__ means, float cx, float cy,
                          float cz, const float *__restrict__ cp_dev, const float *__restrict__ dc,
                          const float *__restrict__ rest, float *__restrict__ colors,
                          float *__restrict__ rgb_raw) {
    sh_forward16_quad_body((int64_t)blockIdx.x * blockDim.x + threadIdx.x, N, nb, means, cx, cy, cz, cp_dev, dc,
                           rest, colors, 3, rgb_raw);
}

// Host launcher shared with gs_fused.hip (gs_gaussian_forward at K = 16); colors may be NULL.
int launch_sh_forward_fused16_quad(int N, int nb, const float *means, const float *cam_pos,
                                   const float *features_dc, const float *features_rest,
                                   float *colors, float *rgb_raw, hipStream_t s) {
    const bool dev = on_device(cam_pos);
    const int64_t threads = (int64_t)N * 4;
    GS_LAUNCH(k_sh_forward_fused16_quad, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       s, N, nb, means, dev ? 0.f : cam_pos[0], dev ? 0.f : cam_pos[1],
                       dev ? 0.f : cam_pos[2], dev ? cam_pos : nullptr, features_dc, features_rest,
                       colors, rgb_raw);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

template <int K>
static int launch_fwd_fused(int N, int nb, const float *means, const float *cp, const float *dc,
                            const float *rest, float *colors, float *rgb_raw, hipStream_t s) {
    constexpr int BLK = ShSplit<K>::kBlock;
    size_t lds = (size_t)BLK * ShSplit<K>::ROWP * sizeof(float);
    const bool dev = on_device(cp);
    GS_LAUNCH(k_sh_forward_fused<K>, dim3((N + BLK - 1) / BLK), dim3(BLK), lds, s, N, nb,
                       means, dev ? 0.f : cp[0], dev ? 0.f : cp[1], dev ? 0.f : cp[2],
                       dev ? cp : nullptr, dc, rest, colors, rgb_raw);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

template <int K>
static int launch_bwd_fused(int N, int nb, const float *means, const float *cp, const float *rgb_raw,
                            const float *v_colors, float *v_dc, float *v_rest, hipStream_t s) {
    constexpr int BLK = ShSplit<K>::kBlock;
    size_t lds = (size_t)BLK * ShSplit<K>::ROWP * sizeof(float);
    const bool dev = on_device(cp);
    GS_LAUNCH(k_sh_backward_fused<K>, dim3((N + BLK - 1) / BLK), dim3(BLK), lds, s, N, nb,
                       means, dev ? 0.f : cp[0], dev ? 0.f : cp[1], dev ? 0.f : cp[2],
                       dev ? cp : nullptr, rgb_raw, v_colors, v_dc, v_rest);
    GS_LAUNCH_CHECK();
    return GS_OK;
}


}  // namespace gs

extern "C" int gs_sh_forward(int N, int K, int degrees_to_use, const float *dirs,
                             const float *coeffs, float *colors, gs_stream_t stream) {
    int deg = gs::deg_from_bases(K);
    if (N < 0 || deg < 0 || degrees_to_use < 0 || degrees_to_use > deg) return GS_ERR_INVALID_ARGUMENT;
    if (N == 0) return GS_OK;
    if (!dirs || !coeffs || !colors) return GS_ERR_INVALID_ARGUMENT;
    if ((uintptr_t)coeffs & 15u) return GS_ERR_INVALID_ARGUMENT;  // 16-byte vector loads
    int nb = gs::num_bases(degrees_to_use);
    hipStream_t s = (hipStream_t)stream;
    switch (K) {
    case 1: return gs::launch_fwd<1>(N, nb, dirs, coeffs, colors, s);
    case 4: return gs::launch_fwd<4>(N, nb, dirs, coeffs, colors, s);
    case 9: return gs::launch_fwd<9>(N, nb, dirs, coeffs, colors, s);
    case 16: {
        const int64_t threads = (int64_t)N * 4;
        GS_LAUNCH(gs::k_sh_forward16_quad, dim3((unsigned)((threads + 255) / 256)), dim3(256),
                           0, s, N, nb, dirs, coeffs, colors);
        GS_LAUNCH_CHECK();
        return GS_OK;
    }
    default: return gs::launch_fwd<25>(N, nb, dirs, coeffs, colors, s);
    }
}

extern "C" int gs_sh_backward(int N, int K, int degrees_to_use, const float *dirs,
                              const float *v_colors, float *v_coeffs, gs_stream_t stream) {
    int deg = gs::deg_from_bases(K);
    if (N < 0 || deg < 0 || degrees_to_use < 0 || degrees_to_use > deg) return GS_ERR_INVALID_ARGUMENT;
    if (N == 0) return GS_OK;
    if (!dirs || !v_colors || !v_coeffs) return GS_ERR_INVALID_ARGUMENT;
    if ((uintptr_t)v_coeffs & 15u) return GS_ERR_INVALID_ARGUMENT;  // 16-byte vector stores
    int nb = gs::num_bases(degrees_to_use);
    hipStream_t s = (hipStream_t)stream;
    switch (K) {
    case 1: return gs::launch_bwd<1>(N, nb, dirs, v_colors, v_coeffs, s);
    case 4: return gs::launch_bwd<4>(N, nb, dirs, v_colors, v_coeffs, s);
    case 9: return gs::launch_bwd<9>(N, nb, dirs, v_colors, v_coeffs, s);
    case 16: return gs::launch_bwd<16>(N, nb, dirs, v_colors, v_coeffs, s);
    default: return gs::launch_bwd<25>(N, nb, dirs, v_colors, v_coeffs, s);
    }
}

extern "C" int gs_sh_forward_fused(int N, int K, int degrees_to_use, const float *means,
                                   const float *cam_pos, const float *features_dc,
                                   const float *features_rest, float *colors, float *rgb_raw,
                                   gs_stream_t stream) {
    int deg = gs::deg_from_bases(K);
    if (N < 0 || deg < 0 || degrees_to_use < 0 || degrees_to_use > deg) return GS_ERR_INVALID_ARGUMENT;
    if (N == 0) return GS_OK;
    if (!means || !cam_pos || !features_dc || !colors || !rgb_raw || (K > 1 && !features_rest))
        return GS_ERR_INVALID_ARGUMENT;
    if ((uintptr_t)features_rest & 15u) return GS_ERR_INVALID_ARGUMENT;  // 16-byte vector loads
    int nb = gs::num_bases(degrees_to_use);
    hipStream_t s = (hipStream_t)stream;
    switch (K) {
    case 1: return gs::launch_fwd_fused<1>(N, nb, means, cam_pos, features_dc, features_rest, colors, rgb_raw, s);
    case 4: return gs::launch_fwd_fused<4>(N, nb, means, cam_pos, features_dc, features_rest, colors, rgb_raw, s);
    case 9: return gs::launch_fwd_fused<9>(N, nb, means, cam_pos, features_dc, features_rest, colors, rgb_raw, s);
    case 16:
        return gs::launch_sh_forward_fused16_quad(N, nb, means, cam_pos, features_dc, features_rest,
                                                  colors, rgb_raw, s);
    default: return gs::launch_fwd_fused<25>(N, nb, means, cam_pos, features_dc, features_rest, colors, rgb_raw, s);
    }
}

extern "C" int gs_sh_backward_fused(int N, int K, int degrees_to_use, const float *means,
                                    const float *cam_pos, const float *rgb_raw,
                                    const float *v_colors, float *v_dc, float *v_rest,
                                    gs_stream_t stream) {
    int deg = gs::deg_from_bases(K);
    if (N < 0 || deg < 0 || degrees_to_use < 0 || degrees_to_use > deg) return GS_ERR_INVALID_ARGUMENT;
    if (N == 0) return GS_OK;
    if (!means || !cam_pos || !rgb_raw || !v_colors || !v_dc || (K > 1 && !v_rest))
        return GS_ERR_INVALID_ARGUMENT;
    if ((uintptr_t)v_rest & 15u) return GS_ERR_INVALID_ARGUMENT;  // 16-byte vector stores
    int nb = gs::num_bases(degrees_to_use);
    hipStream_t s = (hipStream_t)stream;
    switch (K) {
    case 1: return gs::launch_bwd_fused<1>(N, nb, means, cam_pos, rgb_raw, v_colors, v_dc, v_rest, s);
    case 4: return gs::launch_bwd_fused<4>(N, nb, means, cam_pos, rgb_raw, v_colors, v_dc, v_rest, s);
    case 9: return gs::launch_bwd_fused<9>(N, nb, means, cam_pos, rgb_raw, v_colors, v_dc, v_rest, s);
    case 16: return gs::launch_bwd_fused<16>(N, nb, means, cam_pos, rgb_raw, v_colors, v_dc, v_rest, s);
    default: return gs::launch_bwd_fused<25>(N, nb, means, cam_pos, rgb_raw, v_colors, v_dc, v_rest, s);
    }
}
