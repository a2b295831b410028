// gs_raster.hip — per-tile alpha compositing (forward) and its gradient walk (backward).
//
// Replaces rasterize_forward / rasterize_backward_kernel (reference
// rasterizer/gsplat/forward.cu:256-378, backward.cu:161-355).  Per-pixel recurrence, thresholds
// and clamp constants are the CPU oracle's (rasterizer/gsplat-cpu/gsplat_cpu.cpp:188-240 forward,
// :313-373 backward), evaluated in the same operation order without FMA contraction and with a
// glibc-bit-exact expf, so that contributor sets, final_Ts and the image equal gsplat-cpu's
// bit for bit on identical inputs; the backward differs only in the order in which per-pixel
// terms are summed into a Gaussian's gradient.
//
// CDNA4 mapping (this is not the reference's 16x16-threads-per-tile CUDA layout):
//   * one 64-lane wavefront == one workgroup == one 16x16 tile; every lane owns FOUR pixels
//     (column lane&15, rows (lane>>4) + 4k, k = 0..3).  The per-Gaussian record is read from LDS
//     once per wave (broadcast ds_read_b128) and amortised over 4 pixels per lane, which keeps
//     the LDS pipe far below the VALU pipe; 4 independent pixel chains per lane give the ILP
//     that a 2-cycle-issue SIMD-32 needs.
//   * no workgroup barriers between waves at all: the tile's sorted list is staged 64 entries
//     at a time by the wave itself, and early termination is a 64-bit ballot.
//   * the pixel-in-rectangle test of the CPU oracle costs 2 VALU ops: each staged entry carries
//     a 16+16-bit column/row mask local to the tile, each pixel a constant 2-bit probe.
//   * a 4-row strip that the rectangle does not touch is skipped with a scalar branch, and the
//     fp64 exponential is only issued when some lane of the strip can pass alpha >= 1/255
//     (sigma <= ln(255*opacity) + 1e-3, precomputed per entry).
//   * backward: per-entry partial gradients are summed over the lane's 4 pixels in registers,
//     then over the wave with DPP row reductions + v_readlane, and ONE lane issues the 9
//     global_atomic_add_f32 per (tile, Gaussian) — 4-8x fewer atomics than the reference's
//     per-32-lane-warp scheme.
//
// Roofline: neither kernel is HBM-bound.  Work is ~256 pixel x Gaussian evaluations per sorted
// tile entry (VALU + LDS broadcast); HBM traffic is one 48-byte gather per entry plus 20 B per
// pixel.  DESIGN.md states the algorithmic bytes used for roofline.achieved.
#include "gs_device.h"

namespace gs {

constexpr int kChunk = 64;  // entries staged per pass == wave width

// LDS image of one staged entry (48 B, three ds_read_b128):
//   a = {x, y, conic A, conic B}   b = {conic C, opacity, sigma_max, mask bits}   c = {r, g, b, id}
struct __attribute__((aligned(16))) Staged {
    float4 a, b, c;
};

__device__ __forceinline__ uint32_t tile_mask(uint32_t rx, uint32_t ry, int tile_x0, int tile_y0) {
    int x0 = (int)(rx & 0xFFFF) - tile_x0, x1 = (int)(rx >> 16) - tile_x0;
    int y0 = (int)(ry & 0xFFFF) - tile_y0, y1 = (int)(ry >> 16) - tile_y0;
    x0 = min(max(x0, 0), GS_TILE); x1 = min(max(x1, 0), GS_TILE);
    y0 = min(max(y0, 0), GS_TILE); y1 = min(max(y1, 0), GS_TILE);
    uint32_t cm = (x1 > x0) ? ((1u << x1) - (1u << x0)) : 0u;
    uint32_t rm = (y1 > y0) ? ((1u << y1) - (1u << y0)) : 0u;
    return (cm && rm) ? (cm | (rm << 16)) : 0u;
}

// Load entry `idx` of the sorted list and convert it to its staged form for tile (tx, ty).
__device__ __forceinline__ void stage_entry(Staged *dst, int idx, const int32_t *__restrict__ ids,
                                            const float4 *__restrict__ packed, int tile_x0,
                                            int tile_y0) {
    int g = ids[idx];
    float4 p0 = packed[3 * (size_t)g + 0];
    float4 p1 = packed[3 * (size_t)g + 1];
    float4 p2 = packed[3 * (size_t)g + 2];
    float opac = p1.y;
    // alpha = opac * exp(-sigma) >= 1/255 needs sigma <= ln(255 * opac); the margin makes the
    // skip strictly conservative w.r.t. rounding of the log, the exp and the product.
    float smax = (opac > 0.0f) ? (__logf(255.0f * opac) + 1.0e-3f) : -1.0f;
    uint32_t m = tile_mask(__float_as_uint(p2.y), __float_as_uint(p2.z), tile_x0, tile_y0);
    dst->a = p0;
    dst->b = make_float4(p1.x, opac, smax, __uint_as_float(m));
    dst->c = make_float4(p1.z, p1.w, p2.x, __int_as_float(g));
}

// ---------------------------------------------------------------------------------------------
template <bool EXACT>
__global__ void __launch_bounds__(64)
k_rasterize_forward(int W, int H, int tiles_x, int num_tiles, const int32_t *__restrict__ ids,
                    const int2 *__restrict__ bins, const float4 *__restrict__ packed, float bg0,
                    float bg1, float bg2, float *__restrict__ out_img,
                    float *__restrict__ final_Ts, int32_t *__restrict__ final_idx) {
    __shared__ Staged stage[kChunk];
    __shared__ uint64_t exp_tab[32];
    const int lane = threadIdx.x;
    const int tile = xcd_swizzle(blockIdx.x, num_tiles);
    const int tile_x0 = (tile % tiles_x) * GS_TILE, tile_y0 = (tile / tiles_x) * GS_TILE;
    if (EXACT && lane < 32) exp_tab[lane] = kExp2fTab[lane];

    const int lx = lane & 15, ly = lane >> 4;
    const int px = tile_x0 + lx;
    const float pxf = (float)px;
    float pyf[4], T[4], acc[4][3];
    int last[4];
    bool done[4];
    uint32_t probe[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int py = tile_y0 + ly + 4 * k;
        pyf[k] = (float)py;
        T[k] = 1.0f;
        acc[k][0] = acc[k][1] = acc[k][2] = 0.0f;
        last[k] = -1;
        done[k] = !(px < W && py < H);
        probe[k] = (1u << lx) | (1u << (16 + ly + 4 * k));
    }

    const int2 range = bins[tile];
    for (int c0 = range.x; c0 < range.y; c0 += kChunk) {
        if (__ballot(!(done[0] && done[1] && done[2] && done[3])) == 0ull) break;
        __syncthreads();  // previous chunk fully consumed (single-wave workgroup: cheap)
        if (c0 + lane < range.y) stage_entry(&stage[lane], c0 + lane, ids, packed, tile_x0, tile_y0);
        __syncthreads();
        const int n = min(kChunk, range.y - c0);
        for (int t = 0; t < n; t++) {
            const float4 ea = stage[t].a;
            const float4 eb = stage[t].b;
            const uint32_t mask = __builtin_amdgcn_readfirstlane(__float_as_uint(eb.w));
            if (mask == 0u) continue;
            const float dx = ea.x - pxf;
            const float Adx = ea.z * dx;       // A * xCam
            const float Bdx = ea.w * dx;       // B * xCam
            const float Adxdx = Adx * dx;      // A * xCam * xCam
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (((mask >> (16 + 4 * k)) & 0xFu) == 0u) continue;  // scalar: strip untouched
                const float dy = ea.y - pyf[k];
                // sigma = 0.5f * (A*x*x + C*y*y) + B*x*y, gsplat_cpu.cpp:213-217
                const float sigma = 0.5f * (Adxdx + eb.x * dy * dy) + Bdx * dy;
                const bool need = !done[k] && ((mask & probe[k]) == probe[k]) && (sigma >= 0.0f) &&
                                  (sigma <= eb.z);
                if (__ballot(need) == 0ull) continue;
                if (need) {
                    const float alpha = fminf(0.999f, eb.y * gs_exp<EXACT>(-sigma, exp_tab));
                    if (alpha >= (1.0f / 255.0f)) {
                        const float nextT = T[k] * (1.0f - alpha);
                        if (nextT <= 1e-4f) {
                            done[k] = true;  // this pixel is done; the Gaussian is not rendered
                        } else {
                            const float4 ec = stage[t].c;
                            const float vis = alpha * T[k];
                            acc[k][0] += vis * ec.x;
                            acc[k][1] += vis * ec.y;
                            acc[k][2] += vis * ec.z;
                            T[k] = nextT;
                            last[k] = c0 + t;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int py = tile_y0 + ly + 4 * k;
        if (px < W && py < H) {
            const size_t pix = (size_t)py * W + px;
            out_img[3 * pix + 0] = acc[k][0] + T[k] * bg0;
            out_img[3 * pix + 1] = acc[k][1] + T[k] * bg1;
            out_img[3 * pix + 2] = acc[k][2] + T[k] * bg2;
            final_Ts[pix] = T[k];
            final_idx[pix] = last[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
template <bool EXACT>
__global__ void __launch_bounds__(64)
k_rasterize_backward(int W, int H, int tiles_x, int num_tiles, const int32_t *__restrict__ ids,
                     const int2 *__restrict__ bins, const float4 *__restrict__ packed, float bg0,
                     float bg1, float bg2, const float *__restrict__ final_Ts,
                     const int32_t *__restrict__ final_idx, const float *__restrict__ v_out,
                     const float *__restrict__ v_out_alpha, float *__restrict__ v_xy,
                     float *__restrict__ v_conic, float *__restrict__ v_colors,
                     float *__restrict__ v_opacity) {
    __shared__ Staged stage[kChunk];
    __shared__ uint64_t exp_tab[32];
    const int lane = threadIdx.x;
    const int tile = xcd_swizzle(blockIdx.x, num_tiles);
    const int tile_x0 = (tile % tiles_x) * GS_TILE, tile_y0 = (tile / tiles_x) * GS_TILE;
    if (EXACT && lane < 32) exp_tab[lane] = kExp2fTab[lane];

    const int lx = lane & 15, ly = lane >> 4;
    const int px = tile_x0 + lx;
    const float pxf = (float)px;
    float pyf[4], T[4], Tfin[4], buf[4][3], vo[4][3], voa[4];
    int last[4];
    uint32_t probe[4];
    int my_last = -1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int py = tile_y0 + ly + 4 * k;
        pyf[k] = (float)py;
        probe[k] = (1u << lx) | (1u << (16 + ly + 4 * k));
        buf[k][0] = buf[k][1] = buf[k][2] = 0.0f;
        if (px < W && py < H) {
            const size_t pix = (size_t)py * W + px;
            Tfin[k] = final_Ts[pix];
            last[k] = final_idx[pix];
            vo[k][0] = v_out[3 * pix + 0];
            vo[k][1] = v_out[3 * pix + 1];
            vo[k][2] = v_out[3 * pix + 2];
            voa[k] = v_out_alpha ? v_out_alpha[pix] : 0.0f;
        } else {
            Tfin[k] = 1.0f;
            last[k] = -1;
            vo[k][0] = vo[k][1] = vo[k][2] = 0.0f;
            voa[k] = 0.0f;
        }
        T[k] = Tfin[k];
        my_last = max(my_last, last[k]);
    }
    const int2 range = bins[tile];
    const int wave_last = wave_max_i(my_last);  // last list entry any pixel of the tile used
    if (wave_last < range.x) return;            // (also covers empty tiles / no contributors)

    // walk the list back to front in chunks; slot 0 of a chunk is its furthest-back entry
    for (int hi = wave_last; hi >= range.x; hi -= kChunk) {
        __syncthreads();
        if (hi - lane >= range.x) stage_entry(&stage[lane], hi - lane, ids, packed, tile_x0, tile_y0);
        __syncthreads();
        const int n = min(kChunk, hi - range.x + 1);
        for (int t = 0; t < n; t++) {
            const float4 ea = stage[t].a;
            const float4 eb = stage[t].b;
            const uint32_t mask = __builtin_amdgcn_readfirstlane(__float_as_uint(eb.w));
            if (mask == 0u) continue;
            const int e = hi - t;  // index of this entry in the sorted list
            const float4 ec = stage[t].c;
            const float dx = ea.x - pxf;
            const float Adx = ea.z * dx, Bdx = ea.w * dx, Adxdx = Adx * dx;
            float g_x = 0.f, g_y = 0.f, g_A = 0.f, g_B = 0.f, g_C = 0.f;
            float g_r = 0.f, g_g = 0.f, g_b = 0.f, g_o = 0.f;
            bool any = false;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (((mask >> (16 + 4 * k)) & 0xFu) == 0u) continue;
                const float dy = ea.y - pyf[k];
                const float sigma = 0.5f * (Adxdx + eb.x * dy * dy) + Bdx * dy;
                const bool need = (e <= last[k]) && ((mask & probe[k]) == probe[k]) &&
                                  (sigma >= 0.0f) && (sigma <= eb.z);
                if (__ballot(need) == 0ull) continue;
                if (need) {
                    // gsplat_cpu.cpp:337-370
                    const float vis = gs_exp<EXACT>(-sigma, exp_tab);
                    const float alpha = fminf(0.99f, eb.y * vis);
                    if (alpha >= (1.0f / 255.0f)) {
                        const float ra = 1.0f / (1.0f - alpha);
                        T[k] *= ra;
                        const float fac = alpha * T[k];
                        g_r += fac * vo[k][0];
                        g_g += fac * vo[k][1];
                        g_b += fac * vo[k][2];
                        const float Tr = Tfin[k] * ra;
                        const float v_alpha =
                            ((ec.x * T[k] - buf[k][0] * ra) * vo[k][0]) +
                            ((ec.y * T[k] - buf[k][1] * ra) * vo[k][1]) +
                            ((ec.z * T[k] - buf[k][2] * ra) * vo[k][2]) + (Tr * voa[k]) +
                            (-Tfin[k] * ra * bg0 * vo[k][0]) + (-Tfin[k] * ra * bg1 * vo[k][1]) +
                            (-Tfin[k] * ra * bg2 * vo[k][2]);
                        buf[k][0] += ec.x * fac;
                        buf[k][1] += ec.y * fac;
                        buf[k][2] += ec.z * fac;
                        const float v_sigma = -eb.y * vis * v_alpha;
                        g_A += 0.5f * v_sigma * dx * dx;
                        g_B += 0.5f * v_sigma * dx * dy;
                        g_C += 0.5f * v_sigma * dy * dy;
                        g_x += v_sigma * (Adx + ea.w * dy);
                        g_y += v_sigma * (Bdx + eb.x * dy);
                        g_o += vis * v_alpha;
                        any = true;
                    }
                }
            }
            if (__ballot(any) == 0ull) continue;
            // wave-wide sums (uniform results), then one lane scatters 9 atomics
            const float s_x = wave_sum(g_x), s_y = wave_sum(g_y);
            const float s_A = wave_sum(g_A), s_B = wave_sum(g_B), s_C = wave_sum(g_C);
            const float s_r = wave_sum(g_r), s_g = wave_sum(g_g), s_b = wave_sum(g_b);
            const float s_o = wave_sum(g_o);
            if (lane == 0) {
                const int g = __float_as_int(ec.w);
                atomicAdd(&v_xy[2 * (size_t)g + 0], s_x);
                atomicAdd(&v_xy[2 * (size_t)g + 1], s_y);
                atomicAdd(&v_conic[3 * (size_t)g + 0], s_A);
                atomicAdd(&v_conic[3 * (size_t)g + 1], s_B);
                atomicAdd(&v_conic[3 * (size_t)g + 2], s_C);
                atomicAdd(&v_colors[3 * (size_t)g + 0], s_r);
                atomicAdd(&v_colors[3 * (size_t)g + 1], s_g);
                atomicAdd(&v_colors[3 * (size_t)g + 2], s_b);
                atomicAdd(&v_opacity[g], s_o);
            }
        }
    }
}

// Test hook: the exponential exactly as the compositing kernels evaluate it.
template <bool EXACT>
__global__ void __launch_bounds__(256) k_debug_expf(int64_t n, const float *__restrict__ x,
                                                    float *__restrict__ y) {
    __shared__ uint64_t exp_tab[32];
    if (threadIdx.x < 32) exp_tab[threadIdx.x] = kExp2fTab[threadIdx.x];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        y[i] = gs_exp<EXACT>(x[i], exp_tab);
}

}  // namespace gs

extern "C" int gs_debug_expf(int64_t n, const float *x, float *y, uint32_t flags,
                             gs_stream_t stream) {
    if (n < 0) return GS_ERR_INVALID_ARGUMENT;
    if (n == 0) return GS_OK;
    if (!x || !y) return GS_ERR_INVALID_ARGUMENT;
    int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (flags & GS_FLAG_FAST_EXP)
        hipLaunchKernelGGL(gs::k_debug_expf<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           n, x, y);
    else
        hipLaunchKernelGGL(gs::k_debug_expf<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           n, x, y);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" int gs_rasterize_forward(int W, int H, const int32_t *gaussian_ids_sorted,
                                    const int32_t *tile_bins, const float *packed,
                                    const float *background, float *out_img, float *final_Ts,
                                    int32_t *final_idx, uint32_t flags, gs_stream_t stream) {
    if (W <= 0 || H <= 0) return GS_ERR_INVALID_ARGUMENT;
    if (W > 65535 || H > 65535) return GS_ERR_UNSUPPORTED;
    if (!tile_bins || !background || !out_img || !final_Ts || !final_idx)
        return GS_ERR_INVALID_ARGUMENT;
    if ((uintptr_t)packed & 15u) return GS_ERR_INVALID_ARGUMENT;
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    const int tiles = tiles_x * tiles_y;
    hipStream_t s = (hipStream_t)stream;
    const int2 *bins = reinterpret_cast<const int2 *>(tile_bins);
    const float4 *pk = reinterpret_cast<const float4 *>(packed);
    if (flags & GS_FLAG_FAST_EXP)
        hipLaunchKernelGGL(gs::k_rasterize_forward<false>, dim3(tiles), dim3(64), 0, s, W, H,
                           tiles_x, tiles, gaussian_ids_sorted, bins, pk, background[0],
                           background[1], background[2], out_img, final_Ts, final_idx);
    else
        hipLaunchKernelGGL(gs::k_rasterize_forward<true>, dim3(tiles), dim3(64), 0, s, W, H,
                           tiles_x, tiles, gaussian_ids_sorted, bins, pk, background[0],
                           background[1], background[2], out_img, final_Ts, final_idx);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" int gs_rasterize_backward(int W, int H, const int32_t *gaussian_ids_sorted,
                                     const int32_t *tile_bins, const float *packed,
                                     const float *background, const float *final_Ts,
                                     const int32_t *final_idx, const float *v_out,
                                     const float *v_out_alpha, float *v_xy, float *v_conic,
                                     float *v_colors, float *v_opacity, uint32_t flags,
                                     gs_stream_t stream) {
    if (W <= 0 || H <= 0) return GS_ERR_INVALID_ARGUMENT;
    if (W > 65535 || H > 65535) return GS_ERR_UNSUPPORTED;
    if (!tile_bins || !background || !final_Ts || !final_idx || !v_out || !v_xy || !v_conic ||
        !v_colors || !v_opacity)
        return GS_ERR_INVALID_ARGUMENT;
    if ((uintptr_t)packed & 15u) return GS_ERR_INVALID_ARGUMENT;
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    const int tiles = tiles_x * tiles_y;
    hipStream_t s = (hipStream_t)stream;
    const int2 *bins = reinterpret_cast<const int2 *>(tile_bins);
    const float4 *pk = reinterpret_cast<const float4 *>(packed);
    if (flags & GS_FLAG_FAST_EXP)
        hipLaunchKernelGGL(gs::k_rasterize_backward<false>, dim3(tiles), dim3(64), 0, s, W, H,
                           tiles_x, tiles, gaussian_ids_sorted, bins, pk, background[0],
                           background[1], background[2], final_Ts, final_idx, v_out, v_out_alpha,
                           v_xy, v_conic, v_colors, v_opacity);
    else
        hipLaunchKernelGGL(gs::k_rasterize_backward<true>, dim3(tiles), dim3(64), 0, s, W, H,
                           tiles_x, tiles, gaussian_ids_sorted, bins, pk, background[0],
                           background[1], background[2], final_Ts, final_idx, v_out, v_out_alpha,
                           v_xy, v_conic, v_colors, v_opacity);
    GS_LAUNCH_CHECK();
    return GS_OK;
}
