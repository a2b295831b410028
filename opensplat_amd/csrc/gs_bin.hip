// gs_bin.hip — per-Gaussian 2-D record packing, tile counting, intersection emission, radix sort
// and per-tile range extraction.
//
// Replaces, on the reference side: torch::cumsum + .item() (rasterize_gaussians.cpp:62-63),
// map_gaussian_to_intersects (forward.cu:107-143), torch::sort + torch::gather
// (rasterize_gaussians.cpp:25-32) and get_tile_bin_edges (forward.cu:148-169).
//
// Differences that are deliberate:
//   * which tiles a Gaussian lands in is decided by the CPU oracle's pixel rectangle
//     (gsplat_cpu.cpp:167-168,201-204), not by the GPU reference's radius square — that is what
//     makes the contributor sets equal to gsplat-cpu's;
//   * the sort key uses only the bits that vary: 32 depth bits + ceil(log2(tiles)) tile bits, and
//     the sort moves 4-byte Gaussian ids, not 8-byte argsort indices followed by a gather;
//   * tile_bins is [tiles, 2] (the reference allocates [M, 2], bindings.cu:324-326).
//
// All kernels are HBM/atomic-free streaming passes; the sort itself is rocPRIM's device radix
// sort (a HIP-native header library compiled here for gfx950).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "gs_device.h"

namespace gs {

// ---- pack ------------------------------------------------------------------------------------
// One lane per Gaussian; reads 52 B (+12 B cov2d), writes 48 B + 4 B.
//
// Record: { x, y, A, B | C, opacity, sigma_max, x0|x1<<16 | r, g, b, y0|y1<<16 }.
//
// sigma_max = ln(255*opacity) + margin: a pixel can only reach alpha = opacity*exp(-sigma) >= 1/255
// (gsplat_cpu.cpp:220-222) if sigma <= ln(255*opacity).  The rectangle stored is the CPU oracle's
// pixel rectangle (gsplat_cpu.cpp:167-168,201-204) INTERSECTED with the bounding box of that
// sigma <= sigma_max ellipse, |dx| <= sqrt(2*sigma_max*C/det), |dy| <= sqrt(2*sigma_max*A/det)
// (det = AC - B^2 of the conic the compositing kernels evaluate).  Pixels dropped by the
// intersection fail the alpha threshold in the reference too, so contributor sets are unchanged
// while tiles per Gaussian (and the sort length M) shrink by ~1.5x on the benchmark scenes.
// The lowest mantissa bit of sigma_max is a flag: 1 = the CPU rectangle cuts into the ellipse box
// (or the box could not be trusted), so the compositing kernels must apply the per-pixel
// rectangle test; 0 = the ellipse test alone is exact.
__global__ void __launch_bounds__(256)
k_pack_splats(int W, int H, int N, const float *__restrict__ xys,
              const int32_t *__restrict__ radii, const float *__restrict__ conics,
              const float *__restrict__ colors, const float *__restrict__ opacities,
              const float *__restrict__ cov2d, float4 *__restrict__ packed,
              int32_t *__restrict__ tiles_hit) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float x = xys[2 * n], y = xys[2 * n + 1];
    float A = conics[3 * n], B = conics[3 * n + 1], C = conics[3 * n + 2];
    const float det = A * C - B * B;
    float cxx, cyy;
    if (cov2d) {
        cxx = cov2d[3 * n];
        cyy = cov2d[3 * n + 2];
    } else {
        // conic = cov2d^-1  ->  cov2d = conic^-1: xx = C / det, yy = A / det
        cxx = C / det;
        cyy = A / det;
    }
    PixRect r = pixel_rect(x, y, cxx, cyy, W, H);
    const float opac = opacities[n];
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
    int tiles = (radii[n] > 0 && smax >= 0.0f) ? rect_tiles(r) : 0;
    if (tiles == 0) r.x0 = r.x1 = r.y0 = r.y1 = 0;
    uint32_t rx = (uint32_t)r.x0 | ((uint32_t)r.x1 << 16);
    uint32_t ry = (uint32_t)r.y0 | ((uint32_t)r.y1 << 16);
    smax = __uint_as_float((__float_as_uint(smax) & ~1u) | binding);
    packed[3 * n + 0] = make_float4(x, y, A, B);
    packed[3 * n + 1] = make_float4(C, opac, smax, __uint_as_float(rx));
    packed[3 * n + 2] =
        make_float4(colors[3 * n], colors[3 * n + 1], colors[3 * n + 2], __uint_as_float(ry));
    tiles_hit[n] = tiles;
}

// ---- emit ------------------------------------------------------------------------------------
// One lane per Gaussian writes its (tile | depth) keys and ids at cum[n-1] .. cum[n].
// TODO(perf): a lane-per-intersection mapping (binary search in cum) would coalesce the writes.
__global__ void __launch_bounds__(256)
k_emit_isects(int N, int tiles_x, const float4 *__restrict__ packed,
              const float *__restrict__ depths, const int32_t *__restrict__ cum,
              int64_t *__restrict__ keys, int32_t *__restrict__ ids) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    int start = (n == 0) ? 0 : cum[n - 1];
    int end = cum[n];
    if (end <= start) return;
    uint32_t rx = __float_as_uint(packed[3 * n + 1].w), ry = __float_as_uint(packed[3 * n + 2].w);
    int x0 = rx & 0xFFFF, x1 = rx >> 16, y0 = ry & 0xFFFF, y1 = ry >> 16;
    int tx0 = x0 / GS_TILE, tx1 = (x1 + GS_TILE - 1) / GS_TILE;
    int ty0 = y0 / GS_TILE, ty1 = (y1 + GS_TILE - 1) / GS_TILE;
    // Order-preserving float -> uint map (flip all bits of negatives, set the sign bit of
    // non-negatives).  The reference uses the raw bit pattern, valid only for depth > 0
    // (forward.cu:132); the map sorts identically there and stays correct for ANY key, which
    // lets tests drive the sort with the CPU reference's as-read keys.
    uint32_t db = __float_as_uint(depths[n]);
    db = (db & 0x80000000u) ? ~db : (db | 0x80000000u);
    uint64_t depth_bits = (uint64_t)db;
    int o = start;
    for (int ty = ty0; ty < ty1; ty++)
        for (int tx = tx0; tx < tx1; tx++) {
            uint64_t tile = (uint64_t)(ty * tiles_x + tx);
            keys[o] = (int64_t)((tile << 32) | depth_bits);
            ids[o] = n;
            o++;
        }
}

// ---- tile ranges -----------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_tile_bin_edges(int M, const int64_t *__restrict__ keys_sorted, int2 *__restrict__ bins) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    int cur = (int)(keys_sorted[i] >> 32);
    if (i == 0) bins[cur].x = 0;
    else {
        int prev = (int)(keys_sorted[i - 1] >> 32);
        if (prev != cur) {
            bins[prev].y = i;
            bins[cur].x = i;
        }
    }
    if (i == M - 1) bins[cur].y = M;
}

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

static int tile_bits(int W, int H) {
    int tiles = ((W + GS_TILE - 1) / GS_TILE) * ((H + GS_TILE - 1) / GS_TILE);
    int b = 0;
    while ((1 << b) < tiles) b++;
    return b < 1 ? 1 : b;
}

static size_t scan_temp_bytes(int N) {
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (const int32_t *)nullptr, (int32_t *)nullptr,
                            (size_t)N, rocprim::plus<int32_t>(), (hipStream_t)0);
    return bytes;
}

static size_t sort_temp_bytes(int64_t M, int end_bit) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                              (const int32_t *)nullptr, (int32_t *)nullptr, (size_t)M, 0u,
                              (unsigned)end_bit, (hipStream_t)0);
    return bytes;
}

}  // namespace gs

extern "C" int gs_pack_splats(int W, int H, int N, const float *xys, const float *depths,
                              const int32_t *radii, const float *conics, const float *colors,
                              const float *opacities, const float *cov2d, float *packed,
                              int32_t *tiles_hit, gs_stream_t stream) {
    if (N < 0 || W <= 0 || H <= 0) return GS_ERR_INVALID_ARGUMENT;
    if (W > 65535 || H > 65535) return GS_ERR_UNSUPPORTED;
    if (N == 0) return GS_OK;
    if (!xys || !depths || !radii || !conics || !colors || !opacities || !packed || !tiles_hit)
        return GS_ERR_INVALID_ARGUMENT;
    if ((uintptr_t)packed & 15u) return GS_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(gs::k_pack_splats, dim3((N + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, W, H, N, xys, radii, conics, colors, opacities,
                       cov2d, reinterpret_cast<float4 *>(packed), tiles_hit);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" size_t gs_bin_workspace_bytes(int N, int64_t num_isects, int W, int H) {
    if (N < 0 || num_isects < 0 || W <= 0 || H <= 0) return 0;
    int64_t M = num_isects > 0 ? num_isects : 1;
    size_t scan = gs::scan_temp_bytes(N > 0 ? N : 1);
    size_t sort = gs::sort_temp_bytes(M, 32 + gs::tile_bits(W, H));
    size_t tmp = gs::align_up(scan > sort ? scan : sort);
    // three optional arrays (unsorted keys, unsorted ids, sorted keys) + library temp storage
    return tmp + gs::align_up((size_t)M * 8) * 2 + gs::align_up((size_t)M * 4) + 256;
}

extern "C" int gs_bin_scan(int N, const int32_t *tiles_hit, int32_t *cum_tiles_hit,
                           int32_t *num_isects_host, void *workspace, size_t workspace_bytes,
                           gs_stream_t stream) {
    if (N < 0) return GS_ERR_INVALID_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        if (num_isects_host) *num_isects_host = 0;
        return GS_OK;
    }
    if (!tiles_hit || !cum_tiles_hit || !workspace) return GS_ERR_INVALID_ARGUMENT;
    size_t need = gs::scan_temp_bytes(N);
    if (workspace_bytes < need) return GS_ERR_WORKSPACE;
    GS_HIP_CHECK(rocprim::inclusive_scan(workspace, need, tiles_hit, cum_tiles_hit, (size_t)N,
                                         rocprim::plus<int32_t>(), s));
    if (num_isects_host)
        GS_HIP_CHECK(hipMemcpyAsync(num_isects_host, cum_tiles_hit + (N - 1), sizeof(int32_t),
                                    hipMemcpyDeviceToHost, s));
    return GS_OK;
}

extern "C" int gs_bin_sort(int W, int H, int N, int32_t num_isects, const float *packed,
                           const float *depths, const int32_t *cum_tiles_hit, int64_t *isect_ids,
                           int32_t *gaussian_ids, int64_t *isect_ids_sorted,
                           int32_t *gaussian_ids_sorted, int32_t *tile_bins, void *workspace,
                           size_t workspace_bytes, gs_stream_t stream) {
    if (N < 0 || num_isects < 0 || W <= 0 || H <= 0) return GS_ERR_INVALID_ARGUMENT;
    if (W > 65535 || H > 65535) return GS_ERR_UNSUPPORTED;
    if (!tile_bins) return GS_ERR_INVALID_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    const int tiles = tiles_x * tiles_y;
    GS_HIP_CHECK(hipMemsetAsync(tile_bins, 0, sizeof(int32_t) * 2 * (size_t)tiles, s));
    const int64_t M = num_isects;
    if (N == 0 || M == 0) return GS_OK;
    if (!packed || !depths || !cum_tiles_hit || !gaussian_ids_sorted || !workspace)
        return GS_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < gs_bin_workspace_bytes(N, M, W, H)) return GS_ERR_WORKSPACE;

    const int end_bit = 32 + gs::tile_bits(W, H);
    size_t sort_tmp = gs::sort_temp_bytes(M, end_bit);
    size_t scan_tmp = gs::scan_temp_bytes(N > 0 ? N : 1);
    size_t tmp = gs::align_up(sort_tmp > scan_tmp ? sort_tmp : scan_tmp);
    char *base = static_cast<char *>(workspace);
    char *p = base + tmp;
    if (!isect_ids) isect_ids = reinterpret_cast<int64_t *>(p);
    p += gs::align_up((size_t)M * 8);
    if (!isect_ids_sorted) isect_ids_sorted = reinterpret_cast<int64_t *>(p);
    p += gs::align_up((size_t)M * 8);
    if (!gaussian_ids) gaussian_ids = reinterpret_cast<int32_t *>(p);

    hipLaunchKernelGGL(gs::k_emit_isects, dim3((N + 255) / 256), dim3(256), 0, s, N, tiles_x,
                       reinterpret_cast<const float4 *>(packed), depths, cum_tiles_hit, isect_ids,
                       gaussian_ids);
    GS_LAUNCH_CHECK();
    GS_HIP_CHECK(rocprim::radix_sort_pairs(
        base, sort_tmp, reinterpret_cast<const uint64_t *>(isect_ids),
        reinterpret_cast<uint64_t *>(isect_ids_sorted), gaussian_ids, gaussian_ids_sorted,
        (size_t)M, 0u, (unsigned)end_bit, s));
    hipLaunchKernelGGL(gs::k_tile_bin_edges, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s,
                       (int)M, isect_ids_sorted, reinterpret_cast<int2 *>(tile_bins));
    GS_LAUNCH_CHECK();
    return GS_OK;
}
