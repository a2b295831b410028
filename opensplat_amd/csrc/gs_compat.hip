// gs_compat.hip — the two binning launchers of the REFERENCE's contract (global key list sorted by
// the caller with torch::sort), for callers that keep OpenSplat's rasterize_gaussians.cpp and only
// swap rasterizer/gsplat/bindings.cu (include/gsplat_compat.h).  Not on the native path, which
// bins without a global sort (gs_bin.hip); nothing here is tuned beyond coalesced accesses.
#include "gs_device.h"
#include "../../include/gsplat_compat.h"

namespace gs {

// tiles overlapped by the square of half-width `radius` around the centre, as index ranges
// [x0, x1) x [y0, y1) clipped to the grid (behaviour of helpers.cuh:17-49: truncating conversion of
// centre -+ radius in tile units, +1 on the upper side)
struct TileSpan {
    int x0, x1, y0, y1;
};
__device__ __forceinline__ TileSpan radius_span(float cx, float cy, int radius, int tiles_x,
                                                int tiles_y) {
    const float tcx = cx / (float)GS_TILE, tcy = cy / (float)GS_TILE;
    const float tr = (float)radius / (float)GS_TILE;
    TileSpan s;
    s.x0 = min(max(0, f2i_sat(tcx - tr)), tiles_x);
    s.x1 = min(max(0, f2i_sat(tcx + tr + 1.0f)), tiles_x);
    s.y0 = min(max(0, f2i_sat(tcy - tr)), tiles_y);
    s.y1 = min(max(0, f2i_sat(tcy + tr + 1.0f)), tiles_y);
    return s;
}

__global__ void __launch_bounds__(256)
k_compat_tiles_hit(int N, const float2 *__restrict__ xys, const int32_t *__restrict__ radii,
                   int tiles_x, int tiles_y, int32_t *__restrict__ num_tiles_hit) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    int cnt = 0;
    if (radii[n] > 0) {
        const float2 c = xys[n];
        const TileSpan s = radius_span(c.x, c.y, radii[n], tiles_x, tiles_y);
        cnt = max(s.x1 - s.x0, 0) * max(s.y1 - s.y0, 0);
    }
    num_tiles_hit[n] = cnt;
}

__global__ void __launch_bounds__(256)
k_compat_map_intersects(int N, const float2 *__restrict__ xys, const float *__restrict__ depths,
                        const int32_t *__restrict__ radii, const int32_t *__restrict__ cum_tiles_hit,
                        int tiles_x, int tiles_y, int64_t *__restrict__ isect_ids,
                        int32_t *__restrict__ gaussian_ids) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N || radii[n] <= 0) return;
    const float2 c = xys[n];
    const TileSpan s = radius_span(c.x, c.y, radii[n], tiles_x, tiles_y);
    int64_t k = n == 0 ? 0 : cum_tiles_hit[n - 1];
    const int64_t depth_bits = (int64_t)__float_as_int(depths[n]);  // sign-extended, like the reference
    for (int ty = s.y0; ty < s.y1; ty++)
        for (int tx = s.x0; tx < s.x1; tx++) {
            const int64_t tile = (int64_t)ty * tiles_x + tx;
            isect_ids[k] = (tile << 32) | depth_bits;
            gaussian_ids[k] = n;
            k++;
        }
}

__global__ void __launch_bounds__(256)
k_compat_tile_bin_edges(int64_t M, int64_t rows, const int64_t *__restrict__ ids,
                        int2 *__restrict__ bins) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    // a tile id outside the table is never written (the reference's kernel writes it unchecked,
    // forward.cu:148-176, into a table of num_intersects rows, bindings.cu:324-326)
    const int64_t tile = ids[i] >> 32;
    const bool in = tile >= 0 && tile < rows;
    if (i == 0 && in) bins[tile].x = 0;
    if (i == M - 1 && in) bins[tile].y = (int)M;
    if (i > 0) {
        const int64_t prev = ids[i - 1] >> 32;
        if (prev != tile) {
            if (prev >= 0 && prev < rows) bins[prev].y = (int)i;
            if (in) bins[tile].x = (int)i;
        }
    }
}

}  // namespace gs

extern "C" int gs_compat_tiles_hit(int N, const float *xys, const int32_t *radii, int tiles_x,
                                   int tiles_y, int32_t *num_tiles_hit, gs_stream_t stream) {
    if (N < 0 || tiles_x <= 0 || tiles_y <= 0) return GS_ERR_INVALID_ARGUMENT;
    if (N == 0) return GS_OK;
    if (!xys || !radii || !num_tiles_hit) return GS_ERR_INVALID_ARGUMENT;
    GS_LAUNCH(gs::k_compat_tiles_hit, dim3((N + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, N, reinterpret_cast<const float2 *>(xys), radii, tiles_x,
                       tiles_y, num_tiles_hit);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" int gs_compat_map_intersects(int N, const float *xys, const float *depths,
                                        const int32_t *radii, const int32_t *cum_tiles_hit,
                                        int tiles_x, int tiles_y, int64_t *isect_ids,
                                        int32_t *gaussian_ids, gs_stream_t stream) {
    if (N < 0 || tiles_x <= 0 || tiles_y <= 0) return GS_ERR_INVALID_ARGUMENT;
    if (N == 0) return GS_OK;
    if (!xys || !depths || !radii || !cum_tiles_hit || !isect_ids || !gaussian_ids)
        return GS_ERR_INVALID_ARGUMENT;
    GS_LAUNCH(gs::k_compat_map_intersects, dim3((N + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, N, reinterpret_cast<const float2 *>(xys), depths, radii,
                       cum_tiles_hit, tiles_x, tiles_y, isect_ids, gaussian_ids);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" int gs_compat_tile_bin_edges(int64_t num_intersects, const int64_t *isect_ids_sorted,
                                        int32_t *tile_bins, int64_t tile_bins_rows,
                                        gs_stream_t stream) {
    if (num_intersects < 0 || tile_bins_rows < 0) return GS_ERR_INVALID_ARGUMENT;
    if (num_intersects == 0) return GS_OK;
    if (!isect_ids_sorted || !tile_bins) return GS_ERR_INVALID_ARGUMENT;
    GS_LAUNCH(gs::k_compat_tile_bin_edges, dim3((unsigned)((num_intersects + 255) / 256)),
                       dim3(256), 0, (hipStream_t)stream, num_intersects, tile_bins_rows,
                       isect_ids_sorted, reinterpret_cast<int2 *>(tile_bins));
    GS_LAUNCH_CHECK();
    return GS_OK;
}
