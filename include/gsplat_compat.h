/* gsplat_compat.h — LAUNCHER-LEVEL compatibility entry points of libgsplat_hip.so.
 *
 * The native path (include/gsplat_hip.h) bins with a counting partition and per-tile sorts and never
 * materialises the reference's global (tile << 32 | depth) key list.  A caller that keeps
 * OpenSplat's own operator files (rasterize_gaussians.cpp:6-37: cumsum -> map_gaussian_to_intersects
 * -> torch::sort -> gather -> get_tile_bin_edges) needs the two launchers of that contract; they are
 * provided here, together with the tile count by the GPU reference's radius-square rule that the
 * contract is built on (rasterizer/gsplat/helpers.cuh:17-49, forward.cu:86-94).  Used only by
 * opensplat_amd/csrc/bindings_hip_native.cpp (the eight *_tensor functions of
 * rasterizer/gsplat/bindings.h); the native operators never call them.
 *
 * Conventions as in gsplat_hip.h: device pointers, caller-owned outputs, work enqueued on `stream`,
 * status codes, no allocation, no synchronisation.
 */
#ifndef GSPLAT_COMPAT_H
#define GSPLAT_COMPAT_H

#include "gsplat_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* num_tiles_hit[n] = tiles of the square of half-width radii[n] around xys[n] (0 for radius <= 0),
 * in a grid of tiles_x x tiles_y 16x16 tiles — what project_gaussians_forward_tensor returns as
 * its sixth output (forward.cu:86-94). */
int gs_compat_tiles_hit(int N, const float *xys, const int32_t *radii, int tiles_x, int tiles_y,
                        int32_t *num_tiles_hit, gs_stream_t stream);

/* Replaces map_gaussian_to_intersects_tensor (bindings.h:96-104, forward.cu:107-143): for Gaussian
 * n the tiles of its radius square, written from cum_tiles_hit[n-1] on:
 *   isect_ids[k]    = (int64) tile_id << 32 | bits of depths[n]
 *   gaussian_ids[k] = n                                                                        */
int gs_compat_map_intersects(int N, const float *xys, const float *depths, const int32_t *radii,
                             const int32_t *cum_tiles_hit, int tiles_x, int tiles_y,
                             int64_t *isect_ids, int32_t *gaussian_ids, gs_stream_t stream);

/* Replaces get_tile_bin_edges_tensor (bindings.h:106-109, forward.cu:148-176): [start, end) of
 * every tile's run in the sorted key list.  tile_bins [tile_bins_rows, 2] int32 must be zeroed by
 * the caller; tile ids >= tile_bins_rows are skipped (the reference writes them unchecked). */
int gs_compat_tile_bin_edges(int64_t num_intersects, const int64_t *isect_ids_sorted,
                             int32_t *tile_bins, int64_t tile_bins_rows, gs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_COMPAT_H */
