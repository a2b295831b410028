/*
 * gsplat_hip.h — C ABI of the MI355X (gfx950) Gaussian-splat rasterizer, `libgsplat_hip.so`.
 *
 * This is the drop-in boundary for OpenSplat's hot path: every entry point below replaces one of
 * the `*_tensor` launchers that OpenSplat's three autograd operators call
 * (reference: rasterizer/gsplat/bindings.h, cited per function).  The signatures carry plain
 * device pointers, sizes and a stream — no torch types — so the same library is bindable from
 * C++/libtorch (opensplat_amd/csrc/torch_ops.cpp does exactly that), ctypes, cgo or JNI.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major data unless marked "host";
 *   - fp32 for all real data, int32 for ids / radii / counts, as in the reference
 *     (bindings.h:42-64, 111-126);
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); nothing synchronises
 *     except where stated; nothing is allocated — the caller owns outputs and workspace;
 *   - return value: GS_OK (0) or a negative GsStatus; functions never throw, never exit;
 *   - re-entrant and stateless: safe to call from libtorch's autograd thread.
 *
 * Numerics follow OpenSplat's CPU implementation (rasterizer/gsplat-cpu/gsplat_cpu.cpp), which is
 * the parity oracle: pixel-rectangle culling (gsplat_cpu.cpp:167-168,201-204), alpha clamps
 * 0.999 forward / 0.99 backward (:220,:338), alpha threshold 1/255, transmittance stop 1e-4
 * (:225-228), IEEE expf bit-compatible with glibc.  See DESIGN.md.
 */
#ifndef GSPLAT_HIP_H
#define GSPLAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_TILE 16          /* BLOCK_X == BLOCK_Y == 16, rasterizer/gsplat/config.h:1-2 */
#define GS_SPLAT_DWORDS 12  /* packed per-Gaussian 2-D record, see gs_pack_splats */

typedef void *gs_stream_t; /* hipStream_t */

typedef enum GsStatus {
    GS_OK = 0,
    GS_ERR_INVALID_ARGUMENT = -1, /* null pointer, negative size, K not in {1,4,9,16,25} ... */
    GS_ERR_UNSUPPORTED = -2,      /* image wider/taller than 65535 px */
    GS_ERR_WORKSPACE = -3,        /* workspace too small for (N, max_isects) */
    GS_ERR_HIP = -4,              /* a HIP runtime call failed; see gs_last_hip_error() */
    GS_ERR_CAPACITY = -5          /* more intersections than the caller's id buffer holds */
} GsStatus;

/* Camera + image description: the scalar arguments of ProjectGaussians::apply
 * (project_gaussians.hpp:12-30) as one host-side POD.  Matrices are row-major 4x4. */
typedef struct GsCamera {
    float viewmat[16]; /* world -> camera                                  */
    float projmat[16]; /* full projection (proj @ view), model.cpp:152      */
    float fx, fy, cx, cy;
    int32_t img_width, img_height;
    float clip_thresh; /* near-plane cull, default 0.01 (project_gaussians.hpp:28) */
    float glob_scale;  /* 1.0 everywhere in OpenSplat                        */
    uint32_t flags;    /* GS_CAM_* ; 0 = the reference operator's semantics    */
} GsCamera;

/* GsCamera.flags.  "Fused glue" (SURVEY.md §8 row f1): lets the kernels absorb the element-wise
 * torch ops Model::forward wraps around the three operators (model.cpp:114-225). */
#define GS_CAM_LOG_SCALES 1u /* `scales` holds log-scales: exp() applied inside, v_scales is the  \
                                gradient w.r.t. the log-scales (model.cpp:125,148 torch::exp)    */

/* Flags for the compositing kernels. */
#define GS_FLAG_FAST_EXP 1u /* use the hardware v_exp_f32 path instead of the glibc-bit-exact expf; \
                               contributor sets may then differ from the CPU oracle at thresholds */
/* fused glue (row f1), all optional: */
#define GS_FLAG_LOGIT_OPACITY 2u /* gs_pack_splats: `opacities` are logits, sigmoid applied inside   \
                                    (model.cpp:200,215); gs_rasterize_backward: v_opacity is then  \
                                    the gradient w.r.t. the logit                                  */
#define GS_FLAG_CLAMP_IMAGE 4u   /* gs_rasterize_forward also writes min(image, 1) (model.cpp:222);  \
                                    gs_rasterize_backward masks v_out where the raw image > 1     */

#define GS_FLAG_KEEP_RECORDS 8u    /* gs_rasterize_backward: leave the gradients in the 64-byte       \
                                     records of its workspace (v_xy / v_conic / v_colors / v_opacity \
                                     may be NULL); gs_gaussian_backward consumes them               */
#define GS_FLAG_RECORDS_ZEROED 16u /* gs_rasterize_backward: the record workspace is already zero     \
                                     (gs_gaussian_backward leaves it so): skip the memset           */
#define GS_FLAG_ACCUMULATE_GRADS 32u /* gs_gaussian_backward: ADD the six parameter gradients to the  \
                                     output tensors instead of overwriting them (several cameras   \
                                     per optimiser step on one rank, one gradient exchange)         */
#define GS_FLAG_DETERMINISTIC 64u  /* gs_rasterize_backward: order-independent gradient sums (debug): \
                                     partial sums are accumulated as 64-bit fixed point, so two     \
                                     runs are bit-identical; workspace from                         \
                                     gs_rasterize_backward_workspace_bytes_det                      */
#define GS_FLAG_EMIT_VCOLOR 128u   /* gs_gaussian_backward: write the colour cotangent behind the clamp \
                                     mask (3 floats per Gaussian) to v_dc and leave v_rest alone:    \
                                     the factored gradient exchange forms the SH gradients of ALL    \
                                     cameras with gs_sh_backward_cameras                             */

const char *gs_strerror(int status);
const char *gs_last_hip_error(void); /* thread-local text of the last failing HIP call */
/* ABI version of THIS header: 10000*major + 100*minor + patch.  Bumped whenever an entry point changes its
 * argument list or an entry point is added (0.3.0: block_masks / tile_bins_rows arguments of round 3; 0.4.0:
 * round 4; 0.4.2 / 0.4.3: gs_bin_strips, gs_bin_speculative, 0.4.4: gs_bin_speculative_zero, round 6).  A consumer
 * compiled against another header must refuse the library instead of calling through shifted arguments:
 * opensplat_amd/cabi.py compares gs_version() with this constant when it loads the library, libgsplat_torch.so in
 * front of its first call into it (torch_ops.cpp: current_stream()). */
#define GS_ABI_VERSION 404
int gs_version(void);                /* == GS_ABI_VERSION of the header the library was built from */

/* ---------------------------------------------------------------------------------------------
 * Projection.  Replaces project_gaussians_forward_tensor (rasterizer/gsplat/bindings.h:42-64,
 * kernel forward.cu:19-103); arithmetic follows the CPU oracle gsplat_cpu.cpp:48-131.
 *   in : means[N,3] scales[N,3] (already exp'd) quats[N,4] (w,x,y,z; normalised inside)
 *   out: xys[N,2] depths[N] (view-space z) radii[N] conics[N,3] num_tiles_hit[N] cov3d[N,6]
 *        cov2d[N,3] = (xx, xy, yy) incl. the +0.3 blur — what the CPU path returns as cov2d
 *        (gsplat_cpu.cpp:95-99) and what defines its per-Gaussian pixel rectangle.
 * viewmat_dev / projmat_dev: optional DEVICE copies of the two 4x4 matrices (row-major, 16
 * floats); when non-NULL they override cam->viewmat / cam->projmat and are read by the kernel
 * itself, so a caller holding them as device tensors (model.cpp:93-113,152) needs no
 * device->host copy.
 * Gaussians with view z <= clip_thresh get radius 0 / zero tiles (forward.cu:49-52).
 * num_tiles_hit counts 16x16 tiles overlapped by the CPU pixel rectangle.            */
int gs_project_forward(const GsCamera *cam /*host*/, const float *viewmat_dev /*nullable*/,
                       const float *projmat_dev /*nullable*/, int N, const float *means,
                       const float *scales, const float *quats, float *xys, float *depths,
                       int32_t *radii, float *conics, int32_t *num_tiles_hit, float *cov3d,
                       float *cov2d, gs_stream_t stream);

/* Replaces project_gaussians_backward_tensor (bindings.h:66-93, kernel backward.cu:357-421).
 * The VJP equals libtorch autograd through gsplat_cpu.cpp:48-131 (FOV clamp and glob_scale
 * included).  v_depth may be NULL (treated as zeros).  Rows with radii <= 0 get zero gradients.
 *   out: v_means[N,3] v_scales[N,3] v_quats[N,4] (fully written, no pre-zeroing needed)  */
int gs_project_backward(const GsCamera *cam /*host*/, const float *viewmat_dev /*nullable*/,
                        const float *projmat_dev /*nullable*/, int N, const float *means,
                        const float *scales, const float *quats, const int32_t *radii,
                        const float *v_xy, const float *v_depth, const float *v_conic,
                        float *v_means, float *v_scales, float *v_quats, gs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Spherical harmonics.  Replace compute_sh_forward_tensor / compute_sh_backward_tensor
 * (bindings.h:26-40, kernels sh.cuh:218-260); basis as gsplat_cpu.cpp:424-486.
 * K = number of stored bases (1,4,9,16,25); degrees_to_use <= degree(K).
 *   fwd: dirs[N,3] (unit) coeffs[N,K,3] -> colors[N,3]
 *   bwd: dirs[N,3] v_colors[N,3] -> v_coeffs[N,K,3] (bases above degrees_to_use written as 0) */
int gs_sh_forward(int N, int K, int degrees_to_use, const float *dirs, const float *coeffs,
                  float *colors, gs_stream_t stream);
int gs_sh_backward(int N, int K, int degrees_to_use, const float *dirs, const float *v_colors,
                   float *v_coeffs, gs_stream_t stream);

/* Fused-glue variants (row f1) of the two calls above, replacing torch::cat (model.cpp:114), the
 * view-direction computation (model.cpp:176-177) and clamp_min(rgb + 0.5, 0) (model.cpp:192):
 *   fwd: means[N,3], cam_pos (float[3] in host or device memory: camera centre in world space),
 *        features_dc[N,3],
 *        features_rest[N,K-1,3] (NULL iff K == 1) -> colors[N,3] = max(SH(dir) + 0.5, 0) and
 *        rgb_raw[N,3] = SH(dir), dir = normalize(means - cam_pos)
 *   bwd: v_colors[N,3] (gradient w.r.t. the clamped colours) + rgb_raw -> v_dc[N,3],
 *        v_rest[N,K-1,3]; no gradient to the means (the reference detaches them, model.cpp:176). */
int gs_sh_forward_fused(int N, int K, int degrees_to_use, const float *means,
                        const float *cam_pos /*host or device [3]*/, const float *features_dc,
                        const float *features_rest, float *colors, float *rgb_raw,
                        gs_stream_t stream);
int gs_sh_backward_fused(int N, int K, int degrees_to_use, const float *means,
                         const float *cam_pos /*host or device [3]*/, const float *rgb_raw,
                         const float *v_colors, float *v_dc, float *v_rest, gs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Tile binning + sort.  Together these replace cumsum + map_gaussian_to_intersects_tensor +
 * torch::sort/gather + get_tile_bin_edges_tensor (bindings.h:96-109,
 * rasterize_gaussians.cpp:6-37,62-63).
 *
 * gs_pack_splats: gathers the 2-D attributes of each Gaussian into one 48-byte record
 *   { x, y, conic A, B | C, opacity, sigma_max, (x0 | x1<<16) | r, g, b, (y0 | y1<<16) }
 * sigma_max = ln(255*opacity) + margin is the largest exponent at which the Gaussian can still
 * reach alpha >= 1/255 (gsplat_cpu.cpp:220-222); its lowest mantissa bit flags "the rectangle
 * must be tested per pixel".  [x0,x1) x [y0,y1) is the CPU oracle's pixel rectangle clipped to
 * the image (gsplat_cpu.cpp:167-168,201-204) intersected with the bounding box of the
 * sigma <= sigma_max ellipse — pixels outside that box fail the alpha threshold in the reference
 * as well, so contributor sets are unchanged while fewer tiles are touched.
 * tiles_hit[N] = 16x16 tiles overlapped by the stored rectangle.
 * cov2d may be NULL: the CPU rectangle is then derived from the conic's inverse (legacy call
 * sites that only have the reference's six projection outputs).  radii <= 0 or
 * opacity < 1/255 -> empty rectangle, zero tiles.                                            */
int gs_pack_splats(int W, int H, int N, const float *xys, const int32_t *radii,
                   const float *conics, const float *colors, const float *opacities,
                   const float *cov2d /*nullable*/, float *packed, int32_t *tiles_hit,
                   uint32_t flags /* GS_FLAG_LOGIT_OPACITY */, gs_stream_t stream);

/* Workspace (bytes, 256-byte aligned base) sufficient for gs_bin_scan (any num_isects) and for
 * gs_bin_sort with capacity num_isects, for N Gaussians and a W x H image.  gs_bin_scan leaves
 * per-workgroup segment offsets in it (where each persistent workgroup's entries start inside a tile's
 * segment) that the FOLLOWING gs_bin_sort continues from: give both calls the same workspace (same base pointer; it may be
 * larger for the second call), the same N, W and H, and do not touch it in between.  gs_bin_sort
 * returns GS_ERR_WORKSPACE for a workspace address no gs_bin_scan of the same (N, W, H) has been
 * given (the library remembers the last scan of each workspace address on the host). */
size_t gs_bin_workspace_bytes(int N, int64_t num_isects, int W, int H);

/* Counts the intersections of every 16x16 tile (rectangles of the packed records) and scans the
 * counts:  tile_bins[tiles,2] = [start, end) of each tile's segment in the sorted id list.
 * {M, longest tile list} are also stored (by the scan kernel itself, in stream order) to
 * num_isects_host[0..1], which must be PINNED, device-mapped host memory of two int32
 * (hipHostMalloc / a torch pinned tensor) or NULL; the caller synchronises the stream before reading it (the reference syncs at
 * the same place, rasterize_gaussians.cpp:63) — or does not read it at all and passes a
 * sufficient capacity to gs_bin_sort. */
int gs_bin_scan(int W, int H, int N, const float *packed, int32_t *tile_bins,
                int32_t *tile_order /*[tiles], nullable: tiles by descending list length*/,
                int32_t *num_isects_host /*pinned host int32[2], nullable*/, void *workspace,
                size_t workspace_bytes, gs_stream_t stream);

/* gs_bin_scan also leaves M on the DEVICE: one int32 at this byte offset inside its workspace (valid until the
 * workspace is reused).  Kernels enqueued behind a speculative gs_bin_sort can compare it with the capacity
 * they were given without the host in between — gs_adam_step_scheduled's guard (gsplat_train.h) does, so that
 * a captured training iteration whose id list turned out too small changes nothing. */
size_t gs_bin_num_isects_offset(int W, int H);

/* Fills every tile's segment of gaussian_ids_sorted[capacity] with the ids of the Gaussians
 * overlapping the tile, ordered by `depths` (any finite float key sorts correctly; ties in
 * Gaussian-index order) — the result of the reference's global (tile | depth) sort + gather,
 * built with a counting scatter and one on-chip sort per tile.
 * capacity should be >= M (the total of gs_bin_scan).  It may be a guess made WITHOUT reading M
 * (no host synchronisation between gs_bin_scan and gs_bin_sort): slots beyond capacity are never
 * written and tile_bins is clamped to capacity, so that the compositing kernels stay inside the
 * buffer; the caller compares the true M (num_isects_host, valid once the stream has been
 * synchronised — e.g. after the forward kernel has been enqueued) with capacity and repeats
 * scan + sort + compositing with a larger buffer if it was exceeded. */
int gs_bin_sort(int W, int H, int N, int32_t capacity, const float *packed, const float *depths,
                int32_t *tile_bins, int32_t *gaussian_ids_sorted,
                uint16_t *block_masks /*[capacity], see gs_block_masks*/,
                const int32_t *list_stats /*host int32[2] {M, longest list} of an earlier frame,
                                            nullable: lets the launch skip empty size classes*/,
                void *workspace, size_t workspace_bytes, gs_stream_t stream);

/* gs_bin_scan + gs_bin_sort in ONE call for callers that never look at M in between (the speculative id-list
 * capacity of gs_bin_sort): same outputs, same workspace, same contract — but the scan is no launch of its own any
 * more: every workgroup of the scatter kernel scans the tile counters itself and one extra workgroup writes
 * tile_bins, {M, longest list} (to num_isects_host, pinned, nullable, and M to the device word at
 * gs_bin_num_isects_offset) and tile_order beside the scatter (round 6: -9 us at 1080p, -37 us at 4K).  Images whose
 * tile counters do not fit in LDS (beyond 36 864 tiles) take the two calls internally. */
int gs_bin_speculative(int W, int H, int N, int32_t capacity, const float *packed, const float *depths,
                       int32_t *tile_bins, int32_t *gaussian_ids_sorted, uint16_t *block_masks /*[capacity]*/,
                       int32_t *tile_order /*[tiles], nullable*/,
                       int32_t *num_isects_host /*pinned host int32[2], nullable*/,
                       const int32_t *list_stats /*host int32[2] of an earlier frame, nullable*/, void *workspace,
                       size_t workspace_bytes, gs_stream_t stream);

/* gs_bin_speculative that ALSO zeroes a buffer of the caller's (16-byte aligned, a multiple of 16 bytes) on the way:
 * the gradient-record workspace of the gs_rasterize_backward that follows, which may then be given
 * GS_FLAG_RECORDS_ZEROED and skips its fill (64 MB at 1 M Gaussians: a 12.6 us kernel of its own otherwise).  The stores
 * are issued by the count pass, which waits for its atomics most of the time; the buffer is zero once the call's
 * launches have run, in stream order.  Returns GS_OK_NOT_ZEROED (1: the lists are built, the buffer is UNTOUCHED) for
 * buffers beyond 96 MB — at 5 M Gaussians the count pass has no slack for 320 MB of stores and a fill of that size is
 * best left where it was, right in front of the backward's atomics: the caller then simply does not pass
 * GS_FLAG_RECORDS_ZEROED. */
#define GS_OK_NOT_ZEROED 1
int gs_bin_speculative_zero(int W, int H, int N, int32_t capacity, const float *packed, const float *depths,
                            int32_t *tile_bins, int32_t *gaussian_ids_sorted, uint16_t *block_masks,
                            int32_t *tile_order, int32_t *num_isects_host, const int32_t *list_stats, void *workspace,
                            size_t workspace_bytes, void *zero_ptr /*nullable*/, size_t zero_bytes, gs_stream_t stream);

/* The same lists as gs_bin_scan + gs_bin_sort through a two-level partition (round 6), in ONE call and without a
 * host synchronisation: Gaussians -> strips of sixteen consecutive tiles of a tile row (32-byte records
 * {depth key, id, rectangle, block-row table}, counted with one LDS atomic per strip) -> one workgroup per strip
 * counts, scans and fills its sixteen tiles' segments -> the per-tile sorts.  No per-tile count pass over the
 * Gaussians, no scan launch.  Same contract as the pair it replaces: tile_bins[tiles,2] tile-major and
 * contiguous, lists ordered by (depth, id), block_masks, tile_order (tiles by descending list length, in
 * steps of eight entries), {M, longest list} to num_isects_host (pinned, nullable; [0] is stored by the second
 * kernel, [1] by the last one) and M on the device at gs_bin_num_isects_offset; `capacity` may be a guess
 * (slots beyond it are never written, tile_bins is clamped).  Images of more than 8192 strips (beyond
 * 7680 x 4320) take the tile-level kernels inside the same call.  Workspace: gs_bin_workspace_bytes. */
int gs_bin_strips(int W, int H, int N, int32_t capacity, const float *packed, const float *depths,
                  int32_t *tile_bins, int32_t *gaussian_ids_sorted, uint16_t *block_masks /*[capacity]*/,
                  int32_t *tile_order /*[tiles]*/, int32_t *num_isects_host /*pinned host int32[2], nullable*/,
                  const int32_t *list_stats /*host int32[2] of an earlier frame, nullable*/, void *workspace,
                  size_t workspace_bytes, gs_stream_t stream);

/* Coverage masks of a sorted list: block_masks[i], bit 4 r + c set <=> list entry i (Gaussian
 * gaussian_ids_sorted[i] in the tile whose segment holds i) can reach the 4x4-pixel block at block
 * column c, block row r of that 16x16 tile — its sigma <= sigma_max ellipse (alpha >= 1/255,
 * gsplat_cpu.cpp:220-222) and its rectangle intersect the block (conservatively: a superset of the
 * blocks holding a composited pixel).  The compositing kernels walk a block's list from these bits
 * and never gather a record that misses the wave's part of the tile.  gs_bin_sort / gs_bin_and_sort
 * fill them themselves; this entry point serves lists built elsewhere (the reference's own global
 * sort, include/gsplat_compat.h).  There is no counterpart in the reference: its kernels evaluate all
 * 256 pixels of a tile for every list entry (forward.cu:301-361). */
int gs_block_masks(int W, int H, const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                   const float *packed, uint16_t *block_masks, gs_stream_t stream);

/* gs_bin_scan + stream synchronisation + gs_bin_sort in one call (binAndSortGaussians,
 * rasterize_gaussians.cpp:6-37 together with its caller's cumsum/.item(), :62-63): for callers
 * that keep a gaussian_ids_sorted buffer of `capacity` entries and a workspace of
 * gs_bin_workspace_bytes(N, capacity, W, H) bytes across calls.  *num_isects_host (pinned host
 * memory, required) receives M; returns GS_ERR_CAPACITY without sorting if M > capacity — grow
 * the buffers to *num_isects_host and call again.  Blocks the calling thread once (where the
 * reference blocks). */
int gs_bin_and_sort(int W, int H, int N, int32_t capacity, const float *packed,
                    const float *depths, int32_t *tile_bins, int32_t *gaussian_ids_sorted,
                    uint16_t *block_masks /*[capacity]*/,
                    int32_t *tile_order /*[tiles], nullable*/, int32_t *num_isects_host /*pinned host int32[2]*/, void *workspace,
                    size_t workspace_bytes,
                    gs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Compositing.  Replace rasterize_forward_tensor / rasterize_backward_tensor
 * (bindings.h:111-126,169-190; kernels forward.cu:256-378, backward.cu:161-355); recurrence and
 * thresholds as the CPU oracle gsplat_cpu.cpp:188-240 (fwd) and :313-373 (bwd).
 *   fwd out: out_img[H,W,3] final_Ts[H,W] final_idx[H,W] (index into gaussian_ids_sorted of the
 *            last Gaussian composited into the pixel, -1 if none)
 *   bwd out: v_xy[N,2] v_conic[N,3] v_colors[N,3] v_opacity[N] — fully written (no pre-zeroing
 *            needed; the reference allocates them with torch::zeros, bindings.cu:591-598, and
 *            accumulates into them).  Partial gradients are accumulated with fp32 atomics into
 *            64-byte per-Gaussian records in `workspace` (gs_rasterize_backward_workspace_bytes(N)
 *            bytes, 64-byte aligned) and split into the four tensors at the end.
 *            tile_order (from gs_bin_scan): the launch starts with the longest lists.  list_stats
 *            (the scan's {M, longest list}; nullable): the forward ignores it (every tile is
 *            composited by four quadrant waves); the backward gives a tile to ONE wave with four
 *            pixels per lane, and — when the statistics are unknown or their longest list exceeds
 *            max(512, 2 x mean) — the tiles whose own list exceeds that length to four waves with
 *            one pixel per lane, inside the same launch; a frame of fewer than 1280 / 2560 tiles
 *            gives EVERY tile four / two waves.  Scheduling only: the sums differ by atomic order
 *            as always.  (Round 5: the one-wave-per-tile launch runs sixteen four-lane groups per wave,
 *            one per 4x4-pixel block — gs_raster.hip, backward_wave_q.  Measurement / test bits of
 *            `flags`, not part of the contract: bits 21..22 = 1 / 2 / 3 force one / two / four pixels
 *            per lane with the four-group kernels for every tile; bits 23..24 = 1 / 2 force one / two
 *            list entries per forward step, 3 = the chunk-compacting forward of round 6 on full frames; bits 25..26 = 1 the sixteen-group backward for every
 *            frame, 2 the four-group kernels of rounds 2 - 4.)
 *            v_out_alpha may be NULL (OpenSplat always passes zeros,
 *            rasterize_gaussians.cpp:108).  background: float[3] in host OR device memory
 *            (a device tensor is read by the kernels themselves: no copy, no synchronisation). */
int gs_rasterize_forward(int W, int H, const int32_t *gaussian_ids_sorted,
                         const uint16_t *block_masks /*from gs_bin_sort / gs_block_masks*/,
                         const int32_t *tile_bins, const float *packed,
                         const float *background /*host or device [3]*/, float *out_img, float *final_Ts,
                         int32_t *final_idx,
                         float *out_img_clamped /*[H,W,3], required with GS_FLAG_CLAMP_IMAGE*/,
                         const int32_t *list_stats /*host int32[2], nullable*/,
                         const int32_t *tile_order /*device [tiles], nullable*/, uint32_t flags,
                         gs_stream_t stream);

size_t gs_rasterize_backward_workspace_bytes(int N);
/* workspace size under GS_FLAG_DETERMINISTIC: the float records followed by 64-bit fixed-point
 * accumulators (scale 2^40), into which the per-wave partial sums are added with integer atomics —
 * integer addition commutes, so the result does not depend on the order the waves arrive in.
 * Quantisation: each flushed partial sum (one per entry, wave and 64-entry chunk) is rounded to the
 * nearest multiple of 2^-40 = 9.1e-13 and saturates at +-2^22; with cotangents of order 1 / (3 H W)
 * (a mean-reduced image loss: 1.6e-7 at 1080p) a Gaussian's weakest contributions therefore carry a
 * relative error of up to a few 1e-6 each — a debugging aid for race detection (a difference between
 * two runs under the flag is a race, not summation order), not a precision mode. */
size_t gs_rasterize_backward_workspace_bytes_det(int N);

int gs_rasterize_backward(int W, int H, int N, const int32_t *gaussian_ids_sorted,
                          const uint16_t *block_masks,
                          const int32_t *tile_bins, const float *packed,
                          const float *background /*host or device [3]*/, const float *final_Ts,
                          const int32_t *final_idx, const float *v_out,
                          const float *v_out_alpha /*nullable*/,
                          const float *out_img /*raw image, required with GS_FLAG_CLAMP_IMAGE*/,
                          float *v_xy, float *v_conic, float *v_colors, float *v_opacity,
                          void *workspace, size_t workspace_bytes,
                          const int32_t *list_stats /*host int32[2], nullable*/,
                          const int32_t *tile_order /*device [tiles], nullable*/, uint32_t flags,
                          gs_stream_t stream);

/* Frames that do not fill the chip with one wave per tile, and frames whose longest tile lists lie far beyond
 * the mean (the reduced resolutions OpenSplat's resolution schedule starts a run with, model.cpp:85-92; small
 * and mid-size captures — up to 6144 tiles, 1.5 Mpixel): the compositing launches last as long as one wave
 * needs for the LONGEST list.  No counterpart in the reference, which gives a tile's list to one workgroup
 * (forward.cu:256-378, backward.cu:161-355).  Two things help, both scheduling only:
 *  - the forward takes two list entries per step on frames of at most 2560 tiles (same bits; chosen by itself);
 *  - the backward can start anywhere in a list if it is handed the state in front of that entry.  With a
 *    checkpoint buffer the forward stores four floats per pixel every `seg_len` entries of a tile's list
 *    (and at its end) — the state of the BACKWARD's recurrence there: transmittance and colour buffer as
 *    gsplat_cpu.cpp:337-352 would have them, i.e. with alpha clamped at 0.99 where the forward clamps at
 *    0.999 (:220, :338) — and the backward runs every piece of every list as a wave of its own; the
 *    pieces of a Gaussian's gradient meet in its record with the same atomics as the tiles' always did
 *    (sums differ by their order, as between any two runs).  A piece starts from the forward's own
 *    transmittance product instead of the product of reciprocals the one-pass backward (and
 *    gsplat_cpu.cpp:337-345) unwinds to there: the two differ by the rounding of that unwinding.
 * gs_rasterize_checkpoint_plan: from the scan's {M, longest list} of the previous frame (host, nullable)
 *   -> seg_len (a power of two >= 64), max_segments, bytes of the buffer; bytes = 0: not worthwhile (more than
 *   6144 tiles, no statistics yet, short lists, or — beyond 960 tiles — no list four times the mean) — call
 *   the plain entry points.  A list that outgrows the plan is finished by its last piece.  The pieces' waves
 *   hold 1 / 2 / 4 pixels per lane up to 128 / 1024 / more tiles.
 * gs_rasterize_forward_ckpt / gs_rasterize_backward_ckpt: the entry points above + the buffer (16-byte
 *   aligned, >= tiles * max_segments * 4096 bytes; NULL: exactly the plain call).  The backward must be
 *   given the buffer, seg_len and max_segments its forward wrote with. */
int gs_rasterize_checkpoint_plan(int W, int H, const int32_t *list_stats /*host int32[2], nullable*/,
                                 int32_t *seg_len, int32_t *max_segments, size_t *bytes);
int gs_rasterize_forward_ckpt(int W, int H, const int32_t *gaussian_ids_sorted,
                              const uint16_t *block_masks, const int32_t *tile_bins, const float *packed,
                              const float *background, float *out_img, float *final_Ts,
                              int32_t *final_idx, float *out_img_clamped, const int32_t *list_stats,
                              const int32_t *tile_order, uint32_t flags, void *checkpoints,
                              size_t checkpoint_bytes, int32_t seg_len, int32_t max_segments,
                              gs_stream_t stream);
int gs_rasterize_backward_ckpt(int W, int H, int N, const int32_t *gaussian_ids_sorted,
                               const uint16_t *block_masks, const int32_t *tile_bins, const float *packed,
                               const float *background, const float *final_Ts, const int32_t *final_idx,
                               const float *v_out, const float *v_out_alpha, const float *out_img,
                               float *v_xy, float *v_conic, float *v_colors, float *v_opacity,
                               void *workspace, size_t workspace_bytes, const int32_t *list_stats,
                               const int32_t *tile_order, uint32_t flags, const void *checkpoints,
                               size_t checkpoint_bytes, int32_t seg_len, int32_t max_segments,
                               gs_stream_t stream);

/* Test hook: y[i] = the exponential exactly as the compositing kernels evaluate it (glibc-bit-exact
 * by default, hardware v_exp_f32 with GS_FLAG_FAST_EXP); valid for |x| < 87. */
int gs_debug_expf(int64_t n, const float *x, float *y, uint32_t flags, gs_stream_t stream);

/* Measurement hook: records the two hipEvent_t (passed as void*) on the kernel's stream right
 * before and right after the NEXT k_rasterize_forward / k_rasterize_backward launch made by the
 * calling thread (the compositing kernel alone), then disarms itself. */
int gs_debug_time_next_kernel(void *event_start, void *event_stop);

/* Measurement hook: the kernel timeline of the calling thread.  gs_debug_timeline(1) arms it and forgets
 * earlier records: every kernel (and record memset) the library launches from this thread from then on is
 * bracketed by two HIP events on its launch stream; gs_debug_timeline(0) disarms.  gs_debug_timeline_read
 * waits for the recorded events and returns, in launch order, the kernel names (name_bytes per entry,
 * NUL-terminated, the kernel expression as written at the launch site) and durations in milliseconds;
 * *count = the number of records held (may exceed capacity).  A record costs ~5 us of stream time: use it
 * on an instrumented pass, never inside a timed region (bench.py: `kernels`). */
int gs_debug_timeline(int enable);
int gs_debug_timeline_read(int capacity, char *names, int name_bytes, float *ms, int *count);

/* Test hook: the per-row (16-lane) nine-value reduction of the compositing backward (a transposing
 * DPP butterfly, no LDS).  in [blocks, 9, 64]  ->  out[blocks, 4, 9] = sums over each 16-lane row. */
int gs_debug_row_reduce9(int blocks, const float *in, float *out, gs_stream_t stream);

/* Test hook: the nine-value group reduction as the compositing backward runs it.  mfma = 0: the DPP
 * butterfly above, a group is a 16-lane row; mfma = 1: nine v_mfma_f32_16x16x4_f32 with one-hot B columns
 * + three additions, a group is the sixteen lanes {4 g + q + 16 k : q, k < 4} (one quad of each row).
 * in [blocks, 9, 64]  ->  out[blocks, 4, 9] = the nine sums of each group.
 * gs_debug_backward_uses_mfma: which of the two the library's k_rasterize_backward was built with. */
int gs_debug_group_reduce9(int blocks, const float *in, float *out, int mfma, gs_stream_t stream);
int gs_debug_backward_uses_mfma(void);


/* ---------------------------------------------------------------------------------------------
 * Fused per-Gaussian stages (second step).  The three per-Gaussian forward
 * stages — gs_project_forward, gs_sh_forward_fused, gs_pack_splats — and the three backward ones —
 * the record split of gs_rasterize_backward, gs_sh_backward_fused, gs_project_backward — each as
 * ONE kernel: the 2-D intermediates (xys, conics, cov2d, colours; v_xy, v_conic, v_colors) never
 * touch HBM.  Same device functions as the stage kernels (gs_gaussian.h), so the results are the
 * same bits.  Model::forward's glue is always fused here: `scales` obeys GsCamera.flags
 * (GS_CAM_LOG_SCALES), `opacities` obeys GS_FLAG_LOGIT_OPACITY, colours are max(SH + 0.5, 0).
 *
 * gs_gaussian_forward
 *   in : means[N,3] scales[N,3] quats[N,4] opacities[N] features_dc[N,3] features_rest[N,K-1,3]
 *        (NULL for K = 1), cam_pos[3] (host or device), degrees_to_use
 *   out: packed[N,12] (the record gs_bin_* / gs_rasterize_* consume), depths[N], radii[N],
 *        rgb_raw[N,3] (SH colour before +0.5 / clamp: the backward's clamp mask),
 *        xys[N,2] (optional, NULL to skip: Model::afterTrain only needs its gradient)
 * gs_gaussian_backward
 *   in : the same parameters, radii, rgb_raw, and `records` = the workspace gs_rasterize_backward
 *        filled under GS_FLAG_KEEP_RECORDS
 *   out: v_means[N,3] v_scales[N,3] v_quats[N,4] v_opacity[N] v_dc[N,3] v_rest[N,K-1,3],
 *        v_xy[N,2] (optional: d loss / d xys for the densification statistics);
 *        with GS_FLAG_RECORDS_ZEROED in `flags` the records are zeroed behind the read, so that the
 *        next gs_rasterize_backward may be given the same flag and skip its memset.  (Measured on
 *        MI355X the memset is the better choice at 1 M Gaussians: it leaves the records in the
 *        last-level cache right before the atomics arrive.) */
int gs_gaussian_forward(const GsCamera *cam, const float *viewmat_dev, const float *projmat_dev,
                        int N, int K, int degrees_to_use, const float *means, const float *scales,
                        const float *quats, const float *opacities, const float *features_dc,
                        const float *features_rest, const float *cam_pos, float *packed,
                        float *depths, int32_t *radii, float *rgb_raw, float *xys, uint32_t flags,
                        gs_stream_t stream);
int gs_gaussian_backward(const GsCamera *cam, const float *viewmat_dev, const float *projmat_dev,
                         int N, int K, int degrees_to_use, const float *means, const float *scales,
                         const float *quats, const float *opacities, const float *cam_pos,
                         const int32_t *radii, const float *rgb_raw, void *records,
                         size_t records_bytes, float *v_means, float *v_scales, float *v_quats,
                         float *v_opacity, float *v_dc, float *v_rest, float *v_xy, uint32_t flags,
                         gs_stream_t stream);

/* SH gradients from the colour cotangents of several cameras: v_sh[n][b] = sum_c basis_b(dir(n, c)) *
 * v_colors[c][n] (spherical_harmonics.cpp:60-77 / backward.cu compute_sh_backward per camera, summed
 * in camera order).  The camera-per-rank path exchanges v_colors (12 B per Gaussian and camera,
 * all-gather) instead of all-reducing the 12 K bytes of SH gradients (include/gsplat_dist.h).
 *   in : means[N,3]; cam_pos_dev: n_cams camera centres on the DEVICE, cam_pos_stride floats apart;
 *        v_colors: n_cams blocks of [N,3], v_colors_stride floats apart (what gs_gaussian_backward
 *        wrote to v_dc under GS_FLAG_EMIT_VCOLOR; all-zero rows are skipped)
 *   out: v_dc[N,3] v_rest[N,K-1,3]; GS_FLAG_ACCUMULATE_GRADS adds instead of overwriting            */
int gs_sh_backward_cameras(int N, int K, int degrees_to_use, int n_cams, const float *means,
                           const float *cam_pos_dev, size_t cam_pos_stride, const float *v_colors,
                           size_t v_colors_stride, float *v_dc, float *v_rest, uint32_t flags,
                           gs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_HIP_H */
