// gs_bin.hip — per-Gaussian 2-D record packing, tile counting, intersection scatter and per-tile
// depth sort.
//
// Replaces, on the reference side: torch::cumsum + .item() (rasterize_gaussians.cpp:62-63),
// map_gaussian_to_intersects (forward.cu:107-143), torch::sort + torch::gather
// (rasterize_gaussians.cpp:25-32) and get_tile_bin_edges (forward.cu:148-169).
//
// The reference builds the per-tile depth-ordered lists with ONE global sort of M
// (tile<<32 | depth) int64 keys plus int64 argsort indices (8 radix passes over 16-byte pairs and
// a gather).  On MI355X a counting partition followed by many small on-chip sorts is far cheaper:
//   1. k_count_tiles   — one lane per Gaussian adds 1 to the counter of every tile its rectangle
//                        overlaps (fire-and-forget L2 atomics);
//   2. k_scan_tiles    — one workgroup scans the (<= ~130 k) tile counters: tile_bins[t] =
//                        [start, end), scatter cursors, total M;
//   3. k_scatter       — one lane per Gaussian claims a slot in each of its tiles' segments
//                        (returning atomics on the cursors) and writes the 64-bit key
//                        (order-preserving depth bits << 32 | Gaussian id);
//   4. k_bucket_sort_* — one wave (or workgroup) per tile orders its segment by that key on chip:
//                        a counting sort on the leading bits of (depth - nearest depth of the
//                        tile) puts almost every key in its own bucket, a lane finishes the few
//                        buckets with several keys; heavily tied depths fall back to a bitonic
//                        network, segments too long for LDS are sorted in place in global memory.
// The key is unique per (depth, id), so the result does not depend on the order in which step 3's
// atomics land: lists are depth ordered with ties in Gaussian-index order —
// deterministic, and identical to a stable sort by depth.  Traffic: 8 B written + read per
// intersection plus 4 B of ids, against >= 128 B per intersection for the reference's scheme.
//
// Which tiles a Gaussian lands in is decided by the CPU oracle's pixel rectangle tightened by the
// alpha-threshold ellipse box (k_pack_splats), not by the GPU reference's radius square;
// tile_bins is [tiles, 2] (the reference allocates [M, 2], bindings.cu:324-326).
#include <cstring>
#include <mutex>
#include <unordered_map>

#include "gs_gaussian.h"

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
              int32_t *__restrict__ tiles_hit, uint32_t flags) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float x = xys[2 * n], y = xys[2 * n + 1];
    float A = conics[3 * n], B = conics[3 * n + 1], C = conics[3 * n + 2];
    float4 p0, p1, p2;
    const int tiles = pack_one(W, H, x, y, A, B, C, cov2d != nullptr, cov2d ? cov2d[3 * n] : 0.0f,
                               cov2d ? cov2d[3 * n + 2] : 0.0f, opacities[n], radii[n],
                               colors[3 * n], colors[3 * n + 1], colors[3 * n + 2], flags, p0, p1, p2);
    packed[3 * n + 0] = p0;
    packed[3 * n + 1] = p1;
    packed[3 * n + 2] = p2;
    tiles_hit[n] = tiles;
}

// ---- tile rectangle of a packed record --------------------------------------------------------
struct TileRect {
    int tx0, tx1, ty0, ty1;
    __device__ __forceinline__ int count() const { return (tx1 - tx0) * (ty1 - ty0); }
};
__device__ __forceinline__ TileRect tile_rect_of(uint32_t rx, uint32_t ry) {
    int x0 = rx & 0xFFFF, x1 = rx >> 16, y0 = ry & 0xFFFF, y1 = ry >> 16;
    TileRect r;
    r.tx0 = x0 / GS_TILE; r.tx1 = (x1 + GS_TILE - 1) / GS_TILE;
    r.ty0 = y0 / GS_TILE; r.ty1 = (y1 + GS_TILE - 1) / GS_TILE;
    if (x1 <= x0 || y1 <= y0) r.tx1 = r.tx0, r.ty1 = r.ty0;
    return r;
}
__device__ __forceinline__ TileRect tile_rect(const float4 *__restrict__ packed, int n) {
    uint32_t rx = __float_as_uint(packed[3 * (size_t)n + 1].w);
    uint32_t ry = __float_as_uint(packed[3 * (size_t)n + 2].w);
    int x0 = rx & 0xFFFF, x1 = rx >> 16, y0 = ry & 0xFFFF, y1 = ry >> 16;
    TileRect r;
    r.tx0 = x0 / GS_TILE; r.tx1 = (x1 + GS_TILE - 1) / GS_TILE;
    r.ty0 = y0 / GS_TILE; r.ty1 = (y1 + GS_TILE - 1) / GS_TILE;
    if (x1 <= x0 || y1 <= y0) r.tx1 = r.tx0, r.ty1 = r.ty0;
    return r;
}

// Visit every tile of every Gaussian of this wave's 64-lane slice.  Small rectangles are walked
// by their own lane; a rectangle with more than kLaneTiles tiles is walked by the whole wave
// (one lane looping over thousands of tiles would stall the other 63).
constexpr int kLaneTiles = 16;
// threads per workgroup of the LDS-privatised count / scatter kernels: one workgroup per CU (its
// LDS table may take most of the CU's 160 KiB), so the workgroup must bring enough waves by itself
// to cover the latency of the record loads (with 256 threads the kernels were latency-bound)
constexpr int kPersistentThreads = 1024;
// f(tile, pa, pb): (pa, pb) is the owning lane's payload (broadcast when the wave walks it).
template <typename F>
__device__ __forceinline__ void for_each_tile(const TileRect &r, bool valid, int tiles_x,
                                              uint32_t pa, int pb, F f) {
    const int cnt = valid ? r.count() : 0;
    if (cnt > 0 && cnt <= kLaneTiles) {
        for (int ty = r.ty0; ty < r.ty1; ty++)
            for (int tx = r.tx0; tx < r.tx1; tx++) f(ty * tiles_x + tx, pa, pb);
    }
    uint64_t big = __builtin_amdgcn_ballot_w64(cnt > kLaneTiles);
    const int lane = threadIdx.x & 63;
    while (big) {
        const int src = __builtin_ctzll(big);
        big &= big - 1;
        const int tx0 = __builtin_amdgcn_readlane(r.tx0, src);
        const int tx1 = __builtin_amdgcn_readlane(r.tx1, src);
        const int ty0 = __builtin_amdgcn_readlane(r.ty0, src);
        const int ty1 = __builtin_amdgcn_readlane(r.ty1, src);
        const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)pa, src);
        const int b = __builtin_amdgcn_readlane(pb, src);
        const int w = tx1 - tx0, total = w * (ty1 - ty0);
        for (int i = lane; i < total; i += 64) f((ty0 + i / w) * tiles_x + tx0 + i % w, a, b);
    }
}

// ---- 1. count ----------------------------------------------------------------------------------
// Global atomics run at ~27 G requests/s on MI355X whatever the layout (scripts/ubench/atomics.hip),
// LDS atomics are an order of magnitude faster: a few persistent workgroups each keep a private
// table of all tile counters in LDS (4 B x tiles: 32 KiB at 1080p, 127 KiB at 4K, of the CU's
// 160 KiB) and flush the non-zero ones with coalesced global atomics at the end.
// k_count_tiles_global is the fallback for images with more tiles than fit in LDS.
__global__ void __launch_bounds__(256)
k_count_tiles_global(int N, int tiles_x, const float4 *__restrict__ packed,
                     int32_t *__restrict__ counts) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = n < N;
    TileRect r = {0, 0, 0, 0};
    if (valid) r = tile_rect(packed, n);
    for_each_tile(r, valid, tiles_x, 0u, 0,
                  [&](int tile, uint32_t, int) { atomicAdd(&counts[tile], 1); });
}

// wg_base[workgroup][tile]: where, inside the tile's segment, the intersections this workgroup will
// emit start — the value the flush's (returning) atomic hands back.  k_scatter walks the same
// Gaussians in the same workgroup layout and simply continues from there: it used to recount them
// and reserve its ranges with a second round of returning atomics (48 -> 33 us at C2).
__global__ void __launch_bounds__(kPersistentThreads)
k_count_tiles(int N, int tiles, int tiles_x, const float4 *__restrict__ packed,
              int32_t *__restrict__ counts, int32_t *__restrict__ wg_base, uint4 *__restrict__ zero_ptr,
              size_t zero_n16) {
    extern __shared__ int32_t h[];
    // zero_ptr (gs_bin_speculative_zero, round 6): a buffer of the CALLER's to be zeroed on the way — the gradient
    // records of the compositing backward that follows, 64 MB at 1 M Gaussians, otherwise a 12.6 us fill kernel in
    // front of that backward.  This kernel waits 73 % of its cycles and moves 65 MB in 19 us; the stores are issued
    // first and drain under the walk.  (Where in the step the records are zeroed does not matter to the backward —
    // measured with the fill moved in front of the binning / the forward: step and kernel level.)
    if (zero_ptr) {
        const size_t total = (size_t)gridDim.x * blockDim.x;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < zero_n16; i += total)
            zero_ptr[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    // 64-Gaussian chunks dealt out wave by wave ACROSS the workgroups (chunk = wave-major), so that a
    // few thousand Gaussians with huge rectangles still keep every CU's waves busy.  A wave walks ~4
    // chunks at 1 M Gaussians: the rectangle of the NEXT chunk is requested before the current one is
    // walked (and the first one before the table is cleared) — a chain of dependent round trips
    // otherwise, which is what this kernel spends its time on.
    const int stride = (blockDim.x >> 6) * gridDim.x, lane = threadIdx.x & 63;
    int chunk = (threadIdx.x >> 6) * gridDim.x + blockIdx.x;
    auto request = [&](int ch, uint32_t &rx, uint32_t &ry) {
        const int64_t n = (int64_t)ch * 64 + lane;   // (the chunk beyond the last one may not fit an int)
        const bool ok = n < N;
        rx = ok ? __float_as_uint(packed[3 * (size_t)n + 1].w) : 0u;
        ry = ok ? __float_as_uint(packed[3 * (size_t)n + 2].w) : 0u;
        return ok;
    };
    uint32_t rx, ry;
    bool valid = request(chunk, rx, ry);
    for (int t = threadIdx.x; t < tiles; t += blockDim.x) h[t] = 0;
    __syncthreads();
    // every iteration is taken by whole waves (for_each_tile uses wave-wide operations)
    for (; (int64_t)chunk * 64 < N; chunk += stride) {
        uint32_t nrx, nry;
        const bool nvalid = request(chunk + stride, nrx, nry);
        const TileRect r = tile_rect_of(rx, ry);
        for_each_tile(r, valid, tiles_x, 0u, 0,
                      [&](int tile, uint32_t, int) { atomicAdd(&h[tile], 1); });
        rx = nrx; ry = nry; valid = nvalid;
    }
    __syncthreads();
    // this workgroup's offset inside every tile's segment = the running total at the moment its count is
    // added.  Eight returning atomics in flight per thread (one after the other they are a chain of
    // round trips: 8 per thread at 1080p, 32 at 4K)
    int32_t *my_base = wg_base + (size_t)blockIdx.x * tiles;
    const int step = (int)blockDim.x;
    for (int t0 = threadIdx.x; t0 < tiles; t0 += 8 * step) {
        int32_t c[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; j++) c[j] = t0 + j * step < tiles ? h[t0 + j * step] : 0;
#pragma unroll
        for (int j = 0; j < 8; j++) b[j] = c[j] ? atomicAdd(&counts[t0 + j * step], c[j]) : 0;
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (t0 + j * step < tiles) my_base[t0 + j * step] = b[j];
    }
}

// ---- 2. scan -----------------------------------------------------------------------------------
// One 1024-thread workgroup.  The counters are first copied to LDS with coalesced loads (up to
// 36 864 tiles = 144 KiB), each thread then scans a contiguous slice there and the [start, end)
// pairs are written back coalesced; larger images scan straight from global memory.
// LDS index skew: one pad word per 32 so that threads walking contiguous 32-word-aligned slices hit
// different banks
__device__ __forceinline__ int skew(int i) { return i + (i >> 5); }

// Inclusive scan of one int per thread over a 1024-thread workgroup: shuffles inside the waves,
// the sixteen wave totals through LDS (three barriers instead of the twenty of a Hillis-Steele
// scan in LDS).  `ws` holds 17 ints; ws[16] receives the grand total.
static __device__ __forceinline__ int32_t block_scan_1024(int32_t v, int32_t *ws) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t u = __shfl_up(v, d, 64);
        if (lane >= d) v += u;
    }
    __syncthreads();  // ws may still be read from a previous call
    if (lane == 63) ws[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t run = 0;
        for (int w = 0; w < 16; w++) {
            const int32_t x = ws[w];
            ws[w] = run;
            run += x;
        }
        ws[16] = run;
    }
    __syncthreads();
    return v + ws[wave];
}

// Inclusive scan of one int per lane over a wave: Hillis-Steele inside each 16-lane row (row_shr with
// zero fill), then the two row carries.
__device__ __forceinline__ int wave_inclusive_scan_i(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);  // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);  // row_bcast:31 -> rows 2,3
    return v;
}

// ---- the order the compositing launches start their tiles in ---------------------------------------
// Longest lists first ("longest processing time first": evens out the end of the launches) — but only to
// classes of sixteen entries, and INSIDE a class the tiles are scattered over the image: tile j of the
// sequence i(j) = j * mult mod tiles (mult ~ 0.618 tiles, coprime to tiles) is visited j-th, so tiles that
// start together are far apart.  Measured (scripts/exp_tile_order.py, profiles/exp_tile_order_r06_*.json):
// with the exact order of rounds 2 - 5 (ties in the arrival order of contiguous slices: neighbouring tiles
// next to each other) the C2 compositing kernels take 174 / 273 us, with tiles of a strip side by side
// 198 / 298 us, with this order 162 / 237 us — neighbouring tiles gather the same Gaussians' lines and add to
// the same gradient records at the same time.  Scheduling only.
__device__ __forceinline__ int order_shift(int longest) {
    int shift = 4;
    while ((longest >> shift) >= 1024) shift++;
    return shift;
}
__device__ __forceinline__ int order_first(int t, int tiles, int mult) {
    return (int)(((int64_t)t * mult) % tiles);
}
__device__ __forceinline__ int order_step(int tiles, int mult) { return (int)(((int64_t)1024 * mult) % tiles); }
__device__ __forceinline__ int order_next(int i, int tiles, int step) {   // i(j + 1024) from i(j)
    i += step;
    return i >= tiles ? i - tiles : i;
}

// The scan for images whose counters fit in LDS (up to 36 864 tiles) — the usual case.  One 1024-thread
// workgroup sits between two launches that fill the chip, so what counts is its LATENCY: every thread
// reads its contiguous slice of counters straight from global memory (a wave's slices are contiguous:
// coalesced), the prefix is a DPP wave scan + the sixteen wave totals through LDS, and the counting
// sort for the longest-first tile order reuses the same pattern: five barriers in all (the general
// kernel below: a dozen, plus a Hillis-Steele loop of shuffles) — 14 -> ~6 us at 1080p.
// (the body: also run by the extra workgroup of k_scatter_scan, which folds this launch into the scatter's)
__device__ __forceinline__ void scan_tiles_fast_body(int tiles, int order_mult, const int32_t *__restrict__ counts,
                                                     int2 *__restrict__ bins, int32_t *__restrict__ total_dev,
                                                     int32_t *__restrict__ total_host,
                                                     int32_t *__restrict__ order, int32_t *c_lds) {
    __shared__ int32_t part[1024];
    __shared__ int32_t wsum[16], wmax[16], psum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (tiles + 1023) / 1024;
    const int lo = min(t * per, tiles), hi = min(lo + per, tiles);
    int32_t sum = 0, longest = 0;
    for (int i = lo; i < hi; i++) {
        const int32_t c = counts[i];
        c_lds[skew(i)] = c;
        sum += c;
        longest = max(longest, c);
    }
    const int32_t incl = wave_inclusive_scan_i(sum);
    const int32_t wl = wave_max_i(longest);
    if (lane == 63) wsum[wave] = incl;
    if (lane == 0) wmax[wave] = wl;
    part[t] = 0;
    __syncthreads();
    int32_t base = 0, total = 0, longest_all = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const int32_t v = wsum[w];
        base += w < wave ? v : 0;
        total += v;
        longest_all = max(longest_all, wmax[w]);
    }
    int32_t run = base + incl - sum;   // exclusive prefix of this thread's slice
    for (int i = lo; i < hi; i++) {
        const int32_t c = c_lds[skew(i)];
        bins[i] = make_int2(run, run + c);
        run += c;
    }
    if (t == 1023) {
        *total_dev = total;
        // pinned (device-mapped) host memory: the stores land there without a copy kernel
        if (total_host) {
            total_host[0] = total;
            total_host[1] = longest_all;
        }
    }
    if (!order) return;
    // tiles by descending list length in classes of (at least) sixteen entries, scattered over the image inside a
    // class (order_classes below): counting sort on min(n >> shift, 1023), bucket 0 = longest
    const int shift = order_shift(longest_all), ostep = order_step(tiles, order_mult);
    for (int k = 0, i = order_first(t, tiles, order_mult); k * 1024 + t < tiles; k++, i = order_next(i, tiles, ostep))
        atomicAdd(&part[1023 - (c_lds[skew(i)] >> shift)], 1);
    __syncthreads();
    const int32_t own = part[t];
    const int32_t incl2 = wave_inclusive_scan_i(own);
    if (lane == 63) psum[wave] = incl2;
    __syncthreads();
    int32_t base2 = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) base2 += w < wave ? psum[w] : 0;
    part[t] = base2 + incl2 - own;
    __syncthreads();
    for (int k = 0, i = order_first(t, tiles, order_mult); k * 1024 + t < tiles; k++, i = order_next(i, tiles, ostep))
        order[atomicAdd(&part[1023 - (c_lds[skew(i)] >> shift)], 1)] = i;
}

__global__ void __launch_bounds__(1024)
k_scan_tiles_fast(int tiles, int order_mult, const int32_t *__restrict__ counts, int2 *__restrict__ bins,
                  int32_t *__restrict__ total_dev, int32_t *__restrict__ total_host,
                  int32_t *__restrict__ order) {
    extern __shared__ int32_t c_lds[];
    scan_tiles_fast_body(tiles, order_mult, counts, bins, total_dev, total_host, order, c_lds);
}

__global__ void __launch_bounds__(1024)
k_scan_tiles(int tiles, int use_lds, int order_mult, const int32_t *__restrict__ counts, int2 *__restrict__ bins,
             int32_t *__restrict__ total_dev, int32_t *__restrict__ total_host,
             int32_t *__restrict__ order) {
    extern __shared__ int32_t c_lds[];
    __shared__ int32_t part[1024];
    const int t = threadIdx.x;
    const int per = (tiles + 1023) / 1024;
    const int lo = min(t * per, tiles), hi = min(lo + per, tiles);
    if (use_lds) {
        for (int i = t; i < tiles; i += 1024) {
            c_lds[skew(i)] = counts[i];
        }
        __syncthreads();
    }
    int32_t sum = 0, longest = 0;
    if (use_lds) for (int i = lo; i < hi; i++) { const int32_t c = c_lds[skew(i)]; sum += c; longest = max(longest, c); }
    else for (int i = lo; i < hi; i++) { const int32_t c = counts[i]; sum += c; longest = max(longest, c); }
    // longest list of the frame (steers how many waves share a tile in the compositing kernels)
    __shared__ int32_t s_longest;
    if (t == 0) s_longest = 0;
    __syncthreads();
    atomicMax(&s_longest, longest);
    __shared__ int32_t ws[17];
    const int32_t incl = block_scan_1024(sum, ws);
    const int32_t total_keep = ws[16];
    int32_t run = incl - sum;  // exclusive prefix of this thread's slice
    if (use_lds) {
        for (int i = lo; i < hi; i++) {  // counts -> exclusive starts, in place
            const int32_t c = c_lds[skew(i)];
            c_lds[skew(i)] = run;
            run += c;
        }
        __syncthreads();
        const int32_t total = total_keep;
        for (int i = t; i < tiles; i += 1024) {
            const int32_t st = c_lds[skew(i)];
            const int32_t en = (i + 1 < tiles) ? c_lds[skew(i + 1)] : total;
            bins[i] = make_int2(st, en);
        }
    } else {
        for (int i = lo; i < hi; i++) {
            const int32_t c = counts[i];
            bins[i] = make_int2(run, run + c);
            run += c;
        }
    }
    // tiles by descending list length (counting sort on min(n >> shift, 1023)): the compositing
    // kernels start the long lists first ("longest processing time first"), which evens out the end
    // of their launches and lets a few heavy tiles overlap with the rest instead of trailing it
    if (order) {
        __syncthreads();
        const int32_t longest_all = s_longest;
        const int shift = order_shift(longest_all), ostep = order_step(tiles, order_mult);
        part[t] = 0;
        __syncthreads();
        // list length of tile i: from the exclusive starts still in LDS (no global read-back)
        auto length = [&](int i) -> int32_t {
            if (!use_lds) return counts[i];
            const int32_t st = c_lds[skew(i)];
            return ((i + 1 < tiles) ? c_lds[skew(i + 1)] : total_keep) - st;
        };
        for (int k = 0, i = order_first(t, tiles, order_mult); k * 1024 + t < tiles; k++, i = order_next(i, tiles, ostep))
            atomicAdd(&part[1023 - (length(i) >> shift)], 1);  // bucket 0 = longest lists
        __syncthreads();
        const int32_t own = part[t];
        const int32_t first = block_scan_1024(own, ws) - own;
        part[t] = first;
        __syncthreads();
        for (int k = 0, i = order_first(t, tiles, order_mult); k * 1024 + t < tiles; k++, i = order_next(i, tiles, ostep))
            order[atomicAdd(&part[1023 - (length(i) >> shift)], 1)] = i;
    }
    if (t == 1023) {
        *total_dev = total_keep;
        // pinned (device-mapped) host memory: the stores land there without a copy kernel
        if (total_host) {
            total_host[0] = total_keep;
            total_host[1] = s_longest;
        }
    }
}

// ---- 3. scatter --------------------------------------------------------------------------------
// One 16-byte record per intersection: { depth key, Gaussian id, coverage mask, - }.
//   depth key = order-preserving uint32 image of the depth float (flip all bits of negatives, set
//   the sign bit of non-negatives).  The reference uses the raw bit pattern, valid only for depth > 0
//   (forward.cu:132); the map sorts identically there and stays correct for ANY key, which lets tests
//   drive the sort with the CPU reference's as-read keys.
//   coverage mask = which of the tile's sixteen 4x4-pixel blocks the Gaussian can reach
//   (block_mask16, gs_device.h).  It is formed HERE, where a lane holds its Gaussian's record in
//   registers: the block-row table of the Gaussian once (block_rows_table), then a few integer
//   operations per tile (mask_from_rows) — the kernel waits on its scattered stores, its VALUs were
//   idle (0.8 of 8 busy).  As a kernel of its own over the sorted lists the same masks cost 46 us at
//   C2, gathered per entry in the sorts' epilogue 32 us (404 / 290 us at C3: 19 M random gathers).
// A 16-byte scattered store leaves L2 as one 32-byte write, exactly like the 8-byte store it replaces.
struct SplatForMask {
    float x, y, A, B, C;
    uint32_t smax, rx, ry;
};

// what a lane needs of its Gaussian: requested one chunk ahead of its use (scatter_request), the
// dependent chain of round trips is otherwise what the kernel waits for
struct ScatterIn {
    float4 p0, p1;
    float p2w, depth;
    bool valid;
};
__device__ __forceinline__ ScatterIn scatter_request(const float4 *__restrict__ packed,
                                                     const float *__restrict__ depths, int n, bool valid) {
    ScatterIn in;
    in.valid = valid;
    in.p0 = in.p1 = make_float4(0.f, 0.f, 0.f, 0.f);
    in.p2w = in.depth = 0.f;
    if (valid) {
        in.p0 = packed[3 * (size_t)n + 0];
        in.p1 = packed[3 * (size_t)n + 1];
        in.p2w = packed[3 * (size_t)n + 2].w;
        in.depth = depths[n];
    }
    return in;
}

template <typename Emit>
__device__ __forceinline__ void scatter_tiles(const ScatterIn &in, int n, int tiles_x, Emit emit) {
    const bool valid = in.valid;
    TileRect r = {0, 0, 0, 0};
    uint32_t db = 0;
    SplatForMask g = {0.f, 0.f, 0.f, 0.f, 0.f, 0u, 0u, 0u};
    RowTable w = {kNoRowTable, 0u, 0u, 0u, 0u};
    if (valid) {
        g.x = in.p0.x; g.y = in.p0.y; g.A = in.p0.z; g.B = in.p0.w; g.C = in.p1.x;
        g.smax = __float_as_uint(in.p1.z);
        g.rx = __float_as_uint(in.p1.w);
        g.ry = __float_as_uint(in.p2w);
        r = tile_rect_of(g.rx, g.ry);
        db = __float_as_uint(in.depth);
        db = (db & 0x80000000u) ? ~db : (db | 0x80000000u);
    }
    const int cnt = valid ? r.count() : 0;
    // the Gaussian's block-row table once, for every lane with tiles (rectangles beyond 64 x 64 pixels
    // get none: their tiles fall back to block_mask16)
    if (__builtin_amdgcn_ballot_w64(cnt > 0) != 0ull) {
        if (cnt > 0) w = block_rows_table(g.x, g.y, g.A, g.B, g.C, g.smax, g.rx, g.ry);
    }
    // small rectangles: the lane walks its own tiles
    if (__builtin_amdgcn_ballot_w64(cnt > 0 && cnt <= kLaneTiles) != 0ull) {
        if (cnt > 0 && cnt <= kLaneTiles) {
            for (int ty = r.ty0; ty < r.ty1; ty++)
                for (int tx = r.tx0; tx < r.tx1; tx++) {
                    const uint32_t m = w.base != kNoRowTable
                                           ? mask_from_rows(w, tx, ty)
                                           : block_mask16(g.x, g.y, g.A, g.B, g.C, g.smax, g.rx, g.ry,
                                                          tx * GS_TILE, ty * GS_TILE);
                    emit(ty * tiles_x + tx, db, n, m);
                }
        }
    }
    // a rectangle with more than kLaneTiles tiles is walked by the whole wave (one lane looping over
    // thousands of tiles would stall the other 63): every lane takes tiles of the broadcast record and
    // assembles its mask from the broadcast row table
    uint64_t big = __builtin_amdgcn_ballot_w64(cnt > kLaneTiles);
    const int lane = threadIdx.x & 63;
    while (big) {
        const int src = __builtin_ctzll(big);
        big &= big - 1;
#define GS_BCAST_I(v) __builtin_amdgcn_readlane((int)(v), src)
#define GS_BCAST_F(v) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src))
        const int tx0 = GS_BCAST_I(r.tx0), tx1 = GS_BCAST_I(r.tx1);
        const int ty0 = GS_BCAST_I(r.ty0), ty1 = GS_BCAST_I(r.ty1);
        const uint32_t d = (uint32_t)GS_BCAST_I(db);
        const int id = GS_BCAST_I(n);
        RowTable bw;
        bw.base = (uint32_t)GS_BCAST_I(w.base);
        const int wd = tx1 - tx0, total = wd * (ty1 - ty0);
        if (bw.base != kNoRowTable) {
            bw.r0 = (uint32_t)GS_BCAST_I(w.r0); bw.r1 = (uint32_t)GS_BCAST_I(w.r1);
            bw.r2 = (uint32_t)GS_BCAST_I(w.r2); bw.r3 = (uint32_t)GS_BCAST_I(w.r3);
            for (int i = lane; i < total; i += 64) {
                const int tx = tx0 + i % wd, ty = ty0 + i / wd;
                emit(ty * tiles_x + tx, d, id, mask_from_rows(bw, tx, ty));
            }
            continue;
        }
        const float bx = GS_BCAST_F(g.x), by = GS_BCAST_F(g.y), bA = GS_BCAST_F(g.A);
        const float bB = GS_BCAST_F(g.B), bC = GS_BCAST_F(g.C);
        const uint32_t bs = (uint32_t)GS_BCAST_I(g.smax), brx = (uint32_t)GS_BCAST_I(g.rx);
        const uint32_t bry = (uint32_t)GS_BCAST_I(g.ry);
#undef GS_BCAST_I
#undef GS_BCAST_F
        for (int i = lane; i < total; i += 64) {
            const int tx = tx0 + i % wd, ty = ty0 + i / wd;
            emit(ty * tiles_x + tx, d, id,
                 block_mask16(bx, by, bA, bB, bC, bs, brx, bry, tx * GS_TILE, ty * GS_TILE));
        }
    }
}

__global__ void __launch_bounds__(256)
k_scatter_global(int N, int tiles_x, int32_t capacity, const float4 *__restrict__ packed,
                 const float *__restrict__ depths, const int2 *__restrict__ bins,
                 int32_t *__restrict__ fill, uint4 *__restrict__ keys) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    scatter_tiles(scatter_request(packed, depths, n, n < N), n, tiles_x, [&](int tile, uint32_t d, int g, uint32_t m) {
        const int pos = bins[tile].x + atomicAdd(&fill[tile], 1);
        if (pos < capacity) keys[pos] = make_uint4(d, (uint32_t)g, m, 0u);
    });
}

// LDS-privatised variant: the workgroup's cursors start at the tile's segment start plus the offset
// k_count_tiles reserved for this workgroup (wg_base), and every intersection takes its slot from
// the LDS cursor.  Slot order inside a tile is arbitrary — the per-tile sort fixes it.  Must be
// launched with k_count_tiles' grid and block size: the chunk -> workgroup assignment is the same.
__global__ void __launch_bounds__(kPersistentThreads)
k_scatter(int N, int tiles, int tiles_x, int32_t capacity, const float4 *__restrict__ packed,
          const float *__restrict__ depths, const int2 *__restrict__ bins,
          const int32_t *__restrict__ wg_base, uint4 *__restrict__ keys) {
    extern __shared__ int32_t h[];
    const int32_t *my_base = wg_base + (size_t)blockIdx.x * tiles;
    // 64-Gaussian chunks dealt out wave by wave ACROSS the workgroups (chunk = wave-major), so that a
    // few thousand Gaussians with huge rectangles still keep every CU's waves busy; the records of the
    // next chunk are in flight while the current one is scattered (see k_count_tiles)
    const int stride = (blockDim.x >> 6) * gridDim.x, lane = threadIdx.x & 63;
    int chunk = (threadIdx.x >> 6) * gridDim.x + blockIdx.x;
    ScatterIn in = scatter_request(packed, depths, chunk * 64 + lane, (int64_t)chunk * 64 + lane < N);
    {   // cursors: eight pairs of loads in flight per thread
        const int step = (int)blockDim.x;
        for (int t0 = threadIdx.x; t0 < tiles; t0 += 8 * step) {
            int32_t a[8], b[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int t = t0 + j * step;
                a[j] = t < tiles ? bins[t].x : 0;
                b[j] = t < tiles ? my_base[t] : 0;
            }
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (t0 + j * step < tiles) h[t0 + j * step] = a[j] + b[j];
        }
    }
    __syncthreads();
    for (; (int64_t)chunk * 64 < N; chunk += stride) {
        const int n = chunk * 64 + lane;
        const int64_t nn = (int64_t)n + (int64_t)stride * 64;   // (may not fit an int beyond the last chunk)
        const ScatterIn next = scatter_request(packed, depths, (int)(nn < N ? nn : 0), nn < N);
        scatter_tiles(in, n, tiles_x, [&](int tile, uint32_t d, int g, uint32_t m) {
            const int pos = atomicAdd(&h[tile], 1);
            if (pos < capacity) keys[pos] = make_uint4(d, (uint32_t)g, m, 0u);
        });
        in = next;
    }
}

// k_scatter with the scan folded in (round 6): the launch of k_scan_tiles_fast — one workgroup between two
// chip-filling launches, 10 us at C2 and 38 us at 4K — disappears.  Every scattering workgroup scans the tile
// counters ITSELF (32 KB at 1080p, from L2: its cursors need the segment starts anyway, which it used to read from
// tile_bins), and ONE EXTRA workgroup (the last of the grid) does what the scan kernel did for everybody else —
// tile_bins, {M, longest list}, the tile order — beside the scatter instead of in front of it.  Grid = the count
// kernel's + 1; only for images whose counters fit in LDS.
__global__ void __launch_bounds__(kPersistentThreads)
k_scatter_scan(int N, int tiles, int tiles_x, int32_t capacity, int order_mult,
               const float4 *__restrict__ packed, const float *__restrict__ depths,
               const int32_t *__restrict__ counts, const int32_t *__restrict__ wg_base,
               int2 *__restrict__ bins, int32_t *__restrict__ total_dev, int32_t *__restrict__ total_host,
               int32_t *__restrict__ order, uint4 *__restrict__ keys) {
    extern __shared__ int32_t h[];
    const int blocks = (int)gridDim.x - 1;
    if ((int)blockIdx.x == blocks) {
        scan_tiles_fast_body(tiles, order_mult, counts, bins, total_dev, total_host, order, h);
        return;
    }
    __shared__ int32_t ws[17];
    const int32_t *my_base = wg_base + (size_t)blockIdx.x * tiles;
    const int stride = (blockDim.x >> 6) * blocks, lane = threadIdx.x & 63;
    int chunk = (threadIdx.x >> 6) * blocks + blockIdx.x;
    ScatterIn in = scatter_request(packed, depths, chunk * 64 + lane, (int64_t)chunk * 64 + lane < N);
    {   // cursors = exclusive scan of the counters + this workgroup's offset inside every segment.  Coalesced:
        // wave w scans a contiguous run of the tiles 64 at a time (DPP wave scan + a carry), the sixteen run
        // totals go through LDS.  (Per-thread slices — 32 consecutive counters per thread at 4K, a cache line per
        // lane and load — made this preamble cost 42 us at C3.)
        const int wave = threadIdx.x >> 6;
        const int run_len = ((tiles + 15) / 16 + 63) / 64 * 64;
        const int lo = wave * run_len, hi = min(lo + run_len, tiles);
        // (eight loads in flight at a time — a wave's whole run at 1080p: one after the other they were eight round
        // trips in front of the first record, twice: the counters, then this workgroup's offsets)
        int32_t carry = 0;
        for (int i0 = lo; i0 < hi; i0 += 8 * 64) {
            int32_t c[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int i = i0 + j * 64 + lane;
                c[j] = i < hi ? counts[i] : 0;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int i = i0 + j * 64 + lane;
                const int32_t incl = wave_inclusive_scan_i(c[j]);
                if (i < hi) h[i] = carry + incl - c[j];
                carry += __builtin_amdgcn_readlane(incl, 63);
            }
        }
        if (lane == 0) ws[wave] = carry;
        // this workgroup's offsets: requested before the barrier, added behind it
        int32_t mb[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int i = lo + j * 64 + lane;
            mb[j] = i < hi ? my_base[i] : 0;
        }
        __syncthreads();
        int32_t off = 0;
        for (int w = 0; w < wave; w++) off += ws[w];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int i = lo + j * 64 + lane;
            if (i < hi) h[i] += off + mb[j];
        }
        for (int i0 = lo + 8 * 64; i0 < hi; i0 += 8 * 64) {   // (runs beyond 512 tiles per wave: 4K)
            int32_t m2[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int i = i0 + j * 64 + lane;
                m2[j] = i < hi ? my_base[i] : 0;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int i = i0 + j * 64 + lane;
                if (i < hi) h[i] += off + m2[j];
            }
        }
    }
    __syncthreads();
    for (; (int64_t)chunk * 64 < N; chunk += stride) {
        const int n = chunk * 64 + lane;
        const int64_t nn = (int64_t)n + (int64_t)stride * 64;
        const ScatterIn next = scatter_request(packed, depths, (int)(nn < N ? nn : 0), nn < N);
        scatter_tiles(in, n, tiles_x, [&](int tile, uint32_t d, int g, uint32_t m) {
            const int pos = atomicAdd(&h[tile], 1);
            if (pos < capacity) keys[pos] = make_uint4(d, (uint32_t)g, m, 0u);
        });
        in = next;
    }
}

// ---- coverage mask of one sorted list entry (gathers the record; see block_mask16) -----------------
// (for lists sorted elsewhere: gs_block_masks)
__device__ __forceinline__ uint16_t entry_mask(const float4 *__restrict__ packed, int32_t id, int tx0,
                                               int ty0) {
    const size_t g = (size_t)id;
    const float4 p0 = packed[3 * g + 0], p1 = packed[3 * g + 1];
    const uint32_t ry = __float_as_uint(reinterpret_cast<const float *>(packed)[12 * g + 11]);
    return (uint16_t)block_mask16(p0.x, p0.y, p0.z, p0.w, p1.x, __float_as_uint(p1.z),
                                  __float_as_uint(p1.w), ry, tx0, ty0);
}

// ---- 4. per-tile sort --------------------------------------------------------------------------
// The sort key of a record is (depth key << 32 | id); the coverage mask travels with it.
__device__ __forceinline__ uint64_t rec_key(const uint4 r) { return ((uint64_t)r.x << 32) | r.y; }

// Bitonic network over P = next power of two >= n keys, in the all-ascending formulation: each
// merge phase starts with a "flip" step (i <-> mirror position inside the 2k block) followed by
// half-cleaners, and every compare-exchange moves the smaller key to the lower index.  Padding
// keys (all-ones) at positions >= n therefore never move, so they can be virtual.
template <typename Mem>
__device__ __forceinline__ void bitonic_sort(Mem &m, int P, int tid, int nthreads) {
    for (int k = 2; k <= P; k <<= 1) {
        const int half = k >> 1;
        for (int i = tid; i < (P >> 1); i += nthreads) {
            const int lo = ((i & ~(half - 1)) << 1) | (i & (half - 1));
            const int hi = lo ^ (k - 1);
            if (m.key(lo) > m.key(hi)) m.swap(lo, hi);
        }
        __syncthreads();
        for (int j = k >> 2; j > 0; j >>= 1) {
            for (int i = tid; i < (P >> 1); i += nthreads) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int hi = lo | j;
                if (m.key(lo) > m.key(hi)) m.swap(lo, hi);
            }
            __syncthreads();
        }
    }
}

// keys + masks in LDS; positions >= n read as the padding key and are never swapped (a padding key
// is never smaller than anything)
struct LdsRecs {
    uint64_t *k;
    uint16_t *m;
    int n;
    __device__ __forceinline__ uint64_t key(int i) const { return i < n ? k[i] : ~0ull; }
    __device__ __forceinline__ void swap(int i, int j) {
        const uint64_t a = k[i];
        k[i] = k[j];
        k[j] = a;
        const uint16_t b = m[i];
        m[i] = m[j];
        m[j] = b;
    }
};
// the records in place in global memory
struct GlobalRecs {
    uint4 *p;
    int n;
    __device__ __forceinline__ uint64_t key(int i) const { return i < n ? rec_key(p[i]) : ~0ull; }
    __device__ __forceinline__ void swap(int i, int j) {
        const uint4 a = p[i];
        p[i] = p[j];
        p[j] = a;
    }
};

// a segment too long for the on-chip paths: network in place, then ids and masks out
template <int NT>
__device__ __forceinline__ void sort_in_global(uint4 *__restrict__ recs, int n, int tid,
                                               int32_t *__restrict__ ids_out,
                                               uint16_t *__restrict__ masks_out) {
    int P = 2;
    while (P < n) P <<= 1;
    GlobalRecs m{recs, n};
    bitonic_sort(m, P, tid, NT);
    for (int i = tid; i < n; i += NT) {
        const uint4 r = recs[i];
        ids_out[i] = (int32_t)r.y;
        masks_out[i] = (uint16_t)r.z;
    }
}

// ---- 4b. bucket sort ---------------------------------------------------------------------------
// The keys of one tile are (nearly) uniformly spread between the tile's nearest and farthest
// Gaussian, so a counting sort on the leading bits of (depth key - min) puts almost every key in
// its own bucket: O(n) LDS work instead of the bitonic network's O(n log^2 n).
//   1. min / max of the 32-bit depth keys (wave reductions, combined through LDS);
//   2. bucket = (depth key - min) >> shift  (shift chosen so that bucket < B; monotone in the key);
//      LDS histogram, exclusive scan, scatter of the 64-bit keys (+ masks) through LDS cursors;
//   3. every bucket holding more than one key is put in order by a single lane (insertion sort on
//      the full (depth, id) key).  If some bucket is longer than kMaxBucket keys (many equal
//      depths — pathological) the whole segment is re-sorted with the bitonic network instead.
// CAP: segment capacity in keys; B: buckets; NT: threads.  LDS: 10*CAP + 4*B (+ a few words).
constexpr int kMaxBucket = 24;
// Bucket of a depth key.  Rounds 1 - 5: (key - min) >> shift — linear in the float's BITS, i.e. logarithmic in the depth:
// depths spread evenly between 1 and 100 put a third of the keys into the twelfth of the buckets that the top binade
// owns, and the lanes that own those buckets finish them alone (insertion sort, a chain of LDS round trips).  Round 6
// (GS_SORT_LINEAR): linear in the DEPTH — bucket = (int)((depth - nearest) * B / (farthest - nearest)), every step of
// which is monotone in the depth (rounded subtraction, multiplication by a positive constant, truncation), so the
// buckets are still ordered; whenever nearest / farthest are not finite or too close for the scale the bit-linear
// map stays.  The sorted lists are the same either way (ties and in-bucket order by the full (depth key, id) key).
// Measured (profiles/r06/exp_sort_linear/): the long class on the hot-spot scene 27.3 -> 15.8 us, at 4K 47.6 -> 23.4 us.
#ifndef GS_SORT_LINEAR
#define GS_SORT_LINEAR 1
#endif
struct BucketMap {
    uint32_t mn;
    int shift;
    float fmin, scale;   // scale > 0: the linear map
    int last;
};
__device__ __forceinline__ float key_depth(uint32_t k) {   // inverse of the order-preserving image of a float
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
template <bool LIN>
__device__ __forceinline__ BucketMap bucket_map(uint32_t mn, uint32_t mx, int B) {
    BucketMap m;
    m.mn = mn;
    m.last = B - 1;
    const uint32_t span = mx - mn;
    int shift = 0;
    while ((span >> shift) >= (uint32_t)B) shift++;
    m.shift = shift;
    m.fmin = 0.0f;
    m.scale = 0.0f;
    if (LIN && GS_SORT_LINEAR) {
        const float lo = key_depth(mn), hi = key_depth(mx);
        const float width = hi - lo;
        if (mx > mn && fabsf(lo) < 3.0e38f && fabsf(hi) < 3.0e38f && width > 1.0e-30f && width < 3.0e38f) {
            const float sc = (float)B / width;
            if (sc < 3.0e38f) {
                m.fmin = lo;
                m.scale = sc;
            }
        }
    }
    return m;
}
template <bool LIN>
__device__ __forceinline__ int bucket_of(const BucketMap &m, uint32_t key) {
    if (LIN && GS_SORT_LINEAR) {
        if (m.scale > 0.0f) return min((int)((key_depth(key) - m.fmin) * m.scale), m.last);
    }
    return (int)((key - m.mn) >> m.shift);
}
// buckets of the 512-key wave class: with the masks out of LDS (wave_bucket_sort) 256 buckets make 4096 + 1024 = 5120
// bytes per wave — 32 waves per CU, the whole 8160-wave launch of a 1080p frame resident at once; 512 buckets: 26
#ifndef GS_SHORT_BUCKETS
#define GS_SHORT_BUCKETS 256
#endif
constexpr int kShortBuckets = GS_SHORT_BUCKETS;
// buckets of the 1024-key wave class (8192 bytes of keys + 4 per bucket).  Frames of MANY tiles, where the class runs in
// rounds (4K: ten) and residency is what counts: 512 buckets, linear in the depth — 10 KiB, 16 waves per CU instead of
// 13: C3 235.9 -> 224.3 us, with the linear map (which lost 5 % at 1024 buckets) 211.6 us.  Frames of few tiles, where a
// wave is alone on its SIMD and waits for its slowest lane's insertion sorts: 1024 buckets, bit-linear as before
// (end-to-end training on 384 x 288: 3450 / 3684 / 3430 it/s against 3381 / 3594 / 3425 with 512).
constexpr int kMidBuckets = 512, kMidBucketsFewTiles = 1024;
constexpr int kMidBucketsTiles = 4096;   // frames beyond: kMidBuckets
// buckets of the 8192-key class (k_bucket_sort_tiles<8192, B, 1024>)
#ifndef GS_LONG_BUCKETS
#define GS_LONG_BUCKETS 8192
#endif

template <int NT>
__device__ __forceinline__ void block_minmax(uint32_t &mn, uint32_t &mx, uint32_t *scratch) {
    for (int off = 32; off > 0; off >>= 1) {
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, off));
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
    }
    if (NT > 64) {
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { scratch[2 * w] = mn; scratch[2 * w + 1] = mx; }
        __syncthreads();
        for (int i = 0; i < NT / 64; i++) { mn = min(mn, scratch[2 * i]); mx = max(mx, scratch[2 * i + 1]); }
        __syncthreads();
    }
}

// insertion sort of out[s0, e) / outm[s0, e) by key (a bucket of at most kMaxBucket keys)
__device__ __forceinline__ void insertion_sort_bucket(uint64_t *out, uint16_t *outm, int s0, int e) {
    for (int i = s0 + 1; i < e; i++) {
        const uint64_t k = out[i];
        const uint16_t km = outm[i];
        int q = i - 1;
        while (q >= s0 && out[q] > k) {
            out[q + 1] = out[q];
            outm[q + 1] = outm[q];
            q--;
        }
        out[q + 1] = k;
        outm[q + 1] = km;
    }
}

// keys only (the wave sorts: the masks stay in registers until the final positions are known)
__device__ __forceinline__ void insertion_sort_keys(uint64_t *out, int s0, int e) {
    for (int i = s0 + 1; i < e; i++) {
        const uint64_t k = out[i];
        int q = i - 1;
        while (q >= s0 && out[q] > k) {
            out[q + 1] = out[q];
            q--;
        }
        out[q + 1] = k;
    }
}
struct LdsKeys {
    uint64_t *k;
    int n;
    __device__ __forceinline__ uint64_t key(int i) const { return i < n ? k[i] : ~0ull; }
    __device__ __forceinline__ void swap(int i, int j) {
        const uint64_t a = k[i];
        k[i] = k[j];
        k[j] = a;
    }
};

template <int CAP, int B, int NT>
__device__ __forceinline__ void bucket_sort_one_tile(const int2 range, int lo_n, int hi_n, int32_t capacity,
                                                     uint4 *__restrict__ keys, int32_t *__restrict__ ids_sorted,
                                                     uint16_t *__restrict__ masks, unsigned char *smem) {
    uint64_t *out = reinterpret_cast<uint64_t *>(smem);           // CAP keys
    int32_t *cnt = reinterpret_cast<int32_t *>(out + CAP);         // B counters / cursors
    uint32_t *scratch = reinterpret_cast<uint32_t *>(cnt + B);     // 2 * NT/64 + NT/64 + 1 words (64 reserved)
    uint16_t *outm = reinterpret_cast<uint16_t *>(scratch + 64);   // CAP masks
    const int start = range.x;
    const int n = min(range.y, capacity) - start;
    if (n <= lo_n) return;
    const int tid = threadIdx.x;
    if (n > hi_n) {  // longer than the LDS capacity: bitonic network in place in global memory
        sort_in_global<NT>(keys + start, n, tid, ids_sorted + start, masks + start);
        return;
    }
    const uint4 *src = keys + start;
    // the records ONCE into registers (CAP / NT per thread, all loads in flight together): the three passes below
    // used to read them from global memory one after the other — three round trips of a lone workgroup
    static_assert(CAP % NT == 0, "k_bucket_sort_tiles: CAP must be a multiple of the block size");
    constexpr int PL = CAP / NT;
    uint4 rr[PL];
#pragma unroll
    for (int j = 0; j < PL; j++) {
        const int i = j * NT + tid;
        rr[j] = i < n ? src[i] : make_uint4(0u, 0u, 0u, 0u);
    }
    // 1. range of the depth keys
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
    for (int j = 0; j < PL; j++)
        if (j * NT + tid < n) {
            mn = min(mn, rr[j].x);
            mx = max(mx, rr[j].x);
        }
    block_minmax<NT>(mn, mx, scratch);
    const BucketMap bm = bucket_map<true>(mn, mx, B);
    // 2. histogram
    for (int i = tid; i < B; i += NT) cnt[i] = 0;
    __syncthreads();
    int bk[PL];   // (the bucket of each record, kept for the scatter pass)
#pragma unroll
    for (int j = 0; j < PL; j++) {
        bk[j] = bucket_of<true>(bm, rr[j].x);
        if (j * NT + tid < n) atomicAdd(&cnt[bk[j]], 1);
    }
    __syncthreads();
    //    exclusive scan of the B counters: each thread owns B/NT consecutive ones
    constexpr int PER = B / NT;
    int32_t local[PER];
    int32_t sum = 0, longest = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        local[j] = cnt[tid * PER + j];
        longest = max(longest, local[j]);
        sum += local[j];
    }
    int32_t incl = sum;  // inclusive scan across the wave
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t v = __shfl_up(incl, off);
        if ((tid & 63) >= off) incl += v;
    }
    int32_t wave_base = 0;
    if (NT > 64) {
        int32_t *wsum = reinterpret_cast<int32_t *>(scratch + 2 * (NT / 64));
        if ((tid & 63) == 63) wsum[tid >> 6] = incl;
        __syncthreads();
        for (int w = 0; w < (tid >> 6); w++) wave_base += wsum[w];
    }
    int32_t run = wave_base + incl - sum;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        cnt[tid * PER + j] = run;  // becomes the bucket's cursor
        run += local[j];
    }
    // pathological segment?  (wave-level OR, then across waves through LDS)
    uint32_t *flag = scratch + 3 * (NT / 64);
    if (tid == 0) *flag = 0u;
    __syncthreads();
    if (longest > kMaxBucket) atomicOr(flag, 1u);
    // 3. scatter into bucket order
#pragma unroll
    for (int j = 0; j < PL; j++)
        if (j * NT + tid < n) {
            const int pos = atomicAdd(&cnt[bk[j]], 1);
            out[pos] = rec_key(rr[j]);
            outm[pos] = (uint16_t)rr[j].z;
        }
    __syncthreads();
    if (*flag) {
        // many equal / clustered depths: bitonic network on the LDS copy (virtual padding)
        int P = 2;
        while (P < n) P <<= 1;
        LdsRecs m{out, outm, n};
        bitonic_sort(m, P, tid, NT);
    } else {
        // order inside buckets: cursor[b] is now the END of bucket b, the end of b-1 its start
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int e = cnt[tid * PER + j];
            insertion_sort_bucket(out, outm, e - local[j], e);
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += NT) {
        ids_sorted[start + i] = (int32_t)(uint32_t)out[i];
        masks[start + i] = outm[i];
    }
}

// One workgroup per tile (order == nullptr: grid = tiles; a tile outside (lo_n, ...] costs its workgroup's dispatch:
// 8160 x 16 waves launched to find nine lists beyond 1024 entries on the hot-spot scene, 3 of the launch's 33 us there,
// 39 us for the 256-thread class of round 2 at 4K), or — round 6 — a SMALL grid that walks the head of `tile_order`: the scan leaves the tiles sorted by descending class
// (length >> order_shift(longest)), so the lists of this size class are its first entries and a workgroup stops at
// the first tile of a lower class than lo_n + 1's (tiles of that class itself may be on either side of lo_n: skipped).
// The class and the stop test use the UNCLAMPED length, as the order does; the shift follows from order[0], which is
// in the longest list's class.
template <int CAP, int B, int NT>
__global__ void __launch_bounds__(NT)
k_bucket_sort_tiles(int lo_n, int hi_n, int32_t capacity, int tiles, const int32_t *__restrict__ order,
                    const int2 *__restrict__ bins, uint4 *__restrict__ keys, int32_t *__restrict__ ids_sorted,
                    uint16_t *__restrict__ masks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (!order) {
        bucket_sort_one_tile<CAP, B, NT>(bins[blockIdx.x], lo_n, hi_n, capacity, keys, ids_sorted, masks, smem);
        return;
    }
    // (On a frame whose id list is too small the short class clamps overflowing ranges to {min(start, capacity),
    // capacity} — behind this launch today, but nothing here relies on that: a range that ends at `capacity` may be a
    // clamped one, its length then says nothing about its class.  Such a tile never stops the walk, and if it is
    // order[0] nothing does: the frame is repeated anyway, what counts is that no list is left unsorted.)
    const int2 first = bins[order[0]];
    const bool can_stop = first.y != capacity;
    const int shift = order_shift(first.y - first.x);
    const int stop_class = (lo_n + 1) >> shift;
    for (int i = blockIdx.x; i < tiles; i += gridDim.x) {
        const int2 range = bins[order[i]];
        if (can_stop && range.y != capacity && ((range.y - range.x) >> shift) < stop_class) break;
        bucket_sort_one_tile<CAP, B, NT>(range, lo_n, hi_n, capacity, keys, ids_sorted, masks, smem);
        __syncthreads();   // the LDS buffers are the next tile's
    }
}

// Single-wave variant for the common short segments (n <= 64*PL keys): the records are loaded ONCE
// into registers (PL per lane), bucket b is owned by lane b % 64 (conflict-free LDS access), the
// exclusive scan is B/64 DPP wave scans.  LDS: 10*64*PL + 4*B bytes (7 KiB for PL = 8, B = 512),
// so that many tiles are resident per CU and their global-memory round trips overlap.
// The sort of one wave's n <= 64 * PL records held in registers (kk: (depth key << 32 | id), mm: the 16-bit
// masks two to a register, slot j * 64 + lane; mn / mx: this lane's range of the depth keys) through the
// wave's LDS buffers, written to ids_dst / masks_dst[0 .. n).
template <int PL, int B>
__device__ __forceinline__ void wave_bucket_sort(const uint64_t (&kk)[PL], const uint32_t (&mm)[PL / 2],
                                                 uint32_t mn, uint32_t mx, int n, int lane, uint64_t *out,
                                                 int32_t *cnt, int32_t *__restrict__ ids_dst,
                                                 uint16_t *__restrict__ masks_dst) {
    constexpr int PER = B / 64;
    for (int off = 32; off > 0; off >>= 1) {
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, off));
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
    }
    // (the linear map: the 512-key class at C2 27.0 -> 24.7 us.  The 1024-key class, ten rounds of waves at C3, measured
    // 5 % SLOWER with it at 1024 buckets (243 -> 255 us) and 6 % faster at 512 (224.3 -> 211.6): kMidBuckets)
    constexpr bool LIN = PL <= 8 || B < 1024;
    const BucketMap bm = bucket_map<LIN>(mn, mx, B);
#pragma unroll
    for (int j = 0; j < PER; j++) cnt[j * 64 + lane] = 0;
    __syncthreads();
    int bk[PL];   // (the bucket of each record, kept for the scatter pass)
#pragma unroll
    for (int j = 0; j < PL; j++) {
        bk[j] = bucket_of<LIN>(bm, (uint32_t)(kk[j] >> 32));
        if (j * 64 + lane < n) atomicAdd(&cnt[bk[j]], 1);
    }
    __syncthreads();
    // exclusive scan, bucket b = j*64 + lane
    int32_t local[PER];
    int32_t base = 0, longest = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        local[j] = cnt[j * 64 + lane];
        longest = max(longest, local[j]);
        const int32_t incl = wave_inclusive_scan_i(local[j]);
        cnt[j * 64 + lane] = base + incl - local[j];
        base += __builtin_amdgcn_readlane(incl, 63);
    }
    const bool pathological = __builtin_amdgcn_ballot_w64(longest > kMaxBucket) != 0ull;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PL; j++)
        if (j * 64 + lane < n) {
            const int pos = atomicAdd(&cnt[bk[j]], 1);
            out[pos] = kk[j];
        }
    __syncthreads();
    if (pathological) {
        int P = 2;
        while (P < n) P <<= 1;  // P <= CAP because n <= CAP and CAP is a power of two
        LdsKeys m{out, n};
        bitonic_sort(m, P, lane, 64);
    } else {
#pragma unroll
        for (int j = 0; j < PER; j++) {
            if (local[j] < 2) continue;
            const int e = cnt[j * 64 + lane];  // cursor == end of the bucket
            insertion_sort_keys(out, e - local[j], e);
        }
        __syncthreads();
    }
    // Only the KEYS went through LDS (round 6: the 2 bytes of mask per slot were the difference between 22 and 26 ... 32
    // waves per CU — the 512-key class of a 1080p frame is 8160 waves).  Every lane now looks up where each of ITS
    // records ended: inside its bucket (the cursors are the buckets' ends, a bucket starts where the one before it
    // ends; keys are unique), and writes id and mask there itself.
#pragma unroll
    for (int j = 0; j < PL; j++) {
        if (j * 64 + lane >= n) continue;
        const int b = bk[j];
        int lo = b > 0 ? cnt[b - 1] : 0, hi = cnt[b];
        while (hi - lo > 1) {   // (a bucket of one key: no read at all)
            const int mid = (lo + hi) >> 1;
            if (out[mid] <= kk[j]) lo = mid;
            else hi = mid;
        }
        ids_dst[lo] = (int32_t)(uint32_t)kk[j];
        masks_dst[lo] = (uint16_t)(mm[j >> 1] >> (16 * (j & 1)));
    }
}

template <int PL, int B>
__global__ void __launch_bounds__(64)
k_bucket_sort_wave(int lo_n, int hi_n, int32_t capacity, int clamp_bins, int take_longer,
                   int2 *__restrict__ bins, uint4 *__restrict__ keys,
                   int32_t *__restrict__ ids_sorted, uint16_t *__restrict__ masks) {
    constexpr int CAP = 64 * PL;
    __shared__ uint64_t out[CAP];
    __shared__ int32_t cnt[B];
    const int2 range = bins[blockIdx.x];
    const int start = range.x;
    const int n = min(range.y, capacity) - start;
    // capacity overflow (the caller sized the id list from a stale count): make the ranges safe to
    // walk — the compositing kernels then read inside the buffer; the caller detects the overflow
    // from the true total and repeats the call
    if (clamp_bins && range.y > capacity && threadIdx.x == 0)
        bins[blockIdx.x] = make_int2(min(start, capacity), capacity);
    if (n <= lo_n) return;
    const int lane = threadIdx.x;
    if (n > hi_n) {
        // a longer segment: normally another launch's job.  When the host skipped those launches
        // (the previous frame had no long list) this wave sorts it in place in global memory —
        // slow, but only ever hit on the frame where a list first outgrows this class.
        if (!take_longer) return;
        sort_in_global<64>(keys + start, n, lane, ids_sorted + start, masks + start);
        return;
    }
    const uint4 *src = keys + start;
    uint64_t kk[PL];
    uint32_t mm[PL / 2];   // the 16-bit masks, two to a register
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
    for (int j = 0; j < PL / 2; j++) mm[j] = 0u;
#pragma unroll
    for (int j = 0; j < PL; j++) {
        const int i = j * 64 + lane;
        uint4 r = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
        if (i < n) r = src[i];
        kk[j] = rec_key(r);
        mm[j >> 1] |= (r.z & 0xFFFFu) << (16 * (j & 1));
        if (i < n) {
            mn = min(mn, r.x);
            mx = max(mx, r.x);
        }
    }
    wave_bucket_sort<PL, B>(kk, mm, mn, mx, n, lane, out, cnt, ids_sorted + start, masks + start);
}

// ---- 5. coverage masks ------------------------------------------------------------------------
// One wave per tile: block_masks[i] = block_mask16 (gs_device.h) of list entry i against its tile —
// which of the tile's sixteen 4x4-pixel blocks the Gaussian's sigma_max ellipse (and rectangle) can
// reach.  The compositing kernels build their per-block walk lists from these bits instead of
// testing the bounding rectangle in every wave of the tile, and skip the gather of entries that miss
// their part of the tile.  2 B written per intersection; the record gather is served by L2 (the sort
// and the compositing kernels read the same lines).
__global__ void __launch_bounds__(64)
k_block_masks(int tiles_x, const int2 *__restrict__ bins, const int32_t *__restrict__ ids,
              const float4 *__restrict__ packed, uint16_t *__restrict__ masks) {
    const int tile = blockIdx.x;
    const int2 range = bins[tile];
    const int tx0 = (tile % tiles_x) * GS_TILE, ty0 = (tile / tiles_x) * GS_TILE;
    for (int i = range.x + (int)threadIdx.x; i < range.y; i += 64) {
        masks[i] = entry_mask(packed, ids[i], tx0, ty0);
    }
}

// ---- 6. strip binning (round 6) -------------------------------------------------------------------
// The same lists through a two-level partition, without a per-tile count pass over the Gaussians and
// without the 256 x tiles offset table:
//   a. k_cell_count     one lane per Gaussian counts, per STRIP (kStripTiles = 16 consecutive tiles of
//                       one tile row — "cell"), how many of its rectangles reach the strip (entries) and
//                       how many tiles of the strip they cover (intersections): 544 pairs of counters at
//                       1080p instead of 8160 tile counters, one 64-bit LDS atomic per (Gaussian, strip);
//   b. k_cell_scatter   every workgroup scans the strip counters itself (no scan launch), then one lane
//                       per Gaussian forms the Gaussian's block-row table ONCE and appends a 32-byte
//                       record {depth key, id, rectangle | row table} to each strip it reaches: 1.5
//                       records per Gaussian at C2 (2.15 tile intersections), 2.1 at C3 (3.8), each a
//                       full 32-byte sector;
//   c. k_strip_scatter  one workgroup per strip reads the strip's records ONCE, coalesced; counts its
//                       sixteen tiles with ballots, scans them (tile_bins: the strip's segment starts at
//                       the scanned intersection count of the strips before it — strips of a row are
//                       sixteen consecutive tiles, so the list stays tile-major), and writes the 16-byte
//                       {depth key, id, mask} records of the per-tile sort into the tiles' segments: a
//                       68-KiB window of memory per strip, written by one workgroup in one go, which the
//                       L2 merges into full lines (the Gaussian-major scatter opened a line per store);
//   d. the per-tile bucket sorts of section 4, unchanged.
// A rectangle is clipped to the strip by tile index alone, so the records carry no strip coordinates.
constexpr int kStripTiles = 16;
constexpr int kStripShift = 4;
// strips whose counters (8 B) and cursors fit the LDS of the persistent kernels comfortably: 8192 strips =
// a 7680 x 4320 frame; larger images take the tile-level path above
constexpr int kMaxCells = 8192;

// Visit every strip of every Gaussian of this wave's 64-lane slice: f(cell, payload lane).  Rectangles
// reaching up to kLaneTiles strips are walked by their own lane (src = the lane itself), larger ones by
// the whole wave (src = the owning lane: the caller broadcasts what it needs with readlane).
template <typename F>
__device__ __forceinline__ void for_each_cell(const TileRect &r, bool valid, int cells_x, F f) {
    const int lane = threadIdx.x & 63;
    const int cx0 = r.tx0 >> kStripShift;
    const int cw = (valid && r.count() > 0) ? (((r.tx1 - 1) >> kStripShift) - cx0 + 1) : 0;
    const int cnt = cw * (r.ty1 - r.ty0);
    if (cnt > 0 && cnt <= kLaneTiles) {
        for (int ty = r.ty0; ty < r.ty1; ty++)
            for (int c = 0; c < cw; c++) f(ty * cells_x + cx0 + c, lane, false);
    }
    uint64_t big = __builtin_amdgcn_ballot_w64(cnt > kLaneTiles);
    while (big) {
        const int src = __builtin_ctzll(big);
        big &= big - 1;
        const int bcx0 = __builtin_amdgcn_readlane(cx0, src), bcw = __builtin_amdgcn_readlane(cw, src);
        const int bty0 = __builtin_amdgcn_readlane(r.ty0, src);
        const int total = __builtin_amdgcn_readlane(cnt, src);
        for (int i = lane; i < total; i += 64) f((bty0 + i / bcw) * cells_x + bcx0 + i % bcw, src, true);
    }
}
// tiles of the rectangle [tx0, tx1) inside strip column cx
__device__ __forceinline__ int tiles_in_strip(int tx0, int tx1, int cx) {
    return min(tx1, (cx + 1) << kStripShift) - max(tx0, cx << kStripShift);
}

// a. counters: low word = records (entries), high word = tile intersections.  wg_base[workgroup][cell] =
// where this workgroup's records start inside the strip's list (the value the flush's returning atomic
// hands back), exactly as k_count_tiles does per tile.
__global__ void __launch_bounds__(kPersistentThreads)
k_cell_count(int N, int cells, int cells_x, const float4 *__restrict__ packed,
             unsigned long long *__restrict__ counts, int32_t *__restrict__ wg_base) {
    extern __shared__ unsigned long long hc[];
    const int stride = (blockDim.x >> 6) * gridDim.x, lane = threadIdx.x & 63;
    int chunk = (threadIdx.x >> 6) * gridDim.x + blockIdx.x;
    auto request = [&](int ch, uint32_t &rx, uint32_t &ry) {
        const int64_t n = (int64_t)ch * 64 + lane;
        const bool ok = n < N;
        rx = ok ? __float_as_uint(packed[3 * (size_t)n + 1].w) : 0u;
        ry = ok ? __float_as_uint(packed[3 * (size_t)n + 2].w) : 0u;
        return ok;
    };
    uint32_t rx, ry;
    bool valid = request(chunk, rx, ry);
    for (int t = threadIdx.x; t < cells; t += blockDim.x) hc[t] = 0ull;
    __syncthreads();
    for (; (int64_t)chunk * 64 < N; chunk += stride) {
        uint32_t nrx, nry;
        const bool nvalid = request(chunk + stride, nrx, nry);
        const TileRect r = tile_rect_of(rx, ry);
        for_each_cell(r, valid, cells_x, [&](int cell, int src, bool bcast) {
            const int tx0 = bcast ? __builtin_amdgcn_readlane(r.tx0, src) : r.tx0;
            const int tx1 = bcast ? __builtin_amdgcn_readlane(r.tx1, src) : r.tx1;
            const int nt = tiles_in_strip(tx0, tx1, cell % cells_x);
            atomicAdd(&hc[cell], 1ull | ((unsigned long long)nt << 32));
        });
        rx = nrx; ry = nry; valid = nvalid;
    }
    __syncthreads();
    int32_t *my_base = wg_base + (size_t)blockIdx.x * cells;
    const int step = (int)blockDim.x;
    for (int t0 = threadIdx.x; t0 < cells; t0 += 4 * step) {
        unsigned long long c[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; j++) c[j] = t0 + j * step < cells ? hc[t0 + j * step] : 0ull;
#pragma unroll
        for (int j = 0; j < 4; j++) b[j] = c[j] ? atomicAdd(&counts[t0 + j * step], c[j]) : 0ull;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (t0 + j * step < cells) my_base[t0 + j * step] = (int32_t)(uint32_t)b[j];
    }
}

// b. cell_bins[cell] = {first record, records, first intersection, intersections}.  Records are 32 bytes
// {depth key, id, rectangle x, rectangle y | row table r0..r3}, stored as two adjacent 16-byte halves: one
// full sector, and ONE open cache line per (workgroup, strip) — with the halves in two arrays the lines a
// workgroup keeps open (2 x strips x 128 B) no longer fitted the XCD's L2 next to its 31 neighbours'
// (k_cell_scatter 47 -> 32 us at C2).  filt[pos] repeats the rectangle's x range (4 B): the filter pass of
// k_tile_gather_sort reads that array alone.
struct StripRec {
    uint4 a, b;
};
__global__ void __launch_bounds__(kPersistentThreads)
k_cell_scatter(int N, int cells, int cells_x, int32_t capacity,
               const float4 *__restrict__ packed, const float *__restrict__ depths,
               const unsigned long long *__restrict__ counts, const int32_t *__restrict__ wg_base,
               int4 *__restrict__ cell_bins, int32_t *__restrict__ total_dev, int32_t *__restrict__ total_host,
               StripRec *__restrict__ recs, uint32_t *__restrict__ filt) {
    extern __shared__ int32_t h[];              // the workgroup's cursor of every strip
    __shared__ int32_t ws[17];
    const int t = threadIdx.x;
    const int per = (cells + 1023) / 1024;
    const int lo = min(t * per, cells), hi = min(lo + per, cells);
    const int32_t *my_base = wg_base + (size_t)blockIdx.x * cells;
    const int stride = (blockDim.x >> 6) * gridDim.x, lane = threadIdx.x & 63;
    int chunk = (threadIdx.x >> 6) * gridDim.x + blockIdx.x;
    ScatterIn in = scatter_request(packed, depths, chunk * 64 + lane, (int64_t)chunk * 64 + lane < N);
    {   // exclusive scan of the strip counters (records and intersections), by every workgroup for itself
        int32_t se = 0, si = 0;
        for (int i = lo; i < hi; i++) {
            const unsigned long long c = counts[i];
            se += (int32_t)(uint32_t)c;
            si += (int32_t)(uint32_t)(c >> 32);
        }
        int32_t re = block_scan_1024(se, ws) - se;
        const int32_t incl_i = block_scan_1024(si, ws);
        int32_t ri = incl_i - si;
        const int32_t total_i = ws[16];
        for (int i = lo; i < hi; i++) {
            const unsigned long long c = counts[i];
            const int32_t ce = (int32_t)(uint32_t)c, ci = (int32_t)(uint32_t)(c >> 32);
            h[i] = re + my_base[i];
            if (blockIdx.x == 0) cell_bins[i] = make_int4(re, ce, ri, ci);
            re += ce;
            ri += ci;
        }
        if (blockIdx.x == 0 && t == 1023) {
            *total_dev = total_i;
            if (total_host) total_host[0] = total_i;   // pinned, device-mapped
        }
    }
    __syncthreads();
    for (; (int64_t)chunk * 64 < N; chunk += stride) {
        const int n = chunk * 64 + lane;
        const int64_t nn = (int64_t)n + (int64_t)stride * 64;
        const ScatterIn next = scatter_request(packed, depths, (int)(nn < N ? nn : 0), nn < N);
        TileRect r = {0, 0, 0, 0};
        uint4 a = make_uint4(0u, 0u, 0u, 0u), b = make_uint4(0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu);
        if (in.valid) {
            a.z = __float_as_uint(in.p1.w);
            a.w = __float_as_uint(in.p2w);
            r = tile_rect_of(a.z, a.w);
            uint32_t db = __float_as_uint(in.depth);
            a.x = (db & 0x80000000u) ? ~db : (db | 0x80000000u);
            a.y = (uint32_t)n;
        }
        const bool any = in.valid && r.count() > 0;
        if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
            if (any) {
                const RowTable w = block_rows_table(in.p0.x, in.p0.y, in.p0.z, in.p0.w, in.p1.x,
                                                    __float_as_uint(in.p1.z), a.z, a.w);
                b = make_uint4(w.r0, w.r1, w.r2, w.r3);   // (base follows from the rectangle)
            }
        }
        for_each_cell(r, in.valid, cells_x, [&](int cell, int src, bool bcast) {
            uint4 ra = a, rb = b;
            if (bcast) {
#define GS_BC(v) (uint32_t) __builtin_amdgcn_readlane((int)(v), src)
                ra = make_uint4(GS_BC(a.x), GS_BC(a.y), GS_BC(a.z), GS_BC(a.w));
                rb = make_uint4(GS_BC(b.x), GS_BC(b.y), GS_BC(b.z), GS_BC(b.w));
#undef GS_BC
            }
            const int pos = atomicAdd(&h[cell], 1);
            if (pos < capacity) {
                recs[pos].a = ra;
                recs[pos].b = rb;
                if (filt) filt[pos] = ra.z;   // the x range alone: what k_tile_gather_sort's filter pass reads
            }
        });
        in = next;
    }
}

// the row table a strip record carries: its base follows from the rectangle, as block_rows_table sets it
__device__ __forceinline__ RowTable row_table_of(const uint4 a, const uint4 b) {
    const int x0 = (int)(a.z & 0xFFFFu), x1 = (int)(a.z >> 16) - 1;
    const int y0 = (int)(a.w & 0xFFFFu), y1 = (int)(a.w >> 16) - 1;
    RowTable w = {kNoRowTable, b.x, b.y, b.z, b.w};
    if (x1 < x0 || y1 < y0) return w;
    const int br0 = y0 >> 2, br1 = y1 >> 2, bc0 = x0 >> 2, bc1 = x1 >> 2;
    if (br1 - br0 >= kRowTableRows || bc1 - bc0 >= 16) return w;
    w.base = (uint32_t)br0 | ((uint32_t)bc0 << 16);
    return w;
}

// c. one workgroup per strip.
__global__ void __launch_bounds__(kPersistentThreads)
k_strip_scatter(int cells_x, int tiles_x, int32_t capacity, const int4 *__restrict__ cell_bins,
                const StripRec *__restrict__ recs, const float4 *__restrict__ packed,
                uint4 *__restrict__ keys, int2 *__restrict__ bins, int32_t *__restrict__ longest) {
    __shared__ int32_t cnt[kStripTiles], cur[kStripTiles];
    const int cell = blockIdx.x, ty = cell / cells_x, cx = cell % cells_x;
    const int tile0 = cx << kStripShift, nt = min(tiles_x - tile0, kStripTiles);
    const int4 cb = cell_bins[cell];
    const int es = min(cb.x, capacity), en = min(cb.x + cb.y, capacity) - es;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < kStripTiles) cnt[threadIdx.x] = 0;
    __syncthreads();
    // span of a record inside the strip: tiles [ta, tb], strip-local
    auto span = [&](uint32_t rx, int &ta, int &tb) {
        const int x0 = (int)(rx & 0xFFFFu), x1 = (int)(rx >> 16);
        ta = max(x0 / GS_TILE, tile0) - tile0;
        tb = min((x1 + GS_TILE - 1) / GS_TILE, tile0 + nt) - 1 - tile0;
    };
    {   // 1. tile counts: sixteen ballots per 64 records, the wave's totals into LDS once
        // (four records per thread and pass with every load requested up front: 33 -> 45 us at C2 — measured,
        // profiles/r06/bench_binab_e_*.json; the registers cost more waves than the round trips saved)
        int32_t c = 0;   // lane t < 16 keeps the wave's count of tile t
        for (int base = (threadIdx.x >> 6) * 64; base < en; base += (int)blockDim.x) {
            const int i = base + lane;
            int ta = 1, tb = 0;
            if (i < en) span(recs[es + i].a.z, ta, tb);
#pragma unroll
            for (int t = 0; t < kStripTiles; t++) {
                const int k = __builtin_popcountll(__builtin_amdgcn_ballot_w64(ta <= t && t <= tb));
                c += lane == t ? k : 0;
            }
        }
        if (lane < kStripTiles && c) atomicAdd(&cnt[lane], c);
    }
    __syncthreads();
    if (threadIdx.x < kStripTiles) {
        const int t = threadIdx.x;
        const int32_t len = cnt[t];
        int32_t start = cb.z;
        for (int j = 0; j < t; j++) start += cnt[j];
        cur[t] = start;
        if (t < nt) bins[ty * tiles_x + tile0 + t] = make_int2(start, start + len);
    }
    __syncthreads();
    // 2. the records of the per-tile sort
    for (int base = (threadIdx.x >> 6) * 64; base < en; base += (int)blockDim.x) {
        const int i = base + lane;
        if (i >= en) continue;
        const uint4 a = recs[es + i].a, b = recs[es + i].b;
        int ta, tb;
        span(a.z, ta, tb);
        const RowTable w = row_table_of(a, b);
        for (int t = ta; t <= tb; t++) {
            uint32_t m;
            if (w.base != kNoRowTable) {
                m = mask_from_rows(w, tile0 + t, ty);
            } else {
                const float4 p0 = packed[3 * (size_t)a.y + 0], p1 = packed[3 * (size_t)a.y + 1];
                m = block_mask16(p0.x, p0.y, p0.z, p0.w, p1.x, __float_as_uint(p1.z), a.z, a.w,
                                 (tile0 + t) * GS_TILE, ty * GS_TILE);
            }
            const int pos = atomicAdd(&cur[t], 1);
            if (pos < capacity) keys[pos] = make_uint4(a.x, a.y, m, 0u);
        }
    }
}

// d. (round 6, second step) k_strip_scatter + the per-tile sort in ONE kernel, one wave per tile, without the
// 16-byte key records in between.  The wave reads the x ranges of ALL its strip's records (filt: 4 B each, the
// strip's sixteen waves run on one XCD and share the lines), keeps the records that reach its tile by ballot
// compaction and — from the same pass — counts the intersections of the strip's tiles to its LEFT: its segment
// starts at (intersections of the strips before) + that count, so tile_bins stays tile-major without a count
// pass, a scan or any word exchanged between waves.  The kept records are then gathered (32 B, lines the
// strip's other waves pull through the same L2), their masks formed from the row tables, and the keys sorted
// by the wave exactly as k_bucket_sort_wave sorts them.  A list beyond the wave's capacity is written out as
// key records for the larger sort classes (or sorted in place by the wave itself when the host skipped them).
template <int PL, int B>
__global__ void __launch_bounds__(64)
k_tile_gather_sort(int cells, int cells_x, int tiles_x, int32_t capacity, int take_longer,
                   const int4 *__restrict__ cell_bins, const uint32_t *__restrict__ filt,
                   const StripRec *__restrict__ recs, const float4 *__restrict__ packed,
                   int2 *__restrict__ bins, uint4 *__restrict__ keys, int32_t *__restrict__ ids_sorted,
                   uint16_t *__restrict__ masks) {
    constexpr int CAP = 64 * PL;
    __shared__ uint64_t out[CAP];
    __shared__ int32_t cnt[B];
    uint32_t *queue = reinterpret_cast<uint32_t *>(out);   // kept record indices, until the sort needs `out`
    // consecutive workgroups go to the 8 XCDs in turn: the sixteen waves of a strip are 8 apart
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int cell = (q >> kStripShift) * 8 + xcd, tl = q & (kStripTiles - 1);
    if (cell >= cells) return;
    const int ty = cell / cells_x, cx = cell % cells_x;
    const int tile0 = cx << kStripShift, nt = min(tiles_x - tile0, kStripTiles);
    if (tl >= nt) return;
    const int lane = threadIdx.x;
    const int4 cb = cell_bins[cell];
    const int es = min(cb.x, capacity), en = min(cb.x + cb.y, capacity) - es;
    const int tx = tile0 + tl;
    // 1. filter: four independent loads in flight per lane
    int32_t before = 0;
    int n = 0;
    for (int base = 0; base < en; base += 256) {
        uint32_t f[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = base + 64 * k + lane;
            f[k] = i < en ? filt[es + i] : 0u;     // (x0 = x1 = 0: reaches no tile)
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x0 = (int)(f[k] & 0xFFFFu), x1 = (int)(f[k] >> 16);
            const int ta = max(x0 / GS_TILE, tile0), tb = min((x1 + GS_TILE - 1) / GS_TILE, tile0 + nt) - 1;   // inclusive
            before += max(min(tb, tx - 1) - ta + 1, 0);
            const bool keep = ta <= tx && tx <= tb;
            const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
            const int slot = n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (keep && slot < CAP) queue[slot] = (uint32_t)(base + 64 * k + lane);
            n += __builtin_popcountll(m);
        }
    }
    before = __builtin_amdgcn_readlane(wave_inclusive_scan_i(before), 63);
    const int start = cb.z + before;
    if (lane == 0) bins[ty * tiles_x + tx] = make_int2(min(start, capacity), min(start + n, capacity));
    const int nfit = min(start + n, capacity) - min(start, capacity);   // what the id list holds of this segment
    if (nfit <= 0) return;
    auto entry = [&](int i, uint64_t &key, uint32_t &mask) {
        const uint4 a = recs[es + i].a, b = recs[es + i].b;
        const RowTable w = row_table_of(a, b);
        if (w.base != kNoRowTable) {
            mask = mask_from_rows(w, tx, ty);
        } else {
            const float4 p0 = packed[3 * (size_t)a.y + 0], p1 = packed[3 * (size_t)a.y + 1];
            mask = block_mask16(p0.x, p0.y, p0.z, p0.w, p1.x, __float_as_uint(p1.z), a.z, a.w, tx * GS_TILE,
                                ty * GS_TILE);
        }
        key = ((uint64_t)a.x << 32) | a.y;
    };
    if (n > CAP) {
        // a list beyond this class: the filter again, writing the key records of the larger classes
        int m2 = 0;
        for (int base = 0; base < en; base += 64) {
            const int i = base + lane;
            bool keep = false;
            if (i < en) {
                const uint32_t f = filt[es + i];
                const int x0 = (int)(f & 0xFFFFu), x1 = (int)(f >> 16);
                keep = max(x0 / GS_TILE, tile0) <= tx && tx <= min((x1 + GS_TILE - 1) / GS_TILE, tile0 + nt) - 1;
            }
            const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
            const int slot = m2 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (keep && slot < nfit) {
                uint64_t key;
                uint32_t mask;
                entry(i, key, mask);
                keys[start + slot] = make_uint4((uint32_t)(key >> 32), (uint32_t)key, mask, 0u);
            }
            m2 += __builtin_popcountll(m);
        }
        if (take_longer) {
            __syncthreads();   // (one wave: orders the global stores above against the loads of the network)
            __threadfence_block();
            sort_in_global<64>(keys + start, nfit, lane, ids_sorted + start, masks + start);
        }
        return;
    }
    __syncthreads();   // the queue is complete
    // 2. gather the kept records: keys and masks into registers
    uint64_t kk[PL];
    uint32_t mm[PL / 2];
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
    for (int j = 0; j < PL / 2; j++) mm[j] = 0u;
    uint32_t qi[PL];
#pragma unroll
    for (int j = 0; j < PL; j++) qi[j] = j * 64 + lane < nfit ? queue[j * 64 + lane] : 0u;
    __syncthreads();   // `out` is free for the sort
#pragma unroll
    for (int j = 0; j < PL; j++) {
        kk[j] = ~0ull;
        if (j * 64 + lane < nfit) {
            uint32_t mask;
            entry((int)qi[j], kk[j], mask);
            mm[j >> 1] |= (mask & 0xFFFFu) << (16 * (j & 1));
            const uint32_t d = (uint32_t)(kk[j] >> 32);
            mn = min(mn, d);
            mx = max(mx, d);
        }
    }
    wave_bucket_sort<PL, B>(kk, mm, mn, mx, nfit, lane, out, cnt, ids_sorted + start, masks + start);
}

// the same order from finished tile_bins (strip binning: the tile counts only exist once the strips have been
// split): one 1024-thread workgroup, list lengths in LDS when they fit.  Also the frame's longest list, to the
// device word and — with M — to the pinned statistics of the host.
__global__ void __launch_bounds__(1024)
k_tile_order(int tiles, int use_lds, int order_mult, const int2 *__restrict__ bins,
             int32_t *__restrict__ longest_dev, int32_t *__restrict__ stats_host,
             int32_t *__restrict__ order) {
    extern __shared__ int32_t len_lds[];
    __shared__ int32_t part[1024];
    __shared__ int32_t ws[17];
    __shared__ int32_t s_longest;
    const int t = threadIdx.x;
    if (t == 0) s_longest = 0;
    part[t] = 0;
    __syncthreads();
    int32_t mx = 0;
    for (int i = t; i < tiles; i += 1024) {
        const int2 b = bins[i];
        if (use_lds) len_lds[i] = b.y - b.x;
        mx = max(mx, b.y - b.x);
    }
    mx = wave_max_i(mx);
    if ((t & 63) == 0) atomicMax(&s_longest, mx);
    __syncthreads();
    const int32_t longest = s_longest;
    if (t == 0) {
        *longest_dev = longest;
        if (stats_host) stats_host[1] = longest;   // pinned, device-mapped
    }
    if (!order) return;
    const int shift = order_shift(longest), ostep = order_step(tiles, order_mult);
    auto length = [&](int i) -> int32_t {
        if (use_lds) return len_lds[i];
        const int2 b = bins[i];
        return b.y - b.x;
    };
    for (int k = 0, i = order_first(t, tiles, order_mult); k * 1024 + t < tiles; k++, i = order_next(i, tiles, ostep))
        atomicAdd(&part[1023 - (length(i) >> shift)], 1);
    __syncthreads();
    const int32_t own = part[t];
    const int32_t first = block_scan_1024(own, ws) - own;
    part[t] = first;
    __syncthreads();
    for (int k = 0, i = order_first(t, tiles, order_mult); k * 1024 + t < tiles; k++, i = order_next(i, tiles, ostep))
        order[atomicAdd(&part[1023 - (length(i) >> shift)], 1)] = i;
}

// mult of the scattered tile sequence (order_first / order_next): ~0.618 tiles, coprime to tiles
static int order_multiplier(int tiles) {
    auto gcd = [](int a, int b) { while (b) { const int r = a % b; a = b; b = r; } return a; };
    int m = (int)(tiles * 0.6180339887) | 1;
    while (gcd(m, tiles) != 1) m += 2;
    return m;
}

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// LDS budget of the privatised count / scatter kernels (the CU has 160 KiB) and their grid: one
// persistent workgroup per CU is enough to saturate the LDS atomic units; fewer for small N.
constexpr size_t kMaxTileLds = 144 * 1024;
static int persistent_blocks(int N) {
    int b = (N + 63) / 64;  // at least one 64-Gaussian chunk per workgroup
    return b < 256 ? (b < 1 ? 1 : b) : 256;
}

// workspace layout: [ counters: tiles i32 | total: 1 i32 (+pad) | wg_base: 256 x tiles i32 | records:
// capacity x 16 B ].  gs_bin_scan uses the counters as per-tile intersection counts and leaves the
// per-(workgroup, tile) offsets in wg_base for the gs_bin_sort that follows: BOTH CALLS MUST BE GIVEN
// THE SAME WORKSPACE (the offsets of the key array move with the capacity, wg_base does not).
struct BinLayout {
    size_t counters, total_dev, wg_base, keys, total;
    // strip binning (section 6): [ strip counters u64 x cells | longest list ] is one zeroed block; cell_bins
    // int4 x cells; cell_base 256 x cells i32; recs: capacity x 32 B and filt: capacity x 4 B behind the keys
    size_t cell_counts, longest, zero_bytes, cell_bins, cell_base, recs, filt;
    int cells_x, cells;   // cells = 0: the image has too many strips, tile-level path only
};
// What gs_bin_scan remembers (on the host, per workspace address) so that gs_bin_sort can tell that the
// workspace it is handed holds the scan's leftovers — the per-workgroup offsets — for the same
// problem; a mismatch is GS_ERR_WORKSPACE instead of silently corrupt lists.
struct BinStamp {
    int N, W, H, blocks;
    uint64_t seq = 0;   // order of stamping (eviction)
};
static std::mutex g_stamp_mutex;
static std::unordered_map<const void *, BinStamp> g_stamps;
static uint64_t g_stamp_clock = 0;
static void stamp_workspace(const void *ws, BinStamp st) {
    std::lock_guard<std::mutex> lock(g_stamp_mutex);
    st.seq = ++g_stamp_clock;
    g_stamps[ws] = st;
    // workspaces come and go with the allocator: the OLDEST stamps leave, one at a time — a wholesale
    // clear could drop the stamp another thread (or GPU) set between its scan and its sort;
    // 4096 live (scan, sort) pairs in flight at once is far beyond any caller
    while (g_stamps.size() > 4096) {
        auto oldest = g_stamps.begin();
        for (auto it = g_stamps.begin(); it != g_stamps.end(); ++it)
            if (it->second.seq < oldest->second.seq) oldest = it;
        g_stamps.erase(oldest);
    }
}
static bool workspace_matches(const void *ws, const BinStamp &st) {
    std::lock_guard<std::mutex> lock(g_stamp_mutex);
    auto it = g_stamps.find(ws);
    return it != g_stamps.end() && it->second.N == st.N && it->second.W == st.W &&
           it->second.H == st.H && it->second.blocks == st.blocks;
}
static BinLayout bin_layout(int N, int64_t capacity, int W, int H) {
    const size_t tiles = (size_t)((W + GS_TILE - 1) / GS_TILE) * ((H + GS_TILE - 1) / GS_TILE);
    BinLayout L;
    L.counters = 0;
    L.total_dev = L.counters + align_up(tiles * 4);
    L.wg_base = L.total_dev + 256;
    // per-(workgroup, tile) offsets handed from the count to the scatter kernel (LDS variants only)
    const size_t base_bytes = tiles * 4 <= kMaxTileLds ? align_up((size_t)256 * tiles * 4) : 0;
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    L.cells_x = (tiles_x + kStripTiles - 1) / kStripTiles;
    L.cells = (size_t)L.cells_x * tiles_y <= (size_t)kMaxCells ? L.cells_x * tiles_y : 0;
    L.cell_counts = L.wg_base + base_bytes;
    L.longest = L.cell_counts + align_up((size_t)L.cells * 8);
    L.zero_bytes = L.longest + 256 - L.cell_counts;
    L.cell_bins = L.longest + 256;
    L.cell_base = L.cell_bins + align_up((size_t)L.cells * 16);
    L.keys = L.cell_base + align_up((size_t)256 * L.cells * 4);
    const size_t cap_bytes = align_up((size_t)(capacity > 0 ? capacity : 1) * 16);
    L.recs = L.keys + cap_bytes + 256;
    L.filt = L.recs + (L.cells ? 2 * cap_bytes : 0) + 256;
    L.total = L.filt + (L.cells ? align_up((size_t)(capacity > 0 ? capacity : 1) * 4) : 0) + 256;
    return L;
}

// The per-tile sorts of section 4 over every tile's segment of `keys`: which size classes are launched follows
// the previous frame's statistics.
// order: the frame's tile_order if the scan has written it (nullptr otherwise): the long class then walks its head
// with a small grid instead of launching a 1024-thread workgroup for every tile (k_bucket_sort_tiles).
static int launch_tile_sorts(int tiles, int32_t capacity, const int32_t *list_stats, int2 *bins_rw,
                             uint4 *keys, int32_t *gaussian_ids_sorted, uint16_t *block_masks,
                             hipStream_t s, const int32_t *order = nullptr) {
    const int2 *bins = bins_rw;
    // Every class also moves the coverage masks (third word of the records) along with the keys.
    // segments <= 512 keys: one wave, keys in registers, 512 buckets (6 KiB LDS); <= 1024 keys:
    // the same with 1024 buckets (12 KiB); <= 8192 keys: 256 threads, 4096 buckets (80 KiB LDS);
    // longer: in place in global memory.
    // Which classes are launched follows the PREVIOUS frame's statistics (list_stats = {M, longest
    // list}); a launch over all tiles that finds nothing to do still costs its dispatch (39 us for the
    // 256-thread class at 4K).  Whatever is launched last-in-class takes any longer segment itself (in
    // place in global memory: slow, correct, and only on the frame where a list first outgrows the guess).
    //   longest <= 400:              the 512 class alone
    //   longest <= 900:              no 8192 class; lists mostly beyond 512 (mean > 300): the 1024 class
    //                                alone, for every segment
    //   otherwise / no statistics:   all three
    //   at most 1024 tiles (launch-bound): longest <= 900 -> the 1024 class alone; beyond -> 8192 + 1024 classes
    const bool have_stats = list_stats && list_stats[0] > 0;
    const bool only_short = have_stats && list_stats[1] <= 400;
    const bool no_long = have_stats && list_stats[1] <= 900;
    // (a frame of few tiles is bound by the launches, not by the sorting: one launch of the wider class —
    // 6000 Gaussians at 384x288: 23 us instead of 23 + 18)
    const bool only_mid = no_long && !only_short &&
                          ((int64_t)list_stats[0] > 300 * (int64_t)tiles || tiles <= 1024);
    if (only_mid) {
        if (tiles > kMidBucketsTiles)
            GS_LAUNCH((k_bucket_sort_wave<16, kMidBuckets>), dim3(tiles), dim3(64), 0, s, 0, 1024,
                               capacity, 1, 1, bins_rw, keys, gaussian_ids_sorted, block_masks);
        else
            GS_LAUNCH((k_bucket_sort_wave<16, kMidBucketsFewTiles>), dim3(tiles), dim3(64), 0, s, 0, 1024,
                               capacity, 1, 1, bins_rw, keys, gaussian_ids_sorted, block_masks);
        GS_LAUNCH_CHECK();
        return GS_OK;
    }
    // few tiles, long lists: the 8192 class for what is beyond 1024 keys and the 1024 class for everything else
    // (20 000 Gaussians at 384x288: 30 + 28 + 19 us with the 512 class as a third launch)
    const bool few_long = have_stats && !no_long && tiles <= 1024;
    // (Launching the classes side by side on streams of the library's own — they work on disjoint tiles, every class
    // clamps the ranges it reads by `capacity` itself — was built and measured twice: in round 5 as three full grids
    // (each launch got slower by what it shared, the stage 0.233 -> 0.251 ms on the hot-spot scene), in round 6 with
    // the long class as the small-grid walk below and the mid class on a second stream (fork / join events around
    // them: the step 0.682 against 0.681 ms on the hot-spot scene, the instrumented stage 0.226 against 0.216 — what
    // the overlap gains the cross-stream waits cost).  In a row.)
    // The long class walks the head of tile_order on a small grid.  (A longest list of 2^20 entries and more:
    // order_shift is then so wide that the class of 1025 is the lowest one and nothing can stop the walk — one
    // workgroup per tile as before.)
    const bool walk = !no_long && order != nullptr && !(have_stats && list_stats[1] >= (1 << 20));
    // A frame of mostly short lists (mean <= 300) with a few long ones — the hot-spot scene: thirty-odd lists of 513 ...
    // 1024 entries next to nine long ones: the walking workgroups take those as well (lo_n = 512) and the 1024-key wave
    // class — 26 us of ONE wave's latency chain per tile, a launch of its own — is not launched at all.  (Where lists
    // of that length are the bulk of the frame, 4K with a hot spot, the wave class keeps them: 256 workgroups walking
    // twenty thousand tiles would take ten times as long.)
    const bool merge_mid = walk && have_stats && !few_long && !only_short &&
                           (int64_t)list_stats[0] <= 300 * (int64_t)tiles;
    if (!only_short && !few_long && !merge_mid) {
        if (tiles > kMidBucketsTiles)
            GS_LAUNCH((k_bucket_sort_wave<16, kMidBuckets>), dim3(tiles), dim3(64), 0, s, 512, 1024,
                               capacity, 0, no_long ? 1 : 0, bins_rw, keys, gaussian_ids_sorted, block_masks);
        else
            GS_LAUNCH((k_bucket_sort_wave<16, kMidBucketsFewTiles>), dim3(tiles), dim3(64), 0, s, 512, 1024,
                               capacity, 0, no_long ? 1 : 0, bins_rw, keys, gaussian_ids_sorted, block_masks);
        GS_LAUNCH_CHECK();
    }
    if (!no_long) {
        // 1024 threads per tile: the few tiles of this class are latency chains of one workgroup each —
        // with 256 threads 77 us for the nine 5 - 8 k-entry lists of the hot-spot scene, 36 us with 1024
        // (bin_sort 0.272 -> 0.233 ms there)
        constexpr int CAP = 8192, B = GS_LONG_BUCKETS, NT = 1024;
        const size_t lds = 8 * CAP + 4 * B + 256 + 2 * CAP;
        GS_HIP_CHECK(hipFuncSetAttribute(
            reinterpret_cast<const void *>(k_bucket_sort_tiles<CAP, B, NT>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        GS_LAUNCH((k_bucket_sort_tiles<CAP, B, NT>), dim3(walk ? (tiles < 256 ? tiles : 256) : tiles), dim3(NT), lds,
                  s, merge_mid ? 512 : 1024, CAP, capacity, tiles, walk ? order : nullptr, bins, keys, gaussian_ids_sorted,
                  block_masks);
        GS_LAUNCH_CHECK();
    }
    // (the short class last: it also clamps overflowing ranges, after the others have read them)
    if (few_long)
        GS_LAUNCH((k_bucket_sort_wave<16, kMidBucketsFewTiles>), dim3(tiles), dim3(64), 0, s, 0, 1024, capacity,
                           1, 0, bins_rw, keys, gaussian_ids_sorted, block_masks);   // (few_long: at most 1024 tiles)
    else
        GS_LAUNCH((k_bucket_sort_wave<8, kShortBuckets>), dim3(tiles), dim3(64), 0, s, 0, 512, capacity,
                           1, only_short ? 1 : 0, bins_rw, keys, gaussian_ids_sorted, block_masks);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

}  // namespace gs

extern "C" int gs_pack_splats(int W, int H, int N, const float *xys, const int32_t *radii,
                              const float *conics, const float *colors, const float *opacities,
                              const float *cov2d, float *packed, int32_t *tiles_hit,
                              uint32_t flags, gs_stream_t stream) {
    if (N < 0 || W <= 0 || H <= 0) return GS_ERR_INVALID_ARGUMENT;
    if (W > 65535 || H > 65535) return GS_ERR_UNSUPPORTED;
    if (N == 0) return GS_OK;
    if (!xys || !radii || !conics || !colors || !opacities || !packed || !tiles_hit)
        return GS_ERR_INVALID_ARGUMENT;
    if ((uintptr_t)packed & 15u) return GS_ERR_INVALID_ARGUMENT;
    GS_LAUNCH(gs::k_pack_splats, dim3((N + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, W, H, N, xys, radii, conics, colors, opacities,
                       cov2d, reinterpret_cast<float4 *>(packed), tiles_hit, flags);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" size_t gs_bin_workspace_bytes(int N, int64_t num_isects, int W, int H) {
    if (N < 0 || num_isects < 0 || W <= 0 || H <= 0) return 0;
    return gs::bin_layout(N, num_isects, W, H).total;
}

extern "C" size_t gs_bin_num_isects_offset(int W, int H) {
    if (W <= 0 || H <= 0) return 0;
    return gs::bin_layout(0, 0, W, H).total_dev;
}

extern "C" int gs_bin_scan(int W, int H, int N, const float *packed, int32_t *tile_bins,
                           int32_t *tile_order, int32_t *num_isects_host, void *workspace,
                           size_t workspace_bytes, gs_stream_t stream) {
    GS_TRACE("gs_bin_scan");
    if (N < 0 || W <= 0 || H <= 0) return GS_ERR_INVALID_ARGUMENT;
    if (W > 65535 || H > 65535) return GS_ERR_UNSUPPORTED;
    if (!tile_bins || !workspace) return GS_ERR_INVALID_ARGUMENT;
    if ((uintptr_t)workspace & 15u) return GS_ERR_INVALID_ARGUMENT;
    if (N > 0 && !packed) return GS_ERR_INVALID_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    const int tiles = tiles_x * tiles_y;
    const gs::BinLayout L = gs::bin_layout(N, 0, W, H);
    if (workspace_bytes < L.total) return GS_ERR_WORKSPACE;
    char *base = static_cast<char *>(workspace);
    int32_t *counts = reinterpret_cast<int32_t *>(base + L.counters);
    int32_t *total_dev = reinterpret_cast<int32_t *>(base + L.total_dev);
    gs::timeline_before(s);
    GS_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)tiles, s));
    gs::timeline_after("memset(tile counters)", s);
    if (N > 0) {
        const size_t lds = sizeof(int32_t) * (size_t)tiles;
        if (lds <= gs::kMaxTileLds) {
            GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gs::k_count_tiles),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)gs::kMaxTileLds));
            const int blocks = gs::persistent_blocks(N);
            GS_LAUNCH(gs::k_count_tiles, dim3(blocks), dim3(gs::kPersistentThreads), lds, s, N, tiles, tiles_x,
                               reinterpret_cast<const float4 *>(packed), counts,
                               reinterpret_cast<int32_t *>(base + L.wg_base), (uint4 *)nullptr, (size_t)0);
        } else {
            GS_LAUNCH(gs::k_count_tiles_global, dim3((N + 255) / 256), dim3(256), 0, s, N,
                               tiles_x, reinterpret_cast<const float4 *>(packed), counts);
        }
        GS_LAUNCH_CHECK();
    }
    gs::stamp_workspace(workspace, gs::BinStamp{N, W, H, gs::persistent_blocks(N), 0});
    {
        const size_t lds = sizeof(int32_t) * ((size_t)tiles + tiles / 32 + 1);
        const int use_lds = lds <= gs::kMaxTileLds ? 1 : 0;
        if (use_lds)
            GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gs::k_scan_tiles_fast),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)gs::kMaxTileLds));
        if (use_lds)
            GS_LAUNCH(gs::k_scan_tiles_fast, dim3(1), dim3(1024), lds, s, tiles, gs::order_multiplier(tiles), counts,
                               reinterpret_cast<int2 *>(tile_bins), total_dev, num_isects_host, tile_order);
        else
            GS_LAUNCH(gs::k_scan_tiles, dim3(1), dim3(1024), 0, s, tiles, 0, gs::order_multiplier(tiles), counts,
                               reinterpret_cast<int2 *>(tile_bins), total_dev, num_isects_host, tile_order);
    }
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" int gs_block_masks(int W, int H, const int32_t *gaussian_ids_sorted,
                              const int32_t *tile_bins, const float *packed, uint16_t *block_masks,
                              gs_stream_t stream) {
    GS_TRACE("gs_block_masks");
    if (W <= 0 || H <= 0) return GS_ERR_INVALID_ARGUMENT;
    if (W > 65535 || H > 65535) return GS_ERR_UNSUPPORTED;
    if (!tile_bins || !gaussian_ids_sorted || !packed || !block_masks) return GS_ERR_INVALID_ARGUMENT;
    if ((uintptr_t)packed & 15u) return GS_ERR_INVALID_ARGUMENT;
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    GS_LAUNCH(gs::k_block_masks, dim3(tiles_x * tiles_y), dim3(64), 0, (hipStream_t)stream,
                       tiles_x, reinterpret_cast<const int2 *>(tile_bins), gaussian_ids_sorted,
                       reinterpret_cast<const float4 *>(packed), block_masks);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" int gs_bin_sort(int W, int H, int N, int32_t capacity, const float *packed,
                           const float *depths, int32_t *tile_bins,
                           int32_t *gaussian_ids_sorted, uint16_t *block_masks,
                           const int32_t *list_stats,
                           void *workspace, size_t workspace_bytes, gs_stream_t stream) {
    GS_TRACE("gs_bin_sort");
    if (N < 0 || capacity < 0 || W <= 0 || H <= 0) return GS_ERR_INVALID_ARGUMENT;
    if (W > 65535 || H > 65535) return GS_ERR_UNSUPPORTED;
    if (N == 0 || capacity == 0) return GS_OK;
    if (!packed || !depths || !tile_bins || !gaussian_ids_sorted || !block_masks || !workspace)
        return GS_ERR_INVALID_ARGUMENT;
    if ((uintptr_t)workspace & 15u) return GS_ERR_INVALID_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    const int tiles = tiles_x * tiles_y;
    const gs::BinLayout L = gs::bin_layout(N, capacity, W, H);
    if (workspace_bytes < L.total) return GS_ERR_WORKSPACE;
    // the scan's per-workgroup offsets must be the ones of THIS problem
    if (!gs::workspace_matches(workspace, gs::BinStamp{N, W, H, gs::persistent_blocks(N), 0}))
        return GS_ERR_WORKSPACE;
    char *base = static_cast<char *>(workspace);
    int32_t *fill = reinterpret_cast<int32_t *>(base + L.counters);
    uint4 *keys = reinterpret_cast<uint4 *>(base + L.keys);
    const int2 *bins = reinterpret_cast<const int2 *>(tile_bins);
    const size_t lds = sizeof(int32_t) * (size_t)tiles;
    if (lds <= gs::kMaxTileLds) {
        GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gs::k_scatter),
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)gs::kMaxTileLds));
        const int blocks = gs::persistent_blocks(N);
        GS_LAUNCH(gs::k_scatter, dim3(blocks), dim3(gs::kPersistentThreads), lds, s, N, tiles, tiles_x, capacity,
                           reinterpret_cast<const float4 *>(packed), depths, bins,
                           reinterpret_cast<const int32_t *>(base + L.wg_base), keys);
    } else {
        GS_HIP_CHECK(hipMemsetAsync(fill, 0, sizeof(int32_t) * (size_t)tiles, s));
        GS_LAUNCH(gs::k_scatter_global, dim3((N + 255) / 256), dim3(256), 0, s, N, tiles_x,
                           capacity, reinterpret_cast<const float4 *>(packed), depths, bins, fill,
                           keys);
    }
    GS_LAUNCH_CHECK();
    return gs::launch_tile_sorts(tiles, capacity, list_stats, reinterpret_cast<int2 *>(tile_bins), keys,
                                 gaussian_ids_sorted, block_masks, s);
}

extern "C" int gs_bin_strips(int W, int H, int N, int32_t capacity, const float *packed,
                             const float *depths, int32_t *tile_bins, int32_t *gaussian_ids_sorted,
                             uint16_t *block_masks, int32_t *tile_order, int32_t *num_isects_host,
                             const int32_t *list_stats, void *workspace, size_t workspace_bytes,
                             gs_stream_t stream) {
    GS_TRACE("gs_bin_strips");
    if (N < 0 || capacity < 0 || W <= 0 || H <= 0) return GS_ERR_INVALID_ARGUMENT;
    if (W > 65535 || H > 65535) return GS_ERR_UNSUPPORTED;
    if (!tile_bins || !workspace || !tile_order) return GS_ERR_INVALID_ARGUMENT;
    if ((uintptr_t)workspace & 15u) return GS_ERR_INVALID_ARGUMENT;
    if (N > 0 && (!packed || !depths)) return GS_ERR_INVALID_ARGUMENT;
    if (capacity > 0 && (!gaussian_ids_sorted || !block_masks)) return GS_ERR_INVALID_ARGUMENT;
    const gs::BinLayout L = gs::bin_layout(N, capacity, W, H);
    if (workspace_bytes < L.total) return GS_ERR_WORKSPACE;
    if (L.cells == 0 || N == 0 || capacity == 0) {
        // more strips than the persistent kernels keep in LDS (or nothing to sort): the tile-level path
        int rc = gs_bin_scan(W, H, N, packed, tile_bins, tile_order, num_isects_host, workspace, workspace_bytes,
                             stream);
        if (rc != GS_OK) return rc;
        return gs_bin_sort(W, H, N, capacity, packed, depths, tile_bins, gaussian_ids_sorted, block_masks,
                           list_stats, workspace, workspace_bytes, stream);
    }
    hipStream_t s = (hipStream_t)stream;
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    const int tiles = tiles_x * tiles_y;
    char *base = static_cast<char *>(workspace);
    auto *counts = reinterpret_cast<unsigned long long *>(base + L.cell_counts);
    int32_t *longest = reinterpret_cast<int32_t *>(base + L.longest);
    int4 *cell_bins = reinterpret_cast<int4 *>(base + L.cell_bins);
    int32_t *cell_base = reinterpret_cast<int32_t *>(base + L.cell_base);
    uint4 *keys = reinterpret_cast<uint4 *>(base + L.keys);
    gs::StripRec *recs = reinterpret_cast<gs::StripRec *>(base + L.recs);
    int32_t *total_dev = reinterpret_cast<int32_t *>(base + L.total_dev);
    const float4 *pk = reinterpret_cast<const float4 *>(packed);
    gs::timeline_before(s);
    GS_HIP_CHECK(hipMemsetAsync(base + L.cell_counts, 0, L.zero_bytes, s));
    gs::timeline_after("memset(strip counters)", s);
    const int blocks = gs::persistent_blocks(N);
    GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gs::k_cell_count),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, gs::kMaxCells * 8));
    GS_LAUNCH(gs::k_cell_count, dim3(blocks), dim3(gs::kPersistentThreads), (size_t)L.cells * 8, s, N, L.cells,
              L.cells_x, pk, counts, cell_base);
    GS_LAUNCH_CHECK();
    uint32_t *filt = reinterpret_cast<uint32_t *>(base + L.filt);
    int2 *bins = reinterpret_cast<int2 *>(tile_bins);
    // measurement switch: GSPLAT_STRIPS_FUSED=1 replaces k_strip_scatter + the tile-level path's sort kernels by
    // k_tile_gather_sort (measured slower: profiles/HISTORY.md)
    static const bool fused = [] { const char *e = getenv("GSPLAT_STRIPS_FUSED"); return e && e[0] == '1'; }();
    GS_LAUNCH(gs::k_cell_scatter, dim3(blocks), dim3(gs::kPersistentThreads), (size_t)L.cells * 4, s, N, L.cells,
              L.cells_x, capacity, pk, depths, counts, cell_base, cell_bins, total_dev, num_isects_host, recs,
              fused ? filt : nullptr);
    GS_LAUNCH_CHECK();
    int rc = GS_OK;
    auto order_tiles = [&]() -> int {
        const size_t lds = sizeof(int32_t) * (size_t)tiles;
        const int use_lds = lds <= gs::kMaxTileLds ? 1 : 0;
        if (use_lds)
            GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gs::k_tile_order),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)gs::kMaxTileLds));
        GS_LAUNCH(gs::k_tile_order, dim3(1), dim3(1024), use_lds ? lds : 0, s, tiles, use_lds,
                  gs::order_multiplier(tiles), bins, longest, num_isects_host, tile_order);
        GS_LAUNCH_CHECK();
        return GS_OK;
    };
    if (!fused) {
        GS_LAUNCH(gs::k_strip_scatter, dim3(L.cells), dim3(gs::kPersistentThreads), 0, s, L.cells_x, tiles_x, capacity,
                  cell_bins, recs, pk, keys, bins, longest);
        GS_LAUNCH_CHECK();
        // (before the sorts: their last launch clamps tile_bins to the capacity, and the statistics must be the
        // frame's true ones also when the id list was too small)
        rc = order_tiles();
        if (rc != GS_OK) return rc;
        return gs::launch_tile_sorts(tiles, capacity, list_stats, bins, keys, gaussian_ids_sorted, block_masks, s);
    } else {
        // one wave per tile: filter + gather + sort.  Size classes as in launch_tile_sorts, from the previous
        // frame's statistics: the wave takes what fits its registers (512 or 1024 keys) and leaves longer lists
        // as key records to the larger classes launched behind it — or sorts them in place itself where the
        // host skipped those launches.
        const bool have_stats = list_stats && list_stats[0] > 0;
        const bool only_short = have_stats && list_stats[1] <= 400;
        const bool no_long = have_stats && list_stats[1] <= 900;
        const bool only_mid = no_long && !only_short &&
                              ((int64_t)list_stats[0] > 300 * (int64_t)tiles || tiles <= 1024);
        const bool few_long = have_stats && !no_long && tiles <= 1024;
        const int waves = (L.cells + 7) / 8 * 8 * gs::kStripTiles;
#define GS_FUSED(PL, B, TAKE)                                                                                  \
    GS_LAUNCH((gs::k_tile_gather_sort<PL, B>), dim3(waves), dim3(64), 0, s, L.cells, L.cells_x, tiles_x,       \
              capacity, TAKE, cell_bins, filt, recs, pk, bins, keys, gaussian_ids_sorted, block_masks)
        if (only_mid) {
            GS_FUSED(16, gs::kMidBucketsFewTiles, 1);
        } else if (few_long) {
            GS_FUSED(16, gs::kMidBucketsFewTiles, 0);
        } else {
            GS_FUSED(8, gs::kShortBuckets, only_short ? 1 : 0);
        }
        GS_LAUNCH_CHECK();
        if (!only_mid && !only_short && !few_long) {
            GS_LAUNCH((gs::k_bucket_sort_wave<16, gs::kMidBucketsFewTiles>), dim3(tiles), dim3(64), 0, s, 512, 1024, capacity, 0,
                      no_long ? 1 : 0, bins, keys, gaussian_ids_sorted, block_masks);
            GS_LAUNCH_CHECK();
        }
        if (!no_long) {
            constexpr int CAP = 8192, B = GS_LONG_BUCKETS, NT = 1024;
            const size_t lds = 8 * CAP + 4 * B + 256 + 2 * CAP;
            GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gs::k_bucket_sort_tiles<CAP, B, NT>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            GS_LAUNCH((gs::k_bucket_sort_tiles<CAP, B, NT>), dim3(tiles), dim3(NT), lds, s, 1024, CAP, capacity, tiles,
                      (const int32_t *)nullptr, bins, keys, gaussian_ids_sorted, block_masks);
            GS_LAUNCH_CHECK();
        }
#undef GS_FUSED
    }
    if (rc != GS_OK) return rc;
    // (fused: tile_bins are clamped as they are written — on a frame whose id list was too small the longest list
    // reported is the clamped one)
    return order_tiles();
}

extern "C" int gs_bin_speculative(int W, int H, int N, int32_t capacity, const float *packed, const float *depths,
                                  int32_t *tile_bins, int32_t *gaussian_ids_sorted, uint16_t *block_masks,
                                  int32_t *tile_order, int32_t *num_isects_host, const int32_t *list_stats,
                                  void *workspace, size_t workspace_bytes, gs_stream_t stream) {
    return gs_bin_speculative_zero(W, H, N, capacity, packed, depths, tile_bins, gaussian_ids_sorted, block_masks,
                                   tile_order, num_isects_host, list_stats, workspace, workspace_bytes, nullptr, 0,
                                   stream);
}

extern "C" int gs_bin_speculative_zero(int W, int H, int N, int32_t capacity, const float *packed, const float *depths,
                                       int32_t *tile_bins, int32_t *gaussian_ids_sorted, uint16_t *block_masks,
                                       int32_t *tile_order, int32_t *num_isects_host, const int32_t *list_stats,
                                       void *workspace, size_t workspace_bytes, void *zero_ptr, size_t zero_bytes,
                                       gs_stream_t stream) {
    GS_TRACE("gs_bin_speculative");
    if (zero_ptr && (((uintptr_t)zero_ptr & 15u) || (zero_bytes & 15u))) return GS_ERR_INVALID_ARGUMENT;
    if (!zero_ptr) zero_bytes = 0;
    if (N < 0 || capacity < 0 || W <= 0 || H <= 0) return GS_ERR_INVALID_ARGUMENT;
    if (W > 65535 || H > 65535) return GS_ERR_UNSUPPORTED;
    if (!tile_bins || !workspace) return GS_ERR_INVALID_ARGUMENT;
    if ((uintptr_t)workspace & 15u) return GS_ERR_INVALID_ARGUMENT;
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    const int tiles = tiles_x * tiles_y;
    const size_t lds = sizeof(int32_t) * ((size_t)tiles + tiles / 32 + 1);
    // measurement switch: GSPLAT_BIN_SCAN_LAUNCH=1 keeps the scan as a launch of its own (rounds 1 - 5)
    static const bool own_launch = [] { const char *e = getenv("GSPLAT_BIN_SCAN_LAUNCH"); return e && e[0] == '1'; }();
    if (own_launch || N == 0 || capacity == 0 || lds > gs::kMaxTileLds || !packed || !depths ||
        !gaussian_ids_sorted || !block_masks) {
        // (the two-call path: the caller's buffer by a plain fill)
        if (zero_bytes) GS_HIP_CHECK(hipMemsetAsync(zero_ptr, 0, zero_bytes, (hipStream_t)stream));
        int rc = gs_bin_scan(W, H, N, packed, tile_bins, tile_order, num_isects_host, workspace, workspace_bytes,
                             stream);
        if (rc != GS_OK) return rc;
        return gs_bin_sort(W, H, N, capacity, packed, depths, tile_bins, gaussian_ids_sorted, block_masks,
                           list_stats, workspace, workspace_bytes, stream);
    }
    hipStream_t s = (hipStream_t)stream;
    const gs::BinLayout L = gs::bin_layout(N, capacity, W, H);
    if (workspace_bytes < L.total) return GS_ERR_WORKSPACE;
    char *base = static_cast<char *>(workspace);
    int32_t *counts = reinterpret_cast<int32_t *>(base + L.counters);
    int32_t *total_dev = reinterpret_cast<int32_t *>(base + L.total_dev);
    int32_t *wg_base = reinterpret_cast<int32_t *>(base + L.wg_base);
    uint4 *keys = reinterpret_cast<uint4 *>(base + L.keys);
    const float4 *pk = reinterpret_cast<const float4 *>(packed);
    // The count pass carries the caller's zeroes only where it has the slack: at 1 M Gaussians (64 MB) it gets 2.5 us
    // longer and a 12 us fill kernel goes away (step 0.591 -> 0.582 ms); at 5 M (320 MB) it is a streaming kernel itself
    // — 76 -> 150 us against the fill's 41 (measured) — and the fill stays a launch of its own.
    // Nor does a fill of that size belong anywhere near the binning: in front of the count pass it pushed the records
    // that pass reads out of the last-level cache (74 -> 99 us), behind the sorts it cost the compositing kernels 20 us
    // each.  The call then returns GS_OK_NOT_ZEROED and the caller leaves the fill to gs_rasterize_backward, right in
    // front of its atomics, as before.
    constexpr size_t kZeroInCountMax = (size_t)96 << 20;
    const bool not_zeroed = zero_bytes > kZeroInCountMax;
    if (not_zeroed) {
        zero_ptr = nullptr;
        zero_bytes = 0;
    }
    gs::timeline_before(s);
    GS_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)tiles, s));
    gs::timeline_after("memset(tile counters)", s);
    const int blocks = gs::persistent_blocks(N);
    GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gs::k_count_tiles),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)gs::kMaxTileLds));
    GS_LAUNCH(gs::k_count_tiles, dim3(blocks), dim3(gs::kPersistentThreads), sizeof(int32_t) * (size_t)tiles, s, N,
              tiles, tiles_x, pk, counts, wg_base, static_cast<uint4 *>(zero_ptr), zero_bytes / 16);
    GS_LAUNCH_CHECK();
    GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gs::k_scatter_scan),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)gs::kMaxTileLds));
    GS_LAUNCH(gs::k_scatter_scan, dim3(blocks + 1), dim3(gs::kPersistentThreads), lds, s, N, tiles, tiles_x, capacity,
              gs::order_multiplier(tiles), pk, depths, counts, wg_base, reinterpret_cast<int2 *>(tile_bins), total_dev,
              num_isects_host, tile_order, keys);
    GS_LAUNCH_CHECK();
    const int rc = gs::launch_tile_sorts(tiles, capacity, list_stats, reinterpret_cast<int2 *>(tile_bins), keys,
                                         gaussian_ids_sorted, block_masks, s, tile_order);
    if (rc != GS_OK) return rc;
    return not_zeroed ? GS_OK_NOT_ZEROED : GS_OK;
}

extern "C" int gs_bin_and_sort(int W, int H, int N, int32_t capacity, const float *packed,
                               const float *depths, int32_t *tile_bins,
                               int32_t *gaussian_ids_sorted, uint16_t *block_masks,
                               int32_t *tile_order, int32_t *num_isects_host, void *workspace,
                               size_t workspace_bytes, gs_stream_t stream) {
    if (!num_isects_host) return GS_ERR_INVALID_ARGUMENT;
    // a reused pinned buffer must never report the previous frame's counts, whatever path the scan takes
    num_isects_host[0] = 0;
    num_isects_host[1] = 0;
    int rc = gs_bin_scan(W, H, N, packed, tile_bins, tile_order, num_isects_host, workspace,
                         workspace_bytes, stream);
    if (rc != GS_OK) return rc;
    GS_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    const int32_t M = *num_isects_host;
    if (M > capacity) return GS_ERR_CAPACITY;
    return gs_bin_sort(W, H, N, M, packed, depths, tile_bins, gaussian_ids_sorted, block_masks,
                       num_isects_host, workspace, workspace_bytes, stream);
}
