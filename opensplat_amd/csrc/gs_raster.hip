// gs_raster.hip — per-tile alpha compositing (forward) and its gradient walk (backward).
//
// Replaces rasterize_forward / rasterize_backward_kernel (reference
// rasterizer/gsplat/forward.cu:256-378, backward.cu:161-355).  Per-pixel recurrence, thresholds
// and clamp constants are the CPU oracle's (rasterizer/gsplat-cpu/gsplat_cpu.cpp:188-240 forward,
// :313-373 backward).  The forward evaluates them in the same operation order without FMA
// contraction and with a glibc-bit-exact expf, so that contributor sets, final_Ts and the image
// equal gsplat-cpu's bit for bit on identical inputs.  The backward makes the same
// contributor decisions (same alpha >= 1/255 outcome as the forward, exactly) but is free to
// re-associate the gradient arithmetic (fp32 atomics make the sum order undefined anyway).
//
// Both kernels are VALU-issue-bound in the bulk of a launch and latency-bound in its tail (rocprofv3
// counters and the issue-cost model in DESIGN.md 4.1; LDS and HBM are far from their limits), so
// the design minimises the cost-weighted instructions per (tile, Gaussian) entry:
//   * one 64-lane wavefront == one workgroup == one 16x16 tile (two or four waves for tiles with
//     very long lists and for the tail of the forward launch, see "work units" below); every lane
//     owns FOUR pixels (column lane&15, rows (lane>>4) + 4k, k = 0..3), handled as TWO packed pairs
//     so that the sigma quadratic form and the gradient body run on v_pk_{mul,add,fma}_f32.  The
//     per-Gaussian record is read from LDS once per wave (broadcast reads of duplicated pairs, see
//     Staged) and amortised over 4 pixels per lane.
//   * no workgroup barriers between waves: the tile's sorted list is staged 64 entries at a time
//     by the wave itself (next chunk's gather prefetched into registers while the current chunk
//     is consumed), early termination is a 64-bit ballot.
//   * per-pixel "is this Gaussian relevant" is ONE unsigned compare of sigma's bit pattern against
//     sigma_max = ln(255*opacity) (precomputed per Gaussian): 0 <= sigma <= sigma_max, with
//     negatives and NaNs failing by construction.  The CPU oracle's pixel-rectangle test is
//     implied by it whenever the rectangle encloses the sigma_max ellipse box (flag bit set by
//     gs_pack_splats); only for the rare Gaussians whose rectangle cuts the ellipse is the
//     rectangle applied explicitly (scalar-branched path injecting NaN coordinates).  Finished /
//     out-of-image pixels carry a NaN row coordinate in the forward, so they fail the compare too.
//   * an 8-row half of the tile that the (tightened) rectangle does not touch is skipped with a
//     scalar branch; the fp64 exponential is only issued when some lane of the 4-row strip passes;
//     a pixel that is skipped gets alpha = 0, which composites exactly nothing — no per-pixel
//     branches; saturation (once per pixel and frame) is a scalar-branched rare path.
//   * backward: the exponential is v_exp_f32, with the exact fp64 evaluation re-run only for
//     lanes whose alpha lies within 2.5e-6 (relative) of the 1/255 threshold, so the decision
//     equals the forward's; 1/(1-alpha) is v_rcp_f32 + one Newton step; the running colour
//     buffer is tracked as its dot product with the pixel's cotangent (one register per pixel
//     instead of three); sigma moments are accumulated per pixel and converted to the nine
//     gradient components once per entry and lane.
//   * the nine partial gradients x 64 lanes are summed through LDS (reduce9: nine conflict-free
//     stores, four 16-byte reads and 15 adds per lane, one DPP quad reduction), after which nine
//     lanes hold the nine totals and ONE global_atomic_add_f32 instruction (nine active lanes
//     hitting one 64-byte gradient record) scatters them — against 9 x (4 DPP + 4 v_readlane +
//     3 add + 1 atomic) in the first version of this kernel.
//
// Roofline: HBM traffic is one 48-byte gather per entry plus 20 B per pixel; DESIGN.md states the
// algorithmic bytes used for roofline.achieved and the VALU accounting.
#include <type_traits>

#include "gs_device.h"

namespace gs {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kChunk = 64;  // entries staged per pass == wave width
constexpr int kGradRec = 16;  // floats per Gaussian in the backward's gradient records (64 B)

// Optional work counters (build with -DGS_STATS; never in the shipped library): per launch totals of
// [0] entries visited  [1] 8-row halves evaluated  [2] halves with a needing lane  [3] exponential
// passes  [4] needing (lane, pixel) pairs  — forward in slots 0-7, backward in slots 8-15.
#ifdef GS_STATS
__device__ unsigned long long g_stats[16];
#define GS_STAT(slot, v)                                                          \
    do {                                                                          \
        if (threadIdx.x == 0) atomicAdd(&g_stats[slot], (unsigned long long)(v)); \
    } while (0)
#else
#define GS_STAT(slot, v) \
    do {                 \
    } while (0)
#endif

// LDS image of one staged entry (96 B, six ds_read_b128).  Everything a packed (2-pixel) instruction
// consumes is stored as a DUPLICATED pair, so that one 64-bit LDS read yields the broadcast
// operand {v, v} directly (the compiler cannot express "same 32-bit register for both halves" on
// v_pk_*_f32 and would spend a v_mov per use otherwise):
//   q0 = {x, x, y, y}  q1 = {A, A, B, B}  q2 = {C, C, opacity, opacity}  q3 = {r, r, g, g}
//   q4 = {b, b, sigma_max|flag, mask bits}  q5 = {id, -, -, -}
struct __attribute__((aligned(16))) Staged {
    f2 xx, yy, AA, BB, CC, oo, rr, gg, bb;
    float smax;
    uint32_t mask;
    int id;
    int pad[3];
};
static_assert(sizeof(Staged) == 96, "staged entry must be six 16-byte words");

// One sorted-list entry as gathered from HBM (13 VGPRs): prefetched one chunk ahead.
struct Rec {
    float4 p0, p1, p2;
    int g;
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

__device__ __forceinline__ void fetch_entry(Rec &r, int idx, const int32_t *__restrict__ ids,
                                            const float4 *__restrict__ packed) {
    r.g = ids[idx];
    r.p0 = packed[3 * (size_t)r.g + 0];
    r.p1 = packed[3 * (size_t)r.g + 1];
    r.p2 = packed[3 * (size_t)r.g + 2];
}

__device__ __forceinline__ void stage_entry(Staged *dst, const Rec &r, int tile_x0, int tile_y0) {
    const uint32_t m = tile_mask(__float_as_uint(r.p1.w), __float_as_uint(r.p2.w), tile_x0, tile_y0);
    float4 *q = reinterpret_cast<float4 *>(dst);
    q[0] = make_float4(r.p0.x, r.p0.x, r.p0.y, r.p0.y);
    q[1] = make_float4(r.p0.z, r.p0.z, r.p0.w, r.p0.w);
    q[2] = make_float4(r.p1.x, r.p1.x, r.p1.y, r.p1.y);
    q[3] = make_float4(r.p2.x, r.p2.x, r.p2.y, r.p2.y);
    q[4] = make_float4(r.p2.z, r.p2.z, r.p1.z, __uint_as_float(m));
    q[5] = make_float4(__int_as_float(r.g), 0.0f, 0.0f, 0.0f);
}

__device__ __forceinline__ float qnan() { return __uint_as_float(0x7fc00000u); }

// Work units.  A wave walking a tile's list alone on its SIMD is latency-bound (~4x slower than
// its share of a saturated SIMD), so (1) a launch ends with stragglers — measured ~100 us of each
// compositing kernel at C2 — and (2) one tile with a very long list (real captures have them) can
// dominate the whole launch.  A tile can therefore be handled by 1, 2 or 4 waves, each owning a
// set of pixel rows (16, 8 or 4 of them) and skipping the entries that miss its rows:
//   * Sched.mult == 1: one wave per tile, except that tiles from `split_from` on (in dispatch
//     order, the tail of the launch) get two half-tile waves;
//   * Sched.mult == 2 / 4 (chosen by the host when the previous frame had a list longer than
//     t2 / t4): `mult` blocks per tile; a tile whose list is longer than t4 is split in four,
//     longer than t2 in two, otherwise block 0 takes the whole tile and the others exit.
// Returns the 16-bit set of tile rows this wave owns (0 = nothing to do); pixel k of a lane
// (rows ly + 4k) is owned iff bit 4k is set.
struct Sched {
    int split_from, mult, t2, t4;
    const int32_t *order;  // tiles by descending list length (mult > 1 only), or NULL
};

__device__ __forceinline__ uint32_t decode_unit(int block, const Sched sc, int num_tiles,
                                                const int2 *__restrict__ bins, int &tile) {
    if (sc.mult <= 1) {
        int lin = block;
        uint32_t rows = 0xFFFFu;
        if (block >= sc.split_from) {
            const int r = block - sc.split_from;
            lin = sc.split_from + (r >> 1);
            rows = 0xFFu << (8 * (r & 1));
        }
        tile = sc.order ? sc.order[lin] : xcd_swizzle(lin, num_tiles);
        return rows;
    }
    // blocks [0, (mult-1)*tiles): parts 1..mult-1 of every tile — almost all of them exit at once,
    // the few that belong to long lists start right at the beginning of the launch; then part 0 of
    // every tile.  (Interleaving the parts tile by tile would leave three quarters of every CU's
    // workgroup slots to blocks that exit, and the real ones latency-bound.)
    const int extra = (sc.mult - 1) * num_tiles;
    const int part = block < extra ? 1 + block / num_tiles : 0;
    const int lin = block < extra ? block % num_tiles : block - extra;
    // longest lists first when the order is known, else the XCD-banded raster order
    tile = sc.order ? sc.order[lin] : xcd_swizzle(lin, num_tiles);
    const int2 range = bins[tile];
    const int n = range.y - range.x;
    const int split = (sc.mult >= 4 && n > sc.t4) ? 4 : (n > sc.t2 ? 2 : 1);
    if (part >= split) return 0u;
    return split == 1 ? 0xFFFFu : split == 2 ? (0xFFu << (8 * part)) : (0xFu << (4 * part));
}

// ---------------------------------------------------------------------------------------------
template <bool EXACT>
__global__ void __launch_bounds__(64)
k_rasterize_forward_v1(int W, int H, int tiles_x, int num_tiles, Sched sched,
                    const int32_t *__restrict__ ids,
                    const int2 *__restrict__ bins, const float4 *__restrict__ packed, float bg0,
                    float bg1, float bg2, const float *__restrict__ bg_dev,
                    float *__restrict__ out_img,
                    float *__restrict__ final_Ts, int32_t *__restrict__ final_idx,
                    float *__restrict__ out_clamped) {
    __shared__ Staged stage[kChunk];
    __shared__ uint64_t exp_tab[kExpTabLds];
    const int lane = threadIdx.x;
    if (bg_dev) {  // background handed over as a device tensor (no host copy, no sync)
        bg0 = bg_dev[0]; bg1 = bg_dev[1]; bg2 = bg_dev[2];
    }
    int tile;
    const uint32_t rows = decode_unit(blockIdx.x, sched, num_tiles, bins, tile);
    if (rows == 0u) return;
    // rows of the tile this wave owns, as a filter on the entries' row-mask bits
    const uint32_t keep = 0xFFFFu | (rows << 16);
    const int tile_x0 = (tile % tiles_x) * GS_TILE, tile_y0 = (tile / tiles_x) * GS_TILE;
    if (EXACT) load_exp_table(exp_tab, lane, 64);

    const int lx = lane & 15, ly = lane >> 4;
    const int px = tile_x0 + lx;
    const f2 pxf2 = (f2)((float)px);
    // py2[h] = row coordinates of pixels k = 2h, 2h+1; NaN once the pixel is finished (or outside
    // the image): a NaN row makes sigma NaN, which fails "0 <= sigma <= sigma_max".
    f2 py2[2], T2[2], acc2[2][3];
    int last[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int py = tile_y0 + ly + 4 * k;
        const bool mine = ((rows >> (4 * k)) & 1u) != 0u;
        const float v = (px < W && py < H && mine) ? (float)py : qnan();
        if (k & 1) py2[k >> 1].y = v; else py2[k >> 1].x = v;
        last[k] = -1;
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
        T2[h] = (f2)(1.0f);
        acc2[h][0] = acc2[h][1] = acc2[h][2] = (f2)(0.0f);
    }
    const uint32_t colbit = 1u << lx;

    const int2 range = bins[tile];
    Rec nxt;
    if (range.x + lane < range.y) fetch_entry(nxt, range.x + lane, ids, packed);
    for (int c0 = range.x; c0 < range.y; c0 += kChunk) {
        const bool alive = (py2[0].x == py2[0].x) || (py2[0].y == py2[0].y) ||
                           (py2[1].x == py2[1].x) || (py2[1].y == py2[1].y);
        if (__builtin_amdgcn_ballot_w64(alive) == 0ull) break;
        __syncthreads();  // previous chunk fully consumed (single-wave workgroup: cheap)
        if (c0 + lane < range.y) stage_entry(&stage[lane], nxt, tile_x0, tile_y0);
        __syncthreads();
        if (c0 + kChunk + lane < range.y) fetch_entry(nxt, c0 + kChunk + lane, ids, packed);
        const int n = min(kChunk, range.y - c0);
        for (int t = 0; t < n; t++) {
            const Staged &e = stage[t];
            const uint32_t mask = __builtin_amdgcn_readfirstlane(e.mask) & keep;
            const uint32_t sbits = __builtin_amdgcn_readfirstlane(__float_as_uint(e.smax));
            const bool rect_binds = (sbits & 1u) != 0u;  // wave-uniform
            f2 dx2 = e.xx - pxf2;  // {xCam, xCam}
            if (rect_binds) {  // rare: keep it a scalar branch
                asm volatile("; rectangle binds");
                if ((mask & colbit) == 0u) dx2 = (f2)(qnan());  // column outside the rectangle
            }
            const f2 Adx = e.AA * dx2;     // A * xCam
            const f2 Adxdx = Adx * dx2;    // A * xCam * xCam
            const f2 Bdx = e.BB * dx2;     // B * xCam
            const f2 yy = e.yy, CC = e.CC, oo = e.oo, rr = e.rr, gg = e.gg, bb = e.bb;
            GS_STAT(0, 1);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (((mask >> (16 + 8 * h)) & 0xFFu) == 0u) continue;  // scalar: half untouched
                GS_STAT(1, 1);
                f2 py = py2[h];
                if (rect_binds) {  // rows outside the rectangle
                    asm volatile("; rectangle binds");
                    if ((mask & (1u << (16 + ly + 8 * h))) == 0u) py.x = qnan();
                    if ((mask & (1u << (16 + ly + 8 * h + 4))) == 0u) py.y = qnan();
                }
                const f2 dy = yy - py;
                // sigma = 0.5f * (A*x*x + C*y*y) + B*x*y, gsplat_cpu.cpp:213-217 (same op order)
                f2 sg = (CC * dy) * dy;
                sg = Adxdx + sg;
                sg = 0.5f * sg;
                sg = sg + Bdx * dy;
                // 0 <= sigma <= sigma_max as ONE unsigned compare of the bit patterns: negative
                // values and NaNs have larger patterns than any sigma_max (< 6).  Only -0.0 would be
                // misjudged; it needs a negative conic entry, which gs_pack_splats routes to the
                // flagged path, where adding +0.0 turns -0.0 into +0.0 first.
                if (rect_binds) sg = sg + (f2)(0.0f);
                const bool need0 = __float_as_uint(sg.x) <= sbits;
                const bool need1 = __float_as_uint(sg.y) <= sbits;
                const uint64_t m0 = __builtin_amdgcn_ballot_w64(__float_as_uint(sg.x) <= sbits);
                const uint64_t m1 = __builtin_amdgcn_ballot_w64(__float_as_uint(sg.y) <= sbits);
                if ((m0 | m1) == 0ull) continue;
                GS_STAT(2, 1);
                GS_STAT(3, (m0 != 0ull) + (m1 != 0ull));
                GS_STAT(4, __builtin_popcountll(m0) + __builtin_popcountll(m1));
                GS_STAT(5, (m0 != 0ull) && (m1 != 0ull) && ((m0 & m1) == 0ull));
                GS_STAT(6, __builtin_popcountll(m0 & m1));
                // exp(-sigma) only where needed; other lanes keep 0 => alpha 0 => no contribution
                f2 vis = (f2)(0.0f);
                if (m0 != 0ull) {
                    if (need0) vis.x = gs_exp<EXACT>(-sg.x, exp_tab);
                }
                if (m1 != 0ull) {
                    if (need1) vis.y = gs_exp<EXACT>(-sg.y, exp_tab);
                }
                // gsplat_cpu.cpp:220-236 for both pixels of the pair:
                //   alpha = min(0.999, opacity*vis); skip if alpha < 1/255; nextT = T*(1-alpha);
                //   nextT <= 1e-4 -> pixel done (Gaussian not rendered); else composite.
                // A skipped pixel is given alpha = 0, which composites exactly nothing
                // (T*(1-0) == T, acc + 0*c == acc), so no per-pixel branch is needed.
                f2 alpha = oo * vis;
                alpha.x = __builtin_amdgcn_fmed3f(alpha.x, 0.0f, 0.999f);
                alpha.y = __builtin_amdgcn_fmed3f(alpha.y, 0.0f, 0.999f);
                bool ok0 = alpha.x >= (1.0f / 255.0f), ok1 = alpha.y >= (1.0f / 255.0f);
                alpha.x = ok0 ? alpha.x : 0.0f;
                alpha.y = ok1 ? alpha.y : 0.0f;
                f2 nT = T2[h] * (1.0f - alpha);
                // a pixel saturating (at most once per pixel and frame): rare, scalar-branched
                if ((__builtin_amdgcn_ballot_w64(nT.x <= 1e-4f) |
                     __builtin_amdgcn_ballot_w64(nT.y <= 1e-4f)) != 0ull) {
                    asm volatile("; pixel saturates");
                    if (nT.x <= 1e-4f) { py2[h].x = qnan(); alpha.x = 0.0f; nT.x = T2[h].x; ok0 = false; }
                    if (nT.y <= 1e-4f) { py2[h].y = qnan(); alpha.y = 0.0f; nT.y = T2[h].y; ok1 = false; }
                }
                const f2 w = alpha * T2[h];
                acc2[h][0] = acc2[h][0] + w * rr;
                acc2[h][1] = acc2[h][1] + w * gg;
                acc2[h][2] = acc2[h][2] + w * bb;
                T2[h] = nT;
                last[2 * h] = ok0 ? (c0 + t) : last[2 * h];
                last[2 * h + 1] = ok1 ? (c0 + t) : last[2 * h + 1];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int py = tile_y0 + ly + 4 * k;
        if (px < W && py < H && ((rows >> (4 * k)) & 1u)) {
            const size_t pix = (size_t)py * W + px;
            const int h = k >> 1;
            const float Tk = (k & 1) ? T2[h].y : T2[h].x;
            const float a0 = (k & 1) ? acc2[h][0].y : acc2[h][0].x;
            const float a1 = (k & 1) ? acc2[h][1].y : acc2[h][1].x;
            const float a2 = (k & 1) ? acc2[h][2].y : acc2[h][2].x;
            const float o0 = a0 + Tk * bg0, o1 = a1 + Tk * bg1, o2 = a2 + Tk * bg2;
            out_img[3 * pix + 0] = o0;
            out_img[3 * pix + 1] = o1;
            out_img[3 * pix + 2] = o2;
            if (out_clamped) {  // fused torch::clamp_max(rgb, 1), model.cpp:222
                out_clamped[3 * pix + 0] = fminf(o0, 1.0f);
                out_clamped[3 * pix + 1] = fminf(o1, 1.0f);
                out_clamped[3 * pix + 2] = fminf(o2, 1.0f);
            }
            final_Ts[pix] = Tk;
            final_idx[pix] = last[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Wave reduction of nine per-lane values through LDS.  Every lane stores its nine values (rows of
// 64 + 4 floats: conflict-free stores), lane 4c + j then adds the 16 values [16j, 16j+16) of row c
// (four 16-byte reads) and a DPP quad reduction leaves the total of component c in lane 4c:
// 15 adds + 2 DPP per lane and 13 LDS instructions per entry.  Measured against the register-only
// alternative (v_permlane32/16_swap folding value pairs across halves / rows + DPP inside rows:
// 8 swaps at ~8 cycles each + ~25 VALU): 632 vs 673 us for the backward kernel at C2.
constexpr int kRedStride = 68;
__device__ __forceinline__ float reduce9(float v0, float v1, float v2, float v3, float v4,
                                             float v5, float v6, float v7, float v8, int lane,
                                             float *red) {
    red[0 * kRedStride + lane] = v0; red[1 * kRedStride + lane] = v1;
    red[2 * kRedStride + lane] = v2; red[3 * kRedStride + lane] = v3;
    red[4 * kRedStride + lane] = v4; red[5 * kRedStride + lane] = v5;
    red[6 * kRedStride + lane] = v6; red[7 * kRedStride + lane] = v7;
    red[8 * kRedStride + lane] = v8;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int c = min(lane >> 2, 8), j = lane & 3;
    const float4 *src = reinterpret_cast<const float4 *>(red + c * kRedStride + 16 * j);
    const float4 a = src[0], b = src[1], d = src[2], e = src[3];
    float r = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) +
              (((d.x + d.y) + (d.z + d.w)) + ((e.x + e.y) + (e.z + e.w)));
    r += dpp_f<0xB1>(r);  // quad_perm [1,0,3,2]
    r += dpp_f<0x4E>(r);  // quad_perm [2,3,0,1]
    __builtin_amdgcn_wave_barrier();  // the next entry's stores come after these loads
    return r;
}
__device__ __forceinline__ int reduce9_role(int lane) {
    return ((lane & 3) == 0 && lane < 36) ? (lane >> 2) : -1;
}

__global__ void __launch_bounds__(64) k_debug_reduce9(const float *__restrict__ in,
                                                      float *__restrict__ out) {
    const int lane = threadIdx.x;
    const float *p = in + (size_t)blockIdx.x * 9 * 64;
    float v[9];
#pragma unroll
    for (int i = 0; i < 9; i++) v[i] = p[i * 64 + lane];
    __shared__ __attribute__((aligned(16))) float red[9 * kRedStride];
    const float r = reduce9(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], lane, red);
    const int role = reduce9_role(lane);
    if (role >= 0) out[(size_t)blockIdx.x * 9 + role] = r;
}

// ---------------------------------------------------------------------------------------------
template <bool EXACT>
__global__ void __launch_bounds__(64)
k_rasterize_backward_v1(int W, int H, int tiles_x, int num_tiles, Sched sched,
                     const int32_t *__restrict__ ids,
                     const int2 *__restrict__ bins, const float4 *__restrict__ packed, float bg0,
                     float bg1, float bg2, const float *__restrict__ bg_dev,
                     const float *__restrict__ final_Ts,
                     const int32_t *__restrict__ final_idx, const float *__restrict__ v_out,
                     const float *__restrict__ v_out_alpha, const float *__restrict__ img_raw,
                     float *__restrict__ gacc) {
    __shared__ Staged stage[kChunk];
    const int lane = threadIdx.x;
    if (bg_dev) {
        bg0 = bg_dev[0]; bg1 = bg_dev[1]; bg2 = bg_dev[2];
    }
    int tile;
    const uint32_t rows = decode_unit(blockIdx.x, sched, num_tiles, bins, tile);
    if (rows == 0u) return;
    const uint32_t keep = 0xFFFFu | (rows << 16);
    const int tile_x0 = (tile % tiles_x) * GS_TILE, tile_y0 = (tile / tiles_x) * GS_TILE;

    const int lx = lane & 15, ly = lane >> 4;
    const int px = tile_x0 + lx;
    const f2 pxf2 = (f2)((float)px);
    // per pixel pair h (pixels k = 2h, 2h+1): row coordinate, transmittance being unwound,
    // T_final * (v_out_alpha - bg . v_out), running <colour buffer, v_out>, cotangent
    f2 py2[2], T2[2], TW2[2], bv2[2], vo2[2][3];
    int last[4];
    int my_last = -1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int py = tile_y0 + ly + 4 * k;
        float Tfin = 1.0f, o0 = 0.0f, o1 = 0.0f, o2 = 0.0f, oa = 0.0f;
        int l = -1;
        if (px < W && py < H && ((rows >> (4 * k)) & 1u)) {
            const size_t pix = (size_t)py * W + px;
            Tfin = final_Ts[pix];
            l = final_idx[pix];
            o0 = v_out[3 * pix + 0];
            o1 = v_out[3 * pix + 1];
            o2 = v_out[3 * pix + 2];
            if (img_raw) {  // backward of the fused clamp_max(rgb, 1): torch passes where rgb <= 1
                if (!(img_raw[3 * pix + 0] <= 1.0f)) o0 = 0.0f;
                if (!(img_raw[3 * pix + 1] <= 1.0f)) o1 = 0.0f;
                if (!(img_raw[3 * pix + 2] <= 1.0f)) o2 = 0.0f;
            }
            oa = v_out_alpha ? v_out_alpha[pix] : 0.0f;
        }
        const float tw = Tfin * (oa - (bg0 * o0 + bg1 * o1 + bg2 * o2));
        const int h = k >> 1;
        if (k & 1) {
            py2[h].y = (float)py; T2[h].y = Tfin; TW2[h].y = tw;
            vo2[h][0].y = o0; vo2[h][1].y = o1; vo2[h][2].y = o2;
        } else {
            py2[h].x = (float)py; T2[h].x = Tfin; TW2[h].x = tw;
            vo2[h][0].x = o0; vo2[h][1].x = o1; vo2[h][2].x = o2;
        }
        last[k] = l;
        my_last = max(my_last, l);
    }
    bv2[0] = (f2)(0.0f);
    bv2[1] = (f2)(0.0f);

    // lanes that hold a reduced total scatter it into the Gaussian's 64-byte gradient record
    // gacc[g][0..8] = {v_x, v_y, v_A, v_B, v_C, v_r, v_g, v_b, v_opacity}: the nine lanes of the one
    // atomic instruction hit ONE cache line, which the L2 serves ~4x faster than nine lines
    // (measured, scripts/ubench/atomics.hip); k_unpack_grads splits the records afterwards.
    __shared__ __attribute__((aligned(16))) float red[9 * kRedStride];
    const int role = reduce9_role(lane);
    float *wbase = gacc + (role >= 0 ? role : 0);
    const uint32_t colbit = 1u << lx;

    const int2 range = bins[tile];
    const int wave_last = wave_max_i(my_last);  // last list entry any pixel of the tile used
    if (wave_last < range.x) return;            // (also covers empty tiles / no contributors)

    // walk the list back to front in chunks; slot 0 of a chunk is its furthest-back entry
    Rec nxt;
    if (wave_last - lane >= range.x) fetch_entry(nxt, wave_last - lane, ids, packed);
    for (int hi = wave_last; hi >= range.x; hi -= kChunk) {
        __syncthreads();
        if (hi - lane >= range.x) stage_entry(&stage[lane], nxt, tile_x0, tile_y0);
        __syncthreads();
        if (hi - kChunk - lane >= range.x) fetch_entry(nxt, hi - kChunk - lane, ids, packed);
        const int n = min(kChunk, hi - range.x + 1);
        // One entry; BINDS (compile-time) as in the forward kernel.
        auto entry = [&](int t, uint32_t mask, uint32_t sbits, auto binds_tag) {
            constexpr bool rect_binds = decltype(binds_tag)::value;
            const Staged &en = stage[t];
            const int e = hi - t;  // index of this entry in the sorted list
            const f2 dx2 = en.xx - pxf2;
            const float dx = dx2.x;
            // sigma is evaluated from copies of dx / dy that are NaN outside the rectangle when
            // the rectangle binds (a NaN sigma fails both compares); the moments use the real ones
            f2 dxs = dx2;
            if (rect_binds) {
                if ((mask & colbit) == 0u) dxs = (f2)(qnan());
            }
            const f2 Adx = en.AA * dxs, Adxdx = Adx * dxs, Bdx = en.BB * dxs;
            const f2 yy = en.yy, CC = en.CC, oo = en.oo, crr = en.rr, cgg = en.gg, cbb = en.bb;
            f2 s0 = (f2)(0.0f), s1 = (f2)(0.0f), s2 = (f2)(0.0f);
            f2 gr = (f2)(0.0f), gg = (f2)(0.0f), gb = (f2)(0.0f);
            bool any = false;  // wave-uniform
            GS_STAT(8, 1);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (((mask >> (16 + 8 * h)) & 0xFFu) == 0u) continue;  // scalar: half untouched
                GS_STAT(9, 1);
                f2 pys = py2[h];
                if (rect_binds) {
                    if ((mask & (1u << (16 + ly + 8 * h))) == 0u) pys.x = qnan();
                    if ((mask & (1u << (16 + ly + 8 * h + 4))) == 0u) pys.y = qnan();
                }
                const f2 dys = yy - pys;
                f2 dy = dys;
                if (rect_binds) dy = yy - py2[h];
                f2 sg = (CC * dys) * dys;
                sg = Adxdx + sg;
                sg = 0.5f * sg;
                sg = sg + Bdx * dys;
                // 0 <= sigma <= sigma_max as one unsigned compare (see the forward kernel)
                if (rect_binds) sg = sg + (f2)(0.0f);
                const bool need0 = (e <= last[2 * h]) && (__float_as_uint(sg.x) <= sbits);
                const bool need1 = (e <= last[2 * h + 1]) && (__float_as_uint(sg.y) <= sbits);
                // (ballots of the individual compares: they stay in SGPRs, no VALU round trip)
                const uint64_t m0 = __builtin_amdgcn_ballot_w64(e <= last[2 * h]) &
                                    __builtin_amdgcn_ballot_w64(__float_as_uint(sg.x) <= sbits);
                const uint64_t m1 = __builtin_amdgcn_ballot_w64(e <= last[2 * h + 1]) &
                                    __builtin_amdgcn_ballot_w64(__float_as_uint(sg.y) <= sbits);
                if ((m0 | m1) == 0ull) continue;
                any = true;
                GS_STAT(10, 1);
                GS_STAT(12, __builtin_popcountll(m0) + __builtin_popcountll(m1));
                // vis = exp(-sigma), alpha = min(0.99, opacity * vis), gsplat_cpu.cpp:337-338;
                // lanes that do not take part end up with vis = alpha = 0
                f2 vis;
                vis.x = need0 ? __expf(-sg.x) : 0.0f;
                vis.y = need1 ? __expf(-sg.y) : 0.0f;
                f2 alpha = oo * vis;
                if (EXACT) {
                    // same >= 1/255 decision as the forward: redo the exponential exactly where
                    // the fast one cannot decide (|rel. distance to the threshold| < 2.5e-6)
                    const float thr = 1.0f / 255.0f;
                    const bool amb0 = need0 && fabsf(alpha.x - thr) < 1.0e-8f;
                    const bool amb1 = need1 && fabsf(alpha.y - thr) < 1.0e-8f;
                    const uint64_t ma =
                        (m0 & __builtin_amdgcn_ballot_w64(fabsf(alpha.x - thr) < 1.0e-8f)) |
                        (m1 & __builtin_amdgcn_ballot_w64(fabsf(alpha.y - thr) < 1.0e-8f));
                    if (ma != 0ull) {
                        if (amb0) { vis.x = expf_glibc_cmem(-sg.x); alpha.x = oo.x * vis.x; }
                        if (amb1) { vis.y = expf_glibc_cmem(-sg.y); alpha.y = oo.x * vis.y; }
                    }
                }
                const bool ok0 = alpha.x >= (1.0f / 255.0f);
                const bool ok1 = alpha.y >= (1.0f / 255.0f);
                alpha.x = ok0 ? __builtin_amdgcn_fmed3f(alpha.x, 0.0f, 0.99f) : 0.0f;
                alpha.y = ok1 ? __builtin_amdgcn_fmed3f(alpha.y, 0.0f, 0.99f) : 0.0f;
                vis.x = ok0 ? vis.x : 0.0f;
                vis.y = ok1 ? vis.y : 0.0f;
                // ra = 1 / (1 - alpha): hardware reciprocal + one Newton step
                const f2 om = 1.0f - alpha;
                f2 ra;
                ra.x = __builtin_amdgcn_rcpf(om.x);
                ra.y = __builtin_amdgcn_rcpf(om.y);
                const f2 er = __builtin_elementwise_fma(-om, ra, (f2)(1.0f));
                ra = __builtin_elementwise_fma(ra, er, ra);
                T2[h] = T2[h] * ra;               // transmittance in front of this Gaussian
                const f2 fac = alpha * T2[h];
                gr = __builtin_elementwise_fma(fac, vo2[h][0], gr);
                gg = __builtin_elementwise_fma(fac, vo2[h][1], gg);
                gb = __builtin_elementwise_fma(fac, vo2[h][2], gb);
                // cv = <colour, v_out>;  v_alpha = T*cv + ra*(T_final*w - <buffer, v_out>)
                f2 cv = crr * vo2[h][0];
                cv = __builtin_elementwise_fma(cgg, vo2[h][1], cv);
                cv = __builtin_elementwise_fma(cbb, vo2[h][2], cv);
                const f2 v_alpha = __builtin_elementwise_fma(T2[h], cv, ra * (TW2[h] - bv2[h]));
                bv2[h] = __builtin_elementwise_fma(fac, cv, bv2[h]);
                // u = vis * v_alpha; v_sigma = -opacity * u (applied once per entry below)
                const f2 u = vis * v_alpha;
                s0 = s0 + u;
                const f2 ud = u * dy;
                s1 = s1 + ud;
                s2 = __builtin_elementwise_fma(ud, dy, s2);
            }
            if (!any) return;
            GS_STAT(11, 1);
            // per-lane conversion of the moments to the nine gradient components
            const float S0 = s0.x + s0.y, S1 = s1.x + s1.y, S2 = s2.x + s2.y;
            const float mo = -oo.x;
            const float vs0 = mo * S0;         // sum v_sigma
            const float vs1 = mo * S1;         // sum v_sigma * dy
            const float vs2 = mo * S2;         // sum v_sigma * dy^2
            const float mx = dx * vs0;         // sum v_sigma * dx
            const float g_x = fmaf(en.AA.x, mx, en.BB.x * vs1);  // v_sigma * (A dx + B dy)
            const float g_y = fmaf(en.BB.x, mx, CC.x * vs1);     // v_sigma * (B dx + C dy)
            const float hdx = 0.5f * dx;
            const float g_A = hdx * mx;        // 0.5 * v_sigma * dx^2
            const float g_B = hdx * vs1;       // 0.5 * v_sigma * dx * dy   (gsplat_cpu.cpp:361-363)
            const float g_C = 0.5f * vs2;      // 0.5 * v_sigma * dy^2
            const float r = reduce9(g_x, g_y, g_A, g_B, g_C, gr.x + gr.y, gg.x + gg.y,
                                    gb.x + gb.y, S0, lane, red);
            if (role >= 0) {
                atomicAdd(wbase + (size_t)en.id * kGradRec, r);
            }
        };
        for (int t = 0; t < n; t++) {
            const uint32_t mask = __builtin_amdgcn_readfirstlane(stage[t].mask) & keep;
            const uint32_t sbits = __builtin_amdgcn_readfirstlane(__float_as_uint(stage[t].smax));
            if (sbits & 1u) entry(t, mask, sbits, std::true_type{});   // wave-uniform, rare
            else entry(t, mask, sbits, std::false_type{});
        }
    }
}

// Splits the 64-byte gradient records into the four tensors the operator surface returns
// (rasterize_gaussians.cpp:113-124): v_xy[N,2] v_conic[N,3] v_colors[N,3] v_opacity[N].
__global__ void __launch_bounds__(256)
k_unpack_grads(int N, const float4 *__restrict__ gacc, const float4 *__restrict__ packed_logit,
               float *__restrict__ v_xy, float *__restrict__ v_conic, float *__restrict__ v_colors,
               float *__restrict__ v_opacity) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float4 a = gacc[4 * (size_t)n + 0], b = gacc[4 * (size_t)n + 1];
    float o = reinterpret_cast<const float *>(gacc)[kGradRec * (size_t)n + 8];
    if (packed_logit) {  // opacity = sigmoid(logit): d/dlogit = s (1 - s), model.cpp:215
        const float sg = packed_logit[3 * (size_t)n + 1].y;
        o *= sg * (1.0f - sg);
    }
    v_xy[2 * (size_t)n + 0] = a.x;
    v_xy[2 * (size_t)n + 1] = a.y;
    v_conic[3 * (size_t)n + 0] = a.z;
    v_conic[3 * (size_t)n + 1] = a.w;
    v_conic[3 * (size_t)n + 2] = b.x;
    v_colors[3 * (size_t)n + 0] = b.y;
    v_colors[3 * (size_t)n + 1] = b.z;
    v_colors[3 * (size_t)n + 2] = b.w;
    v_opacity[n] = o;
}

// Test hook: the exponential exactly as the compositing kernels evaluate it.
template <bool EXACT>
__global__ void __launch_bounds__(256) k_debug_expf(int64_t n, const float *__restrict__ x,
                                                    float *__restrict__ y) {
    __shared__ uint64_t exp_tab[kExpTabLds];
    load_exp_table(exp_tab, threadIdx.x, 256);
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

#ifdef GS_STATS
extern "C" int gs_debug_stats(unsigned long long *host16, int reset) {
    GS_HIP_CHECK(hipDeviceSynchronize());
    if (host16) GS_HIP_CHECK(hipMemcpyFromSymbol(host16, HIP_SYMBOL(gs::g_stats), 16 * 8));
    if (reset) {
        unsigned long long z[16] = {0};
        GS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(gs::g_stats), z, 16 * 8));
    }
    return GS_OK;
}
#endif

namespace gs {
// Launch schedule (see decode_unit).  list_stats = { M, longest tile list } as gs_bin_scan reported
// them for this or an EARLIER frame (host memory; may be NULL or stale — it only steers how many
// waves share a tile, every choice renders the same image).
// Tail split, measured at C2 (scripts/ab_raster.py): forward 466 us unsplit, 447 us with the last
// 20 % of the tiles split, 463 at 40 %, 531 at 100 %; the backward only loses (733 / 739 / 757 /
// 866 us) because its per-entry reduction and atomic are paid by both halves — 20 % / 0 %.
// Experimental overrides for A/B runs: flags bits 8..15 = tail percent + 1, bits 16..17 = mult.
// At C3 (32 400 tiles, 587 entries per tile) the tail is a smaller share of the launch: 10 % is best for
// BOTH kernels there (forward 2155 us vs 2185 at 20 % / 2200 unsplit; backward 2839 vs 2869 unsplit).
constexpr int kSplitPercentForward = 20, kSplitPercentBackward = 0, kSplitPercentManyTiles = 10;
constexpr int kManyTiles = 16384;
static inline Sched make_sched(int tiles, uint32_t flags, int tail_pct, const int32_t *list_stats,
                               const int32_t *tile_order, int &units) {
    Sched sc;
    sc.order = tile_order;
    const int o = (int)((flags >> 8) & 0xFFu);
    if (o) tail_pct = o - 1;
    if (tail_pct > 100) tail_pct = 100;
    sc.split_from = tiles - (int)((int64_t)tiles * tail_pct / 100);
    sc.mult = 1;
    sc.t2 = sc.t4 = 0x7fffffff;
    if (list_stats && list_stats[0] > 0) {
        const int64_t avg = list_stats[0] / tiles > 64 ? list_stats[0] / tiles : 64;
        const int64_t t2 = 3 * avg + 256, t4 = 6 * avg + 512;
        sc.t2 = (int)(t2 < 0x7fffffff ? t2 : 0x7fffffff);
        sc.t4 = (int)(t4 < 0x7fffffff ? t4 : 0x7fffffff);
        sc.mult = list_stats[1] > sc.t4 ? 4 : (list_stats[1] > sc.t2 ? 2 : 1);
    }
    const int force = (int)((flags >> 16) & 3u);
    if (force) {  // experiment: force the multiplicity, every tile split
        sc.mult = force == 3 ? 4 : force;
        sc.t2 = sc.mult >= 2 ? 0 : 0x7fffffff;
        sc.t4 = sc.mult >= 4 ? 0 : 0x7fffffff;
    }
    units = sc.mult > 1 ? sc.mult * tiles : sc.split_from + 2 * (tiles - sc.split_from);
    return sc;
}
}  // namespace gs

// Test/measurement hook: HIP events recorded immediately before and after the NEXT compositing
// kernel launched by this thread (k_rasterize_forward or k_rasterize_backward alone, without the
// memset / record-splitting kernels around it) — bench.py's roofline.achieved uses it.
namespace gs {
static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
static inline void ev_before(hipStream_t s) {
    if (g_ev_start) (void)hipEventRecord(g_ev_start, s);
}
static inline void ev_after(hipStream_t s) {
    if (g_ev_stop) (void)hipEventRecord(g_ev_stop, s);
    g_ev_start = g_ev_stop = nullptr;
}
}  // namespace gs

extern "C" int gs_debug_time_next_kernel(void *event_start, void *event_stop) {
    gs::g_ev_start = (hipEvent_t)event_start;
    gs::g_ev_stop = (hipEvent_t)event_stop;
    return GS_OK;
}

extern "C" int gs_debug_reduce9(int blocks, const float *in, float *out, gs_stream_t stream) {
    if (blocks < 0) return GS_ERR_INVALID_ARGUMENT;
    if (blocks == 0) return GS_OK;
    if (!in || !out) return GS_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(gs::k_debug_reduce9, dim3(blocks), dim3(64), 0, (hipStream_t)stream, in, out);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" int gs_rasterize_forward_v1(int W, int H, const int32_t *gaussian_ids_sorted,
                                    const int32_t *tile_bins, const float *packed,
                                    const float *background, float *out_img, float *final_Ts,
                                    int32_t *final_idx, float *out_img_clamped,
                                    const int32_t *list_stats, const int32_t *tile_order,
                                    uint32_t flags, gs_stream_t stream) {
    if (W <= 0 || H <= 0) return GS_ERR_INVALID_ARGUMENT;
    if ((flags & GS_FLAG_CLAMP_IMAGE) && !out_img_clamped) return GS_ERR_INVALID_ARGUMENT;
    float *clamped = (flags & GS_FLAG_CLAMP_IMAGE) ? out_img_clamped : nullptr;
    if (W > 65535 || H > 65535) return GS_ERR_UNSUPPORTED;
    if (!tile_bins || !background || !out_img || !final_Ts || !final_idx)
        return GS_ERR_INVALID_ARGUMENT;
    if ((uintptr_t)packed & 15u) return GS_ERR_INVALID_ARGUMENT;
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    const int tiles = tiles_x * tiles_y;
    hipStream_t s = (hipStream_t)stream;
    const int2 *bins = reinterpret_cast<const int2 *>(tile_bins);
    const float4 *pk = reinterpret_cast<const float4 *>(packed);
    int units;
    const gs::Sched sched = gs::make_sched(tiles, flags, tiles >= gs::kManyTiles ? gs::kSplitPercentManyTiles : gs::kSplitPercentForward,
                                           list_stats, tile_order, units);
    const float *bg_dev = gs::on_device(background) ? background : nullptr;
    const float bg0 = bg_dev ? 0.f : background[0], bg1 = bg_dev ? 0.f : background[1],
                bg2 = bg_dev ? 0.f : background[2];
    gs::ev_before(s);
    if (flags & GS_FLAG_FAST_EXP)
        hipLaunchKernelGGL((gs::k_rasterize_forward_v1<false>), dim3(units), dim3(64), 0, s, W, H,
                           tiles_x, tiles, sched, gaussian_ids_sorted, bins, pk, bg0, bg1, bg2, bg_dev, out_img, final_Ts, final_idx, clamped);
    else
        hipLaunchKernelGGL((gs::k_rasterize_forward_v1<true>), dim3(units), dim3(64), 0, s, W, H,
                           tiles_x, tiles, sched, gaussian_ids_sorted, bins, pk, bg0, bg1, bg2, bg_dev, out_img, final_Ts, final_idx, clamped);
    gs::ev_after(s);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" size_t gs_rasterize_backward_workspace_bytes(int N) {
    return N > 0 ? (size_t)N * gs::kGradRec * sizeof(float) : 0;
}

extern "C" int gs_rasterize_backward_v1(int W, int H, int N, const int32_t *gaussian_ids_sorted,
                                     const int32_t *tile_bins, const float *packed,
                                     const float *background, const float *final_Ts,
                                     const int32_t *final_idx, const float *v_out,
                                     const float *v_out_alpha, const float *out_img, float *v_xy,
                                     float *v_conic, float *v_colors, float *v_opacity,
                                     void *workspace, size_t workspace_bytes,
                                     const int32_t *list_stats, const int32_t *tile_order,
                                     uint32_t flags, gs_stream_t stream) {
    if (W <= 0 || H <= 0 || N < 0) return GS_ERR_INVALID_ARGUMENT;
    if ((flags & GS_FLAG_CLAMP_IMAGE) && !out_img) return GS_ERR_INVALID_ARGUMENT;
    const float *img_raw = (flags & GS_FLAG_CLAMP_IMAGE) ? out_img : nullptr;
    if (W > 65535 || H > 65535) return GS_ERR_UNSUPPORTED;
    if (N == 0) return GS_OK;
    const bool keep_records = (flags & GS_FLAG_KEEP_RECORDS) != 0u;
    if (!tile_bins || !background || !final_Ts || !final_idx || !v_out || !workspace)
        return GS_ERR_INVALID_ARGUMENT;
    if (!keep_records && (!v_xy || !v_conic || !v_colors || !v_opacity)) return GS_ERR_INVALID_ARGUMENT;
    if (((uintptr_t)packed & 15u) || ((uintptr_t)workspace & 63u)) return GS_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < gs_rasterize_backward_workspace_bytes(N)) return GS_ERR_WORKSPACE;
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    const int tiles = tiles_x * tiles_y;
    hipStream_t s = (hipStream_t)stream;
    const int2 *bins = reinterpret_cast<const int2 *>(tile_bins);
    const float4 *pk = reinterpret_cast<const float4 *>(packed);
    float *gacc = static_cast<float *>(workspace);
    if (!(flags & GS_FLAG_RECORDS_ZEROED))
        GS_HIP_CHECK(hipMemsetAsync(gacc, 0, gs_rasterize_backward_workspace_bytes(N), s));
    int units;
    const gs::Sched sched = gs::make_sched(tiles, flags, tiles >= gs::kManyTiles ? gs::kSplitPercentManyTiles : gs::kSplitPercentBackward,
                                           list_stats, tile_order, units);
    const float *bg_dev = gs::on_device(background) ? background : nullptr;
    const float bg0 = bg_dev ? 0.f : background[0], bg1 = bg_dev ? 0.f : background[1],
                bg2 = bg_dev ? 0.f : background[2];
    gs::ev_before(s);
    if (flags & GS_FLAG_FAST_EXP)
        hipLaunchKernelGGL(gs::k_rasterize_backward_v1<false>, dim3(units), dim3(64), 0, s, W, H,
                           tiles_x, tiles, sched, gaussian_ids_sorted, bins, pk, bg0, bg1, bg2, bg_dev, final_Ts, final_idx, v_out, v_out_alpha,
                           img_raw, gacc);
    else
        hipLaunchKernelGGL(gs::k_rasterize_backward_v1<true>, dim3(units), dim3(64), 0, s, W, H,
                           tiles_x, tiles, sched, gaussian_ids_sorted, bins, pk, bg0, bg1, bg2, bg_dev, final_Ts, final_idx, v_out, v_out_alpha,
                           img_raw, gacc);
    gs::ev_after(s);
    GS_LAUNCH_CHECK();
    if (keep_records) return GS_OK;  // the 64-byte records go straight to gs_gaussian_backward
    hipLaunchKernelGGL(gs::k_unpack_grads, dim3((N + 255) / 256), dim3(256), 0, s, N,
                       reinterpret_cast<const float4 *>(gacc),
                       (flags & GS_FLAG_LOGIT_OPACITY) ? pk : nullptr, v_xy, v_conic, v_colors,
                       v_opacity);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

// =============================================================================================
// Version 2 of the compositing kernels: QUADRANT WAVES WITH PER-GROUP LIST WALKS.
//
// Round-1 measurement: with one wave per 16x16 tile and every entry of the tile's
// list evaluated by all 64 lanes, only ~16 of 64 lanes are live in an exponential pass — a
// ~30-pixel footprint on a 256-pixel tile.  Here a wave owns one 8x8 QUADRANT of a tile with ONE
// pixel per lane, and each of its four 16-lane groups (one DPP row = one 4x4 pixel block) walks
// ITS OWN list: the entries of the staged chunk whose (tightened) rectangle touches that block.
// The four groups execute the same instruction stream on four different Gaussians, so a step
// evaluates four (block, Gaussian) pairs with ~60 % live lanes instead of one (tile, Gaussian)
// pair with ~25 %:
//   * per chunk of 64 list entries (staged by the wave itself: lane t gathers entry t), four
//     ballots give four 64-bit "touches block g" masks; they live in SGPRs, the walk is scalar
//     (s_ff1 + s_bitset0 per group), the chunk takes max_g popcount(mask_g) steps;
//   * a group without work reads a sentinel record whose x is NaN (sigma NaN fails the one
//     unsigned compare "0 <= sigma <= sigma_max", as a finished pixel's NaN row does);
//   * the per-entry record is read from LDS with a per-group address (48 bytes, three
//     ds_read_b128; identical addresses inside a group broadcast);
//   * four waves per tile, each staging the tile's list for itself: no workgroup barriers, 4x the
//     gather traffic out of L2 (the four quadrant waves of a tile occupy consecutive positions in
//     ONE XCD's block stream); a long list is automatically shared by four waves;
//   * backward: the nine partial sums are reduced per GROUP through LDS (the same nine stores /
//     four 16-byte loads as round 1, without the final cross-group step), added to per-entry
//     accumulators in LDS (ds_add_f32), and flushed once per chunk with one atomic lane per
//     (entry, component): a Gaussian costs one global atomic line-request per quadrant it
//     contributes to, not one per 4x4 block.  The moments sum(u), sum(u dx), .. are reduced, the
//     conversion to (v_x, v_y, v_A, v_B, v_C) happens once per entry at the flush.
// Arithmetic of the forward: unchanged op order (gsplat_cpu.cpp:213-236), bit-exact.
// =============================================================================================
namespace gs {

struct __attribute__((aligned(16))) SRec {
    float4 p0, p1, p2;  // the packed record as gathered: {x y A B | C o smax rx | r g b ry}
};

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// block -> (tile, quadrant).  Blocks round-robin over the 8 XCDs (block b runs on XCD b % 8): the
// four quadrant waves of a tile take consecutive positions of one XCD's stream, tiles are dealt to
// the XCDs in launch order (longest list first when `order` is given).
__device__ __forceinline__ bool decode_quadrant(int block, int num_tiles, int tiles_x, int W, int H,
                                                const int32_t *__restrict__ order, int &tile,
                                                int &qx0, int &qy0) {
    const int x = block & 7, k = block >> 3;
    const int quad = k & 3;
    const int slot = ((k >> 2) << 3) + x;
    if (slot >= num_tiles) return false;
    tile = order ? order[slot] : xcd_swizzle(slot, num_tiles);
    qx0 = (tile % tiles_x) * GS_TILE + 8 * (quad & 1);
    qy0 = (tile / tiles_x) * GS_TILE + 8 * (quad >> 1);
    return qx0 < W && qy0 < H;
}

// which of the quadrant's four 4x4 blocks (bit g: block column g & 1, block row g >> 1) the
// rectangle [x0,x1) x [y0,y1) (packed as x0 | x1 << 16, y0 | y1 << 16) touches
__device__ __forceinline__ uint32_t block_touch(uint32_t rx, uint32_t ry, int qx0, int qy0) {
    const int x0 = (int)(rx & 0xFFFF) - qx0, x1 = (int)(rx >> 16) - qx0;
    const int y0 = (int)(ry & 0xFFFF) - qy0, y1 = (int)(ry >> 16) - qy0;
    if (x1 <= x0 || y1 <= y0) return 0u;
    const bool c0 = x0 < 4 && x1 > 0, c1 = x0 < 8 && x1 > 4;
    const bool r0 = y0 < 4 && y1 > 0, r1 = y0 < 8 && y1 > 4;
    return (c0 && r0 ? 1u : 0u) | (c1 && r0 ? 2u : 0u) | (c0 && r1 ? 4u : 0u) |
           (c1 && r1 ? 8u : 0u);
}

__device__ __forceinline__ void stage_sentinel(SRec *s) {
    s->p0 = make_float4(qnan(), 0.0f, 0.0f, 0.0f);
    s->p1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    s->p2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// next entry of each group's walk; an exhausted group gets the sentinel slot
#define GS_WALK_STEP(m, e)                                   \
    const int e = (m) != 0ull ? (int)__builtin_ctzll(m) : kChunk; \
    (m) &= (m)-1ull;

// ---------------------------------------------------------------------------------------------
template <bool EXACT>
__global__ void __launch_bounds__(64)
k_rasterize_forward(int W, int H, int tiles_x, int num_tiles, const int32_t *__restrict__ order,
                    const int32_t *__restrict__ ids, const int2 *__restrict__ bins,
                    const float4 *__restrict__ packed, float bg0, float bg1, float bg2,
                    const float *__restrict__ bg_dev, float *__restrict__ out_img,
                    float *__restrict__ final_Ts, int32_t *__restrict__ final_idx,
                    float *__restrict__ out_clamped) {
    __shared__ SRec stage[kChunk + 1];
    __shared__ uint64_t exp_tab[EXACT ? kExpTabLds : 1];
    const int lane = threadIdx.x;
    int tile, qx0, qy0;
    if (!decode_quadrant(blockIdx.x, num_tiles, tiles_x, W, H, order, tile, qx0, qy0)) return;
    if (bg_dev) {  // background handed over as a device tensor (no host copy, no sync)
        bg0 = bg_dev[0]; bg1 = bg_dev[1]; bg2 = bg_dev[2];
    }
    if (EXACT) load_exp_table(exp_tab, lane, 64);
    if (lane == 0) stage_sentinel(&stage[kChunk]);

    const int grp = lane >> 4, li = lane & 15;
    const uint32_t gsh = 8u * (uint32_t)grp;
    const int px = qx0 + 4 * (grp & 1) + (li & 3), py = qy0 + 4 * (grp >> 1) + (li >> 2);
    const bool inimg = px < W && py < H;
    const float pxf = (float)px;
    // NaN once the pixel is finished (or outside the image): a NaN row makes sigma NaN
    float pyf = inimg ? (float)py : qnan();
    float T = 1.0f, a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    int last = -1;

    const int2 range = bins[tile];
    // the next chunk's entry of this lane, gathered one chunk ahead (registers, not a struct: a
    // conditionally filled aggregate ends up in scratch)
    float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0;
    if (range.x + lane < range.y) {
        const size_t g = (size_t)ids[range.x + lane];
        n0 = packed[3 * g + 0]; n1 = packed[3 * g + 1]; n2 = packed[3 * g + 2];
    }
    for (int c0 = range.x; c0 < range.y; c0 += kChunk) {
        const uint64_t alive = __builtin_amdgcn_ballot_w64(pyf == pyf);
        if (alive == 0ull) break;
        __syncthreads();  // previous chunk fully consumed (single-wave workgroup: cheap)
        uint32_t touch = 0u;
        if (c0 + lane < range.y) {
            stage[lane].p0 = n0;
            stage[lane].p1 = n1;
            stage[lane].p2 = n2;
            touch = block_touch(__float_as_uint(n1.w), __float_as_uint(n2.w), qx0, qy0);
        }
        // a block whose 16 pixels are all finished walks nothing
        uint64_t m0 = (alive & 0x000000000000FFFFull) ? __builtin_amdgcn_ballot_w64((touch & 1u) != 0u) : 0ull;
        uint64_t m1 = (alive & 0x00000000FFFF0000ull) ? __builtin_amdgcn_ballot_w64((touch & 2u) != 0u) : 0ull;
        uint64_t m2 = (alive & 0x0000FFFF00000000ull) ? __builtin_amdgcn_ballot_w64((touch & 4u) != 0u) : 0ull;
        uint64_t m3 = (alive & 0xFFFF000000000000ull) ? __builtin_amdgcn_ballot_w64((touch & 8u) != 0u) : 0ull;
        __syncthreads();
        GS_STAT(3, __builtin_popcountll(m0) + __builtin_popcountll(m1) + __builtin_popcountll(m2) + __builtin_popcountll(m3));
        GS_STAT(4, 1);
        if (c0 + kChunk + lane < range.y) {
            const size_t g = (size_t)ids[c0 + kChunk + lane];
            n0 = packed[3 * g + 0]; n1 = packed[3 * g + 1]; n2 = packed[3 * g + 2];
        }
        while ((m0 | m1 | m2 | m3) != 0ull) {
            GS_WALK_STEP(m0, e0)
            GS_WALK_STEP(m1, e1)
            GS_WALK_STEP(m2, e2)
            GS_WALK_STEP(m3, e3)
            // the four slots packed into one SGPR; each lane extracts its group's (one v_bfe_u32)
            const uint32_t ep = (uint32_t)e0 | ((uint32_t)e1 << 8) | ((uint32_t)e2 << 16) | ((uint32_t)e3 << 24);
            const int e = (int)((ep >> gsh) & 0xFFu);
            const float4 q0 = stage[e].p0, q1 = stage[e].p1, q2 = stage[e].p2;
            const uint32_t sbits = __float_as_uint(q1.z);
            GS_STAT(0, 1);
            const float dx = q0.x - pxf, dy = q0.y - pyf;
            // sigma = 0.5f * (A*x*x + C*y*y) + B*x*y, gsplat_cpu.cpp:213-217 (same op order)
            float sg = (q0.z * dx) * dx + (q1.x * dy) * dy;
            sg = 0.5f * sg;
            sg = sg + (q0.w * dx) * dy;
            if (__builtin_amdgcn_ballot_w64((sbits & 1u) != 0u) != 0ull) {
                // rare: some group's Gaussian has a rectangle that cuts its sigma_max ellipse —
                // apply the rectangle per pixel (and turn -0.0 into +0.0, see gs_pack_splats)
                asm volatile("; rectangle binds");
                if (sbits & 1u) {
                    const uint32_t rx = __float_as_uint(q1.w), ry = __float_as_uint(q2.w);
                    const bool in = (uint32_t)px >= (rx & 0xFFFFu) && (uint32_t)px < (rx >> 16) &&
                                    (uint32_t)py >= (ry & 0xFFFFu) && (uint32_t)py < (ry >> 16);
                    sg = in ? sg + 0.0f : qnan();
                }
            }
            // 0 <= sigma <= sigma_max as ONE unsigned compare of the bit patterns
            const bool need = __float_as_uint(sg) <= sbits;
            const uint64_t mneed = __builtin_amdgcn_ballot_w64(need);
            if (mneed == 0ull) continue;
            GS_STAT(1, 1);
            GS_STAT(2, __builtin_popcountll(mneed));
            float vis = 0.0f;
            if (need) vis = gs_exp<EXACT>(-sg, exp_tab);
            // gsplat_cpu.cpp:220-236: alpha = min(0.999, opacity*vis); skip if alpha < 1/255;
            // nextT = T*(1-alpha); nextT <= 1e-4 -> pixel done (Gaussian not rendered).  A skipped
            // pixel is given alpha = 0, which composites exactly nothing.
            float alpha = q1.y * vis;
            alpha = __builtin_amdgcn_fmed3f(alpha, 0.0f, 0.999f);
            bool ok = alpha >= (1.0f / 255.0f);
            alpha = ok ? alpha : 0.0f;
            float nT = T * (1.0f - alpha);
            if (__builtin_amdgcn_ballot_w64(nT <= 1e-4f) != 0ull) {
                asm volatile("; pixel saturates");
                if (nT <= 1e-4f) { pyf = qnan(); alpha = 0.0f; nT = T; ok = false; }
            }
            const float w = alpha * T;
            a0 = a0 + w * q2.x;
            a1 = a1 + w * q2.y;
            a2 = a2 + w * q2.z;
            T = nT;
            last = ok ? (c0 + e) : last;
        }
    }
    if (inimg) {
        const size_t pix = (size_t)py * W + px;
        const float o0 = a0 + T * bg0, o1 = a1 + T * bg1, o2 = a2 + T * bg2;
        out_img[3 * pix + 0] = o0;
        out_img[3 * pix + 1] = o1;
        out_img[3 * pix + 2] = o2;
        if (out_clamped) {  // fused torch::clamp_max(rgb, 1), model.cpp:222
            out_clamped[3 * pix + 0] = fminf(o0, 1.0f);
            out_clamped[3 * pix + 1] = fminf(o1, 1.0f);
            out_clamped[3 * pix + 2] = fminf(o2, 1.0f);
        }
        final_Ts[pix] = T;
        final_idx[pix] = last;
    }
}

// Sums of nine per-lane values over each 16-lane DPP row, in registers: a transposing butterfly.
// Stage 1 (lane ^ 1) folds value PAIRS — even lanes keep the pair sum of the even value, odd lanes of
// the odd value — stage 2 (lane ^ 2) does the same with pairs of those, leaving the quad sum of
// value (lane & 3) [+4]; two rotations by 4 and 8 lanes add the four quads.  13 + 7 + 6 + 2 = 28
// VALU (13 plain DPP adds), no LDS: the round-1 LDS reduction (9 stores + 4 x 16-byte loads per
// lane) made the LDS the bottleneck of this kernel once every step carries a reduction
// (SQ_LDS_IDX_ACTIVE == kernel duration, profiles/r02b).  Lane c (< 9) of each row returns total c.
__device__ __forceinline__ float row_reduce9(float v0, float v1, float v2, float v3, float v4,
                                             float v5, float v6, float v7, float v8, bool odd,
                                             bool bit1, int li) {
    const float w01 = (odd ? v1 : v0) + dpp_f<0xB1>(odd ? v0 : v1);  // quad_perm [1,0,3,2]
    const float w23 = (odd ? v3 : v2) + dpp_f<0xB1>(odd ? v2 : v3);
    const float w45 = (odd ? v5 : v4) + dpp_f<0xB1>(odd ? v4 : v5);
    const float w67 = (odd ? v7 : v6) + dpp_f<0xB1>(odd ? v6 : v7);
    const float w8 = v8 + dpp_f<0xB1>(v8);
    float x03 = (bit1 ? w23 : w01) + dpp_f<0x4E>(bit1 ? w01 : w23);  // quad_perm [2,3,0,1]
    float x47 = (bit1 ? w67 : w45) + dpp_f<0x4E>(bit1 ? w45 : w67);
    float x8 = w8 + dpp_f<0x4E>(w8);
    x03 += dpp_f<0x124>(x03);  // row_ror:4
    x47 += dpp_f<0x124>(x47);
    x8 += dpp_f<0x124>(x8);
    x03 += dpp_f<0x128>(x03);  // row_ror:8
    x47 += dpp_f<0x128>(x47);
    x8 += dpp_f<0x128>(x8);
    return li < 4 ? x03 : (li < 8 ? x47 : x8);
}

// ---------------------------------------------------------------------------------------------
// Backward.  LDS per wave: staged records 3.1 KB, ids 256 B, per-entry accumulators 9 x 65 floats:
// 5.7 KB.
constexpr int kAcc = 9;           // accumulator floats per staged entry
constexpr int kAccStride = kChunk + 1;  // component-major [9][65]: the nine components of an entry in nine banks
constexpr float kFixScale = 1099511627776.0f;  // 2^40: fixed-point scale of GS_FLAG_DETERMINISTIC
constexpr int kBackwardPixelsPerLane = 2;      // default WaveGeom of the backward (measured)

// Wave geometry of the backward kernel, by pixels per lane PX (1, 2 or 4).  A 16-lane group owns a
// block of BW x BH pixels — lane (li % LW, li / LW) holds the PX pixels of its column at rows
// r, r + LH, .. (one column per lane: xCam is shared by a lane's pixels) — and a wave owns 2 x 2
// blocks:  PX = 1: 4x4 blocks, 8x8 wave, four waves per tile;  PX = 2: 4x8 blocks, 8x16 wave, two per
// tile;  PX = 4: 8x8 blocks, 16x16 wave, one per tile.  More pixels per lane amortise the per-step
// costs (walk, record read, the nine-value reduction, the LDS atomic) over more pixels; fewer
// pixels per lane skip more of the pixels a Gaussian cannot reach.
template <int PX>
struct WaveGeom {
    static constexpr int BW = PX == 4 ? 8 : 4, BH = PX == 1 ? 4 : 8;
    static constexpr int LW = BW, LH = 16 / LW;          // lanes of a group: LW columns x LH rows
    static constexpr int WW = 2 * BW, WH = 2 * BH;       // pixels of a wave
    static constexpr int PER_TILE = (GS_TILE / WW) * (GS_TILE / WH);
};

template <int PX>
__device__ __forceinline__ bool decode_wave(int block, int num_tiles, int tiles_x, int W, int H,
                                            const int32_t *__restrict__ order, int &tile, int &wx0,
                                            int &wy0) {
    using G = WaveGeom<PX>;
    const int x = block & 7, k = block >> 3;
    const int part = k % G::PER_TILE;
    const int slot = ((k / G::PER_TILE) << 3) + x;
    if (slot >= num_tiles) return false;
    tile = order ? order[slot] : xcd_swizzle(slot, num_tiles);
    constexpr int PX_COLS = GS_TILE / G::WW;
    wx0 = (tile % tiles_x) * GS_TILE + G::WW * (part % PX_COLS);
    wy0 = (tile / tiles_x) * GS_TILE + G::WH * (part / PX_COLS);
    return wx0 < W && wy0 < H;
}

// which of the wave's 2 x 2 blocks the rectangle touches (bit g: block column g & 1, row g >> 1)
template <int PX>
__device__ __forceinline__ uint32_t block_touch_g(uint32_t rx, uint32_t ry, int wx0, int wy0) {
    using G = WaveGeom<PX>;
    const int x0 = (int)(rx & 0xFFFF) - wx0, x1 = (int)(rx >> 16) - wx0;
    const int y0 = (int)(ry & 0xFFFF) - wy0, y1 = (int)(ry >> 16) - wy0;
    if (x1 <= x0 || y1 <= y0) return 0u;
    const bool c0 = x0 < G::BW && x1 > 0, c1 = x0 < 2 * G::BW && x1 > G::BW;
    const bool r0 = y0 < G::BH && y1 > 0, r1 = y0 < 2 * G::BH && y1 > G::BH;
    return (c0 && r0 ? 1u : 0u) | (c1 && r0 ? 2u : 0u) | (c0 && r1 ? 4u : 0u) |
           (c1 && r1 ? 8u : 0u);
}

template <bool EXACT, bool DET, int PX>
__global__ void __launch_bounds__(64)
k_rasterize_backward(int W, int H, int tiles_x, int num_tiles, const int32_t *__restrict__ order,
                     const int32_t *__restrict__ ids, const int2 *__restrict__ bins,
                     const float4 *__restrict__ packed, float bg0, float bg1, float bg2,
                     const float *__restrict__ bg_dev, const float *__restrict__ final_Ts,
                     const int32_t *__restrict__ final_idx, const float *__restrict__ v_out,
                     const float *__restrict__ v_out_alpha, const float *__restrict__ img_raw,
                     float *__restrict__ gacc, unsigned long long *__restrict__ gfix) {
    using G = WaveGeom<PX>;
    __shared__ SRec stage[kChunk + 1];
    __shared__ int sid[kChunk];
    __shared__ float acc[kAcc * kAccStride];
    const int lane = threadIdx.x;
    int tile, wx0, wy0;
    if (!decode_wave<PX>(blockIdx.x, num_tiles, tiles_x, W, H, order, tile, wx0, wy0)) return;
    if (bg_dev) {
        bg0 = bg_dev[0]; bg1 = bg_dev[1]; bg2 = bg_dev[2];
    }
    const int grp = lane >> 4, li = lane & 15;
    const bool odd = (lane & 1) != 0, bit1 = (lane & 2) != 0;
    const uint32_t gsh = 8u * (uint32_t)grp;
    const int px = wx0 + G::BW * (grp & 1) + (li % G::LW);
    const int py0 = wy0 + G::BH * (grp >> 1) + (li / G::LW);   // pixel p of the lane: row py0 + p * LH
    const float pxf = (float)px;
    // per pixel: row, transmittance being unwound, T_final * (v_out_alpha - bg . v_out), running
    // <colour buffer, v_out>, cotangent, list index of the last contributor
    float pyf[PX], T[PX], TW[PX], bv[PX], vo0[PX], vo1[PX], vo2[PX];
    int last[PX];
    int gl = -1;
#pragma unroll
    for (int p = 0; p < PX; p++) {
        const int py = py0 + p * G::LH;
        float Tfin = 1.0f, oa = 0.0f;
        vo0[p] = vo1[p] = vo2[p] = 0.0f;
        last[p] = -1;
        if (px < W && py < H) {
            const size_t pix = (size_t)py * W + px;
            Tfin = final_Ts[pix];
            last[p] = final_idx[pix];
            vo0[p] = v_out[3 * pix + 0];
            vo1[p] = v_out[3 * pix + 1];
            vo2[p] = v_out[3 * pix + 2];
            if (img_raw) {  // backward of the fused clamp_max(rgb, 1): torch passes where rgb <= 1
                if (!(img_raw[3 * pix + 0] <= 1.0f)) vo0[p] = 0.0f;
                if (!(img_raw[3 * pix + 1] <= 1.0f)) vo1[p] = 0.0f;
                if (!(img_raw[3 * pix + 2] <= 1.0f)) vo2[p] = 0.0f;
            }
            oa = v_out_alpha ? v_out_alpha[pix] : 0.0f;
        }
        pyf[p] = (float)py;
        T[p] = Tfin;
        TW[p] = Tfin * (oa - (bg0 * vo0[p] + bg1 * vo1[p] + bg2 * vo2[p]));
        bv[p] = 0.0f;
        gl = max(gl, last[p]);
    }
    // last contributor of each block (one DPP row) and of the wave
    gl = max(gl, dpp_i<0xB1>(gl));
    gl = max(gl, dpp_i<0x4E>(gl));
    gl = max(gl, dpp_i<0x141>(gl));
    gl = max(gl, dpp_i<0x140>(gl));
    const int gl0 = __builtin_amdgcn_readlane(gl, 0), gl1 = __builtin_amdgcn_readlane(gl, 16);
    const int gl2 = __builtin_amdgcn_readlane(gl, 32), gl3 = __builtin_amdgcn_readlane(gl, 48);
    const int wave_last = max(max(gl0, gl1), max(gl2, gl3));
    const int2 range = bins[tile];
    if (wave_last < range.x) return;  // (also covers empty tiles / no contributors)

    if (lane == 0) stage_sentinel(&stage[kChunk]);
#pragma unroll
    for (int i = 0; i < kAcc; i++) acc[i * kAccStride + lane] = 0.0f;

    // walk the list back to front in chunks; slot 0 of a chunk is its furthest-back entry
    float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0;
    int ng = 0;
    if (wave_last - lane >= range.x) {
        ng = ids[wave_last - lane];
        n0 = packed[3 * (size_t)ng + 0]; n1 = packed[3 * (size_t)ng + 1]; n2 = packed[3 * (size_t)ng + 2];
    }
    for (int hi = wave_last; hi >= range.x; hi -= kChunk) {
        __syncthreads();
        uint32_t touch = 0u;
        if (hi - lane >= range.x) {
            stage[lane].p0 = n0;
            stage[lane].p1 = n1;
            stage[lane].p2 = n2;
            sid[lane] = ng;
            touch = block_touch_g<PX>(__float_as_uint(n1.w), __float_as_uint(n2.w), wx0, wy0);
        }
        uint64_t m0 = __builtin_amdgcn_ballot_w64((touch & 1u) != 0u);
        uint64_t m1 = __builtin_amdgcn_ballot_w64((touch & 2u) != 0u);
        uint64_t m2 = __builtin_amdgcn_ballot_w64((touch & 4u) != 0u);
        uint64_t m3 = __builtin_amdgcn_ballot_w64((touch & 8u) != 0u);
        // entries behind the last contributor of every pixel of a block: slot t has list index
        // hi - t, needed only if hi - t <= gl_g
#define GS_TRIM(m, glg)                                                          \
    {                                                                            \
        const int d = hi - (glg);                                                \
        if (d > 0) (m) = d >= kChunk ? 0ull : ((m) & ~((1ull << d) - 1ull));     \
    }
        GS_TRIM(m0, gl0) GS_TRIM(m1, gl1) GS_TRIM(m2, gl2) GS_TRIM(m3, gl3)
#undef GS_TRIM
        __syncthreads();
        GS_STAT(11, __builtin_popcountll(m0) + __builtin_popcountll(m1) + __builtin_popcountll(m2) + __builtin_popcountll(m3));
        GS_STAT(12, 1);
        if (hi - kChunk - lane >= range.x) {
            ng = ids[hi - kChunk - lane];
            n0 = packed[3 * (size_t)ng + 0]; n1 = packed[3 * (size_t)ng + 1]; n2 = packed[3 * (size_t)ng + 2];
        }
        bool flushed_any = false;  // wave-uniform
        while ((m0 | m1 | m2 | m3) != 0ull) {
            GS_WALK_STEP(m0, e0)
            GS_WALK_STEP(m1, e1)
            GS_WALK_STEP(m2, e2)
            GS_WALK_STEP(m3, e3)
            // the four slots packed into one SGPR; each lane extracts its group's (one v_bfe_u32)
            const uint32_t ep = (uint32_t)e0 | ((uint32_t)e1 << 8) | ((uint32_t)e2 << 16) | ((uint32_t)e3 << 24);
            const int e = (int)((ep >> gsh) & 0xFFu);
            const float4 q0 = stage[e].p0, q1 = stage[e].p1, q2 = stage[e].p2;
            const uint32_t sbits = __float_as_uint(q1.z);
            const int idx = hi - e;  // index of this entry in the sorted list
            GS_STAT(8, 1);
            const float dx = q0.x - pxf;
            const float Adxdx = (q0.z * dx) * dx, Bdx = q0.w * dx;
            // rectangle test data of the rare entries whose rectangle cuts the sigma_max ellipse
            const bool any_binds = __builtin_amdgcn_ballot_w64((sbits & 1u) != 0u) != 0ull;
            float su = 0.0f, suy = 0.0f, suyy = 0.0f, gr = 0.0f, gg = 0.0f, gb = 0.0f;
            bool any = false;  // wave-uniform
#pragma unroll
            for (int p = 0; p < PX; p++) {
                const float dy = q0.y - pyf[p];
                float sg = 0.5f * fmaf(q1.x * dy, dy, Adxdx);
                sg = fmaf(Bdx, dy, sg);
                if (any_binds) {
                    asm volatile("; rectangle binds");
                    if (sbits & 1u) {
                        // decide exactly like the forward: its op order for sigma, rectangle applied
                        float se = Adxdx + (q1.x * dy) * dy;
                        se = 0.5f * se;
                        se = se + Bdx * dy;
                        const uint32_t rx = __float_as_uint(q1.w), ry = __float_as_uint(q2.w);
                        const uint32_t pyu = (uint32_t)(py0 + p * G::LH);
                        const bool in = (uint32_t)px >= (rx & 0xFFFFu) && (uint32_t)px < (rx >> 16) &&
                                        pyu >= (ry & 0xFFFFu) && pyu < (ry >> 16);
                        sg = in ? se + 0.0f : qnan();
                    }
                }
                const bool need = (idx <= last[p]) && (__float_as_uint(sg) <= sbits);
                const uint64_t mneed = __builtin_amdgcn_ballot_w64(need);
                if (mneed == 0ull) continue;
                any = true;
                GS_STAT(9, 1);
                GS_STAT(10, __builtin_popcountll(mneed));
                // vis = exp(-sigma), alpha = min(0.99, opacity * vis), gsplat_cpu.cpp:337-338; lanes
                // that do not take part end up with vis = alpha = 0
                float vis = need ? __expf(-sg) : 0.0f;
                float alpha = q1.y * vis;
                if (EXACT) {
                    // same >= 1/255 decision as the forward: redo the exponential exactly (from the
                    // forward's sigma) where the fast one cannot decide
                    const float thr = 1.0f / 255.0f;
                    const bool amb = need && fabsf(alpha - thr) < 1.0e-8f;
                    if (__builtin_amdgcn_ballot_w64(amb) != 0ull) {
                        asm volatile("; threshold ambiguous");
                        if (amb) {
                            float se = Adxdx + (q1.x * dy) * dy;
                            se = 0.5f * se;
                            se = se + Bdx * dy;
                            vis = expf_glibc_cmem(-se);
                            alpha = q1.y * vis;
                        }
                    }
                }
                const bool ok = alpha >= (1.0f / 255.0f);
                alpha = ok ? __builtin_amdgcn_fmed3f(alpha, 0.0f, 0.99f) : 0.0f;
                vis = ok ? vis : 0.0f;
                // ra = 1 / (1 - alpha): hardware reciprocal + one Newton step
                const float om = 1.0f - alpha;
                float ra = __builtin_amdgcn_rcpf(om);
                ra = fmaf(ra, fmaf(-om, ra, 1.0f), ra);
                T[p] = T[p] * ra;  // transmittance in front of this Gaussian
                const float fac = alpha * T[p];
                gr = fmaf(fac, vo0[p], gr);
                gg = fmaf(fac, vo1[p], gg);
                gb = fmaf(fac, vo2[p], gb);
                // cv = <colour, v_out>;  v_alpha = T*cv + ra*(T_final*w - <buffer, v_out>)
                const float cv = fmaf(q2.z, vo2[p], fmaf(q2.y, vo1[p], q2.x * vo0[p]));
                const float v_alpha = fmaf(T[p], cv, ra * (TW[p] - bv[p]));
                bv[p] = fmaf(fac, cv, bv[p]);
                // u = vis * v_alpha (= d/d opacity); v_sigma = -opacity * u is applied at the flush
                const float u = vis * v_alpha;
                const float uy = u * dy;
                su += u;
                suy += uy;
                suyy = fmaf(uy, dy, suyy);
            }
            if (!any) continue;
            // ---- the nine sums over the group's 16 lanes: lane c of the row ends up with total c ----
            const float ux = su * dx;
            const float r = row_reduce9(ux, suy, ux * dx, suy * dx, suyy, gr, gg, gb, su, odd, bit1, li);
            if (li < kAcc && r != 0.0f && e < kChunk)  // (a group without work has nothing to add)
                __hip_atomic_fetch_add(&acc[li * kAccStride + e], r, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            flushed_any = true;
        }
        if (!flushed_any) continue;
        // ---- flush: moments -> gradient components (once per entry), then one atomic lane per
        //      (entry, component): the nine lanes of an entry hit ONE 64-byte record ----
        wave_sync();
        if (hi - lane >= range.x) {  // (slots beyond the list's head hold no entry)
            const float Ux = acc[0 * kAccStride + lane], Uy = acc[1 * kAccStride + lane];
            const float Uxx = acc[2 * kAccStride + lane], Uxy = acc[3 * kAccStride + lane];
            const float Uyy = acc[4 * kAccStride + lane];
            const float A = stage[lane].p0.z, B = stage[lane].p0.w, C = stage[lane].p1.x;
            const float mo = -stage[lane].p1.y;          // v_sigma = -opacity * u
            acc[0 * kAccStride + lane] = mo * fmaf(A, Ux, B * Uy);   // v_x: v_sigma * (A dx + B dy)
            acc[1 * kAccStride + lane] = mo * fmaf(B, Ux, C * Uy);   // v_y: v_sigma * (B dx + C dy)
            acc[2 * kAccStride + lane] = 0.5f * mo * Uxx;            // v_A  (gsplat_cpu.cpp:361-363)
            acc[3 * kAccStride + lane] = 0.5f * mo * Uxy;            // v_B
            acc[4 * kAccStride + lane] = 0.5f * mo * Uyy;            // v_C
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < kAcc; i++) {
            const int k = i * kChunk + lane;          // record-major enumeration: k = 9 * entry + comp
            const int ent = (k * 7282) >> 16;         // k / 9 for k < 576
            const int comp = k - 9 * ent;
            const float v = acc[comp * kAccStride + ent];
            if (v != 0.0f) {
                const size_t o = (size_t)sid[ent] * kGradRec + comp;
                if (DET)
                    atomicAdd(gfix + o, (unsigned long long)(long long)(v * kFixScale));
                else
                    atomicAdd(gacc + o, v);
            }
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < kAcc; i++) acc[i * kAccStride + lane] = 0.0f;
    }
}
#undef GS_WALK_STEP

// Test hook: row_reduce9 on given values.  in [blocks, 9, 64] -> out [blocks, 4, 9] (row, value).
__global__ void __launch_bounds__(64) k_debug_row_reduce9(const float *__restrict__ in,
                                                          float *__restrict__ out) {
    const int lane = threadIdx.x;
    const float *p = in + (size_t)blockIdx.x * 9 * 64;
    float v[9];
#pragma unroll
    for (int i = 0; i < 9; i++) v[i] = p[i * 64 + lane];
    const int li = lane & 15;
    const float r = row_reduce9(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], (lane & 1) != 0,
                                (lane & 2) != 0, li);
    if (li < 9) out[((size_t)blockIdx.x * 4 + (lane >> 4)) * 9 + li] = r;
}

// GS_FLAG_DETERMINISTIC: 64-bit fixed-point sums -> the float records
__global__ void __launch_bounds__(256)
k_fixed_to_records(int64_t n, const long long *__restrict__ fix, float *__restrict__ rec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rec[i] = (float)((double)fix[i] * (1.0 / 1099511627776.0));
}

}  // namespace gs

extern "C" int gs_debug_row_reduce9(int blocks, const float *in, float *out, gs_stream_t stream) {
    if (blocks < 0) return GS_ERR_INVALID_ARGUMENT;
    if (blocks == 0) return GS_OK;
    if (!in || !out) return GS_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(gs::k_debug_row_reduce9, dim3(blocks), dim3(64), 0, (hipStream_t)stream, in, out);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" size_t gs_rasterize_backward_workspace_bytes_det(int N) {
    // float records + the 64-bit fixed-point accumulators of GS_FLAG_DETERMINISTIC
    return N > 0 ? (size_t)N * gs::kGradRec * (sizeof(float) + sizeof(long long)) : 0;
}

extern "C" int gs_rasterize_forward(int W, int H, const int32_t *gaussian_ids_sorted,
                                    const int32_t *tile_bins, const float *packed,
                                    const float *background, float *out_img, float *final_Ts,
                                    int32_t *final_idx, float *out_img_clamped,
                                    const int32_t *list_stats, const int32_t *tile_order,
                                    uint32_t flags, gs_stream_t stream) {
    if (flags & (1u << 20))  // A/B: the round-1 one-wave-per-tile kernels
        return gs_rasterize_forward_v1(W, H, gaussian_ids_sorted, tile_bins, packed, background, out_img,
                                       final_Ts, final_idx, out_img_clamped, list_stats, tile_order,
                                       flags, stream);
    if (W <= 0 || H <= 0) return GS_ERR_INVALID_ARGUMENT;
    if ((flags & GS_FLAG_CLAMP_IMAGE) && !out_img_clamped) return GS_ERR_INVALID_ARGUMENT;
    float *clamped = (flags & GS_FLAG_CLAMP_IMAGE) ? out_img_clamped : nullptr;
    if (W > 65535 || H > 65535) return GS_ERR_UNSUPPORTED;
    if (!tile_bins || !background || !out_img || !final_Ts || !final_idx)
        return GS_ERR_INVALID_ARGUMENT;
    if ((uintptr_t)packed & 15u) return GS_ERR_INVALID_ARGUMENT;
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    const int tiles = tiles_x * tiles_y;
    hipStream_t s = (hipStream_t)stream;
    const int2 *bins = reinterpret_cast<const int2 *>(tile_bins);
    const float4 *pk = reinterpret_cast<const float4 *>(packed);
    const int units = 4 * 8 * ((tiles + 7) / 8);  // four quadrant waves per tile, see decode_quadrant
    const float *bg_dev = gs::on_device(background) ? background : nullptr;
    const float bg0 = bg_dev ? 0.f : background[0], bg1 = bg_dev ? 0.f : background[1],
                bg2 = bg_dev ? 0.f : background[2];
    gs::ev_before(s);
    if (flags & GS_FLAG_FAST_EXP)
        hipLaunchKernelGGL((gs::k_rasterize_forward<false>), dim3(units), dim3(64), 0, s, W, H, tiles_x,
                           tiles, tile_order, gaussian_ids_sorted, bins, pk, bg0, bg1, bg2, bg_dev,
                           out_img, final_Ts, final_idx, clamped);
    else
        hipLaunchKernelGGL((gs::k_rasterize_forward<true>), dim3(units), dim3(64), 0, s, W, H, tiles_x,
                           tiles, tile_order, gaussian_ids_sorted, bins, pk, bg0, bg1, bg2, bg_dev,
                           out_img, final_Ts, final_idx, clamped);
    gs::ev_after(s);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" int gs_rasterize_backward(int W, int H, int N, const int32_t *gaussian_ids_sorted,
                                     const int32_t *tile_bins, const float *packed,
                                     const float *background, const float *final_Ts,
                                     const int32_t *final_idx, const float *v_out,
                                     const float *v_out_alpha, const float *out_img, float *v_xy,
                                     float *v_conic, float *v_colors, float *v_opacity,
                                     void *workspace, size_t workspace_bytes,
                                     const int32_t *list_stats, const int32_t *tile_order,
                                     uint32_t flags, gs_stream_t stream) {
    if (flags & (1u << 20))  // A/B: the round-1 one-wave-per-tile kernels
        return gs_rasterize_backward_v1(W, H, N, gaussian_ids_sorted, tile_bins, packed, background,
                                        final_Ts, final_idx, v_out, v_out_alpha, out_img, v_xy, v_conic,
                                        v_colors, v_opacity, workspace, workspace_bytes, list_stats,
                                        tile_order, flags, stream);
    if (W <= 0 || H <= 0 || N < 0) return GS_ERR_INVALID_ARGUMENT;
    if ((flags & GS_FLAG_CLAMP_IMAGE) && !out_img) return GS_ERR_INVALID_ARGUMENT;
    const float *img_raw = (flags & GS_FLAG_CLAMP_IMAGE) ? out_img : nullptr;
    if (W > 65535 || H > 65535) return GS_ERR_UNSUPPORTED;
    if (N == 0) return GS_OK;
    const bool keep_records = (flags & GS_FLAG_KEEP_RECORDS) != 0u;
    const bool det = (flags & GS_FLAG_DETERMINISTIC) != 0u;
    if (!tile_bins || !background || !final_Ts || !final_idx || !v_out || !workspace)
        return GS_ERR_INVALID_ARGUMENT;
    if (!keep_records && (!v_xy || !v_conic || !v_colors || !v_opacity)) return GS_ERR_INVALID_ARGUMENT;
    if (((uintptr_t)packed & 15u) || ((uintptr_t)workspace & 63u)) return GS_ERR_INVALID_ARGUMENT;
    const size_t rec_bytes = gs_rasterize_backward_workspace_bytes(N);
    if (workspace_bytes < (det ? gs_rasterize_backward_workspace_bytes_det(N) : rec_bytes))
        return GS_ERR_WORKSPACE;
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    const int tiles = tiles_x * tiles_y;
    hipStream_t s = (hipStream_t)stream;
    const int2 *bins = reinterpret_cast<const int2 *>(tile_bins);
    const float4 *pk = reinterpret_cast<const float4 *>(packed);
    float *gacc = static_cast<float *>(workspace);
    unsigned long long *gfix =
        det ? reinterpret_cast<unsigned long long *>(static_cast<char *>(workspace) + rec_bytes) : nullptr;
    if (det)
        GS_HIP_CHECK(hipMemsetAsync(gfix, 0, (size_t)N * gs::kGradRec * sizeof(long long), s));
    else if (!(flags & GS_FLAG_RECORDS_ZEROED))
        GS_HIP_CHECK(hipMemsetAsync(gacc, 0, rec_bytes, s));
    // pixels per lane of the backward (WaveGeom): flag bits 21..22 select 1 / 2 / 4 for experiments
    int px_per_lane = gs::kBackwardPixelsPerLane;
    if (((flags >> 21) & 3u) != 0u) px_per_lane = 1 << (((flags >> 21) & 3u) - 1u);
    const int units = (px_per_lane == 1 ? 4 : (px_per_lane == 2 ? 2 : 1)) * 8 * ((tiles + 7) / 8);
    const float *bg_dev = gs::on_device(background) ? background : nullptr;
    const float bg0 = bg_dev ? 0.f : background[0], bg1 = bg_dev ? 0.f : background[1],
                bg2 = bg_dev ? 0.f : background[2];
    gs::ev_before(s);
#define GS_BWD_LAUNCH3(EX, DT, PXN)                                                                       \
    hipLaunchKernelGGL((gs::k_rasterize_backward<EX, DT, PXN>), dim3(units), dim3(64), 0, s, W, H, tiles_x, \
                       tiles, tile_order, gaussian_ids_sorted, bins, pk, bg0, bg1, bg2, bg_dev,           \
                       final_Ts, final_idx, v_out, v_out_alpha, img_raw, gacc, gfix)
#define GS_BWD_LAUNCH(EX, DT)                                      \
    do {                                                           \
        if (px_per_lane == 1) GS_BWD_LAUNCH3(EX, DT, 1);           \
        else if (px_per_lane == 2) GS_BWD_LAUNCH3(EX, DT, 2);      \
        else GS_BWD_LAUNCH3(EX, DT, 4);                            \
    } while (0)
    if (det) {
        if (flags & GS_FLAG_FAST_EXP) GS_BWD_LAUNCH(false, true); else GS_BWD_LAUNCH(true, true);
    } else {
        if (flags & GS_FLAG_FAST_EXP) GS_BWD_LAUNCH(false, false); else GS_BWD_LAUNCH(true, false);
    }
#undef GS_BWD_LAUNCH3
#undef GS_BWD_LAUNCH
    gs::ev_after(s);
    GS_LAUNCH_CHECK();
    if (det) {
        const int64_t n = (int64_t)N * gs::kGradRec;
        hipLaunchKernelGGL(gs::k_fixed_to_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n,
                           reinterpret_cast<const long long *>(gfix), gacc);
        GS_LAUNCH_CHECK();
    }
    if (keep_records) return GS_OK;  // the 64-byte records go straight to gs_gaussian_backward
    hipLaunchKernelGGL(gs::k_unpack_grads, dim3((N + 255) / 256), dim3(256), 0, s, N,
                       reinterpret_cast<const float4 *>(gacc),
                       (flags & GS_FLAG_LOGIT_OPACITY) ? pk : nullptr, v_xy, v_conic, v_colors,
                       v_opacity);
    GS_LAUNCH_CHECK();
    return GS_OK;
}
