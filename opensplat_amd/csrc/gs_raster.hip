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
// Design: WAVES OWN A PIECE OF A TILE, AND EACH 16-LANE GROUP WALKS ITS OWN LIST.
//
// The round-1 kernels gave one wave a whole 16x16 tile and let all 64 lanes evaluate every entry of
// the tile's list; at BASELINE config 2 (footprints of ~30 pixels) only ~16 of 64 lanes were live
// in an exponential pass.  Here a wave owns 2 x 2 BLOCKS of pixels (WaveGeom: 4x4
// blocks with one pixel per lane in the forward; 8x8 blocks with FOUR pixels per lane in the backward,
// one wave per tile — 4x8 / 4x4 blocks with two / one pixel per lane on frames that do not fill the chip
// and for tiles with outlying lists), each block belongs to one 16-lane group (a DPP row), and each group
// walks ITS OWN list: the entries of the staged chunk whose coverage mask touches that block.  The four
// groups execute one instruction stream on four different Gaussians:
//   * per chunk of 64 list entries (staged by the wave itself: lane t gathers entry t, the next
//     chunk's gather is in flight while the current one is consumed), four ballots give four
//     64-bit "touches block g" masks; they live in SGPRs, the walk is scalar (s_ff1 + clear the
//     bit, per group), the four current slots travel to the lanes packed in ONE SGPR (one
//     v_bfe_u32 per lane), and the chunk takes max_g popcount(mask_g) steps;
//   * a group without work reads a sentinel record whose x is NaN: sigma is then NaN and fails the
//     ONE unsigned compare "0 <= sigma <= sigma_max" on the bit patterns (negatives and NaNs have
//     larger patterns than any sigma_max = ln(255 * opacity) < 6), exactly as a finished pixel's
//     NaN row coordinate does in the forward.  The CPU oracle's pixel-rectangle test is implied by
//     that compare whenever the rectangle encloses the sigma_max ellipse box (flag bit set by
//     gs_pack_splats); only for the rare Gaussians whose rectangle cuts the ellipse is the rectangle
//     applied per pixel (wave-uniform branch on a ballot of the flag);
//   * the per-entry record is read from LDS with a per-group address (the 48-byte packed record,
//     three ds_read_b128; identical addresses inside a group broadcast);
//   * every wave stages the tile's list for itself: no workgroup barriers, the waves of a tile
//     take consecutive positions in ONE XCD's block stream (shared L2 lines), a long list is
//     automatically shared by several waves, tiles are launched longest list first;
//   * a pixel that is skipped gets alpha = 0, which composites exactly nothing — no per-pixel
//     branches; saturation (once per pixel and frame) is a wave-uniform rare path;
//   * backward: the staged record carries the conic times log2(e), so sigma' = sigma log2(e) comes
//     straight out of two fused multiply-adds and the exponential is ONE v_exp_f32; the forward's
//     alpha >= 1/255 decision is taken on sigma' BEFORE the exponential against the entry's own
//     thresholds log2(255 o) -+ 7.5e-6 (staged once per chunk, SRecB), and only lanes between the two
//     redo the forward's exact arithmetic (its sigma, the glibc-exact fp64 exponential), so the decision
//     equals the forward's; 1/(1-alpha) is v_rcp_f32 alone (T is a running product either way); the
//     running colour buffer is tracked as its dot product with the pixel's cotangent; the moments
//     sum(u), sum(u dy), sum(u dy^2) are accumulated over a lane's pixels (they share xCam);
//   * the nine partial sums are reduced over the group's 16 lanes in registers by a transposing DPP
//     butterfly (row_reduce9, 21 VALU; the same sums on the matrix pipe — mfma_reduce9, GS_BWD_MFMA=1 —
//     measured 0.40 against 0.29 ms at C2 and stay a build option), added to per-entry accumulators in
//     LDS (ds_add_f32), converted from moments to (v_x, v_y, v_A, v_B, v_C) once per entry and flushed once per
//     chunk with one atomic lane per (entry, component): a Gaussian costs one global atomic
//     line-request per wave it contributes to (its nine lanes hit ONE 64-byte record).
//
//
// Round 5 (profiles/HISTORY.md): (a) the lists of the groups are no longer walked as SGPR masks
// but read from per-group slot QUEUES in LDS, built per chunk from the same ballots (GS_FWD_QWALK, GS_BWD_QWALK);
// (b) the backward of a full frame with small footprints runs SIXTEEN four-lane groups per wave, one per 4x4
// block (backward_wave_q: orbit reduction, claimed plain read-add-write instead of LDS atomics), the four-group
// backward_wave below stays for frames of few tiles, the pieces, outlying lists and big splats.
//
// Round 6: (a) the backward waves set their own issue priority (wave_prio: the launch's final residents by the work
// they have left, everyone in front of them at the top) — the tail of a launch of ~2 rounds of equal tiles, taken from
// the arbiter's side: C2 backward -5.5 %; (b) both walks read from `hipcc -S`: the q walk 135 -> 125 VALU per step
// (C3 -3.2 %), the forward walk four steps per iteration on two swapping pairs of queue words, 44 -> ~41.5 (C2 / C3
// -2.6 %); (c) packed fp32 passes in the q walk; same bits throughout.
//
// Roofline: HBM traffic is one 48-byte gather per entry (per wave of the tile, served by L2) plus
// 20 B per pixel; DESIGN.md states the algorithmic bytes used for roofline.achieved and the VALU
// accounting.  Measured history of both kernels: DESIGN.md 4.1, profiles/.
#include <type_traits>

#include "gs_device.h"

namespace gs {

constexpr int kChunk = 64;  // entries staged per pass == wave width
// 1: the forward's four groups read their next slot from per-group queues in LDS (built per chunk from the four
// ballots, like backward_wave_q's sixteen) instead of walking four 64-bit SGPR masks with s_ff1: twenty scalar
// instructions per step less on a kernel whose one scalar unit per CU is a co-bottleneck — 169.5 -> 163.5 us at
// C2, 1.146 -> 1.125 ms at C3, same bits (round 5; 0 = the scalar walk, for measurements)
#ifndef GS_FWD_COMPACT
#define GS_FWD_COMPACT 0   // 1: full frames take k_rasterize_forward_c (round 6; measured, not the default)
#endif
// 1: full-frame chunks without a binding rectangle walk without looking for one — two VALU and a branch less per
// step; measured twice, lost twice (round 4: 185.1 against 185.8 us; round 6, same box, three interleaved lines each:
// 0.1604 against 0.1578 ms at C2 — the per-chunk ballot and a second copy of the walk cost more than they save)
#ifndef GS_FWD_CHUNK_BINDS
#define GS_FWD_CHUNK_BINDS 0
#endif
#ifndef GS_FWD_QWALK
#define GS_FWD_QWALK 1
#endif
// floats per Gaussian in the backward's gradient records: nine are payload {vx vy vA vB | vC vr vg vb | vo}.
// 16 (64 B: a record never straddles a cache line) or 12 (48 B: a quarter less memset, atomic write-back and read
// traffic, two records in eight straddle a 128-byte line) — measured in round 6, profiles/HISTORY.md.
#ifndef GS_GRAD_REC
#define GS_GRAD_REC 16
#endif
constexpr int kGradRec = GS_GRAD_REC;
static_assert(kGradRec == 12 || kGradRec == 16, "records are whole float4s holding nine floats");

// Optional work counters (build with -DGS_STATS; never in the shipped library): per launch totals of
// [0] steps  [1] steps with a needing lane  [2] needing lanes  [3] (block, entry) pairs walked
// [4] wave-chunks  — forward in slots 0-7, backward in slots 8-15 (scripts/work_stats.py).
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
// Optional per-workgroup log of the two full-frame compositing kernels (build with -DGS_WAVELOG; never in the
// shipped library; scripts/wave_timeline.py): record blockIdx.x = {start, end (100 MHz s_memrealtime),
// HW_ID | XCC_ID << 32, tile | list length << 32}.
#ifdef GS_WAVELOG
constexpr int kWaveLog = 65536;
__device__ unsigned long long g_wavelog[4 * kWaveLog];
#define GS_WLOG_T0 const unsigned long long wlog_t0 = wall_clock64();
#define GS_WLOG_T1(tile, len)                                                                              \
    do {                                                                                                   \
        if (threadIdx.x == 0 && blockIdx.x < kWaveLog) {                                                   \
            unsigned int hw, xcc;                                                                          \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                               \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                             \
            unsigned long long *r = &g_wavelog[4 * blockIdx.x];                                            \
            r[0] = wlog_t0;                                                                                \
            r[1] = wall_clock64();                                                                         \
            r[2] = (unsigned long long)hw | ((unsigned long long)xcc << 32);                               \
            r[3] = (unsigned long long)(unsigned int)(tile) | ((unsigned long long)(unsigned int)(len) << 32); \
        }                                                                                                  \
    } while (0)
#else
#define GS_WLOG_T0
#define GS_WLOG_T1(tile, len) \
    do {                      \
    } while (0)
#endif
__device__ __forceinline__ float qnan() { return __uint_as_float(0x7fc00000u); }

struct __attribute__((aligned(16))) SRec {
    float4 p0, p1, p2;  // the packed record as gathered: {x y A B | C o smax rx | r g b ry}
};
// The backward's staged record: the rectangle words (read by the rare binding entries only) move to a
// fourth vector and their places carry the entry's VISIBILITY thresholds: alpha = o * vis >= 1/255 is
// decided as vis >= (1/255) / o, formed once per staged entry instead of once per pixel pass.
//   {x y A B | C o smax t_hi | r g b t_lo | rx ry - -},  t_hi/lo = (1/255) / o * (1 +- kVisBand)
// vis >= t_hi: certainly a contributor; vis < t_lo: certainly none; in between the pass redoes the
// forward's exact arithmetic.  The band covers the fast exponential (2^-21), the backward's fused sigma
// (<= 1.5e-6 relative in vis at sigma = 5.5) and the roundings of the threshold itself.
struct __attribute__((aligned(16))) SRecB {
    float4 p0, p1, p2;
    float4 p3;   // {rx, ry, A, B}: read by the rare exact paths and by the flush
    float4 p4;   // {C, -, -, -}
};
// GS_BWD_LOG2E (round 4): the staged record carries the conic and sigma_max multiplied by log2(e), so that the
// per-pixel pass gets sigma' = sigma log2(e) straight from its two fused multiply-adds and the exponential is
// ONE instruction (v_exp_f32 is 2^x: exp(-sigma) = 2^(-sigma')) instead of a multiply and v_exp_f32; the
// original A, B, C sit in p3 / p4 for the exact re-evaluations and the flush.  The scaled entries are
// rounded once (6e-8 relative): another <= 1e-6 relative in vis at sigma = 5.5, inside kVisBand's budget
// (fast exponential 2^-21 = 4.8e-7, fused sigma 1.5e-6, the scaling 1e-6, the thresholds' own roundings).
#ifndef GS_BWD_LOG2E
#define GS_BWD_LOG2E 1
#endif
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kVisBand = GS_BWD_LOG2E ? 5.0e-6f : 4.0e-6f;
// GS_BWD_SIGMA_THRESH (round 4, needs GS_BWD_LOG2E): the forward's alpha = o exp(-sigma) >= 1/255 decision is
// taken on sigma' BEFORE the exponential, against the entry's own thresholds L' -+ kSigBand with
// L' = log2(255 o) — one compare tells "possibly a contributor" (it replaces the sigma <= sigma_max filter:
// the staged record's sigma_max IS L' + kSigBand), a second one "certainly"; between the two the pass redoes
// the forward's exact arithmetic.  One compare and one scalar AND less per pass than deciding on vis.
// Budget of the band, in sigma' units (sigma' <= 8): fused / scaled sigma' 3.6e-6, v_log_f32 of the
// threshold 1e-6, the forward's own roundings (expf, the product, 1/255) 2e-7, the flag bit 1e-6.
#ifndef GS_BWD_SIGMA_THRESH
#define GS_BWD_SIGMA_THRESH GS_BWD_LOG2E
#endif
constexpr float kSigBand = 7.5e-6f;

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

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

// which of the wave's 2 x 2 blocks an entry touches (bit g: block column g & 1, row g >> 1), from the
// entry's 16-bit coverage mask of the tile's sixteen 4x4-pixel blocks (block_mask16, gs_device.h:
// bit 4 r + c; computed ONCE per (tile, Gaussian) by the binning, exact to the sigma_max ellipse —
// the bounding rectangle used before touched 17 % more 4x4 blocks and 9 % more 8x8 blocks at C2).
// (ox, oy): pixel offset of the wave inside its tile.  A group's block spans BW/4 x BH/4 mask bits.
template <int PX>
__device__ __forceinline__ uint32_t touch_from_mask(uint32_t m, int ox, int oy) {
    using G = WaveGeom<PX>;
    constexpr uint32_t row = G::BW == 8 ? 3u : 1u;                   // one mask row of a group's block
    constexpr uint32_t blk = G::BH == 8 ? (row | (row << 4)) : row;  // the whole block
    constexpr int dc = G::BW / 4, dr = G::BH / 4;
    const uint32_t t = m >> (4 * (oy >> 2) + (ox >> 2));             // the wave's first block at bit 0
    return ((t & blk) ? 1u : 0u) | ((t & (blk << dc)) ? 2u : 0u) | ((t & (blk << (4 * dr))) ? 4u : 0u) |
           ((t & (blk << (4 * dr + dc))) ? 8u : 0u);
}

template <class R>
__device__ __forceinline__ void stage_sentinel(R *s) {
    s->p0 = make_float4(qnan(), 0.0f, 0.0f, 0.0f);
    s->p1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    s->p2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// ---- the walk of a chunk -----------------------------------------------------------------------------
// Each group's next entry = the lowest set bit of its 64-bit SGPR mask; an exhausted group gets the
// sentinel slot.  Clearing the bit just found is ONE scalar instruction (s_bitset0_b64; `m &= m - 1` is
// three) — for an exhausted group the index 64 addresses bit 0 of a mask that is already zero.  The four
// slots are packed into one SGPR (a lane extracts its group's with one v_bfe_u32), and the packed word of
// step i + 1 is formed while step i computes: the scalar work is off the critical path of the step's
// first LDS read and the loop's back edge is one compare against "all four exhausted".  (20 scalar
// instructions per step instead of 31: forward 200 -> 180 us at C2)
#define GS_WALK_STEP(m, e)                                        \
    const int e = (m) != 0ull ? (int)__builtin_ctzll(m) : kChunk; \
    asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(e));
#define GS_WALK_PACK(ep)                                                                             \
    uint32_t ep;                                                                                     \
    {                                                                                                \
        GS_WALK_STEP(m0, e0_)                                                                        \
        GS_WALK_STEP(m1, e1_)                                                                        \
        GS_WALK_STEP(m2, e2_)                                                                        \
        GS_WALK_STEP(m3, e3_)                                                                        \
        ep = (uint32_t)e0_ | ((uint32_t)e1_ << 8) | ((uint32_t)e2_ << 16) | ((uint32_t)e3_ << 24);   \
    }
[[maybe_unused]] constexpr uint32_t kWalkDone = 0x40404040u;   // the packed word of four exhausted groups (slot kChunk = 64)

// ---------------------------------------------------------------------------------------------
// ILP: entries of a group's list taken per step (1: full frames; 2: frames of few tiles, see walk2).
// CK: checkpoints for the segmented backward (k_rasterize_backward_seg) — per tile max_seg records of 256
// pixels; record k >= 1 describes the state in front of entry range.x + (k << seg_shift) of the tile's list,
// record 0 the end of the list.  What a record holds is the state of the BACKWARD's recurrence, not the
// forward's: the reference clamps alpha at 0.999 here (gsplat_cpu.cpp:220) and at 0.99 there (:338), and its
// backward unwinds T = T_final * prod 1 / (1 - min(alpha, 0.99)) and sums the colour buffer with those
// alphas and that T — behind an entry with alpha > 0.99 ("hot") neither is the forward's value.  With
//   g_j = (1 - alpha_j) / (1 - 0.99) for a hot entry, 1 otherwise;  invG(k) = prod_{j < k} 1 / g_j
// the backward's transmittance in front of entry k is T_fwd(k) * invG(k) / invG(end) and its buffer behind
// k is (S(end) - S(k)) / invG(end), S(k) = sum_{j < k} colour_j min(alpha_j, 0.99) T_fwd(j) invG(j):
//   record k = {T_fwd(k) invG(k), S(k)},  record 0 = {invG(end), S(end)}.
// Without hot entries invG = 1 and S is the image's own colour sum, bit for bit.
// GS_BWD_PRIO: a backward wave sets its own issue priority (s_setprio) once per chunk.  The arbiter of a SIMD issues
// oldest first among equal priorities and one wave gets a quarter of the SIMD's slots at most (scripts/wave_timeline.py),
// so a launch of ~2 rounds of equal tiles (1080p: 8160 tiles on 4096 .. 4608 slots) ends with the YOUNGEST waves
// walking alone.  Longest remaining work first is what minimises a makespan: the launch's FINAL residents — the last
// `slots` workgroups, final_first = grid - slots — take the priority of the quarter of their list that is left
// (3, 2, 1, 0: they progress together and end together); every wave in front of them keeps 3 (oldest first among
// themselves: they finish staggered and hand their slots on early).  Round 6, same box: C2 backward 0.2257 -> 0.2112 ms,
// C3 1.59 -> 1.57 ms; equalising ALL waves: 0.2123 at four per SIMD but 0.2346 at 4.5; the same in the forward
// (six rounds of quadrant waves): nothing, left out.  profiles/HISTORY.md.
#ifndef GS_BWD_PRIO
#define GS_BWD_PRIO 1
#endif
__device__ __forceinline__ void wave_prio(int final_first, int rem, int tot) {
#if GS_BWD_PRIO
    if (final_first < 0) return;   // (launches that do not take part: pieces, small frames)
    int q = 3;
    if ((int)blockIdx.x >= final_first) q = (4 * rem > 3 * tot) ? 3 : (2 * rem > tot) ? 2 : (4 * rem > tot) ? 1 : 0;
    q = __builtin_amdgcn_readfirstlane(q);
    if (q == 3) __builtin_amdgcn_s_setprio(3);
    else if (q == 2) __builtin_amdgcn_s_setprio(2);
    else if (q == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
#endif
}
#ifndef GS_FWD_EARLY_ALIVE
#define GS_FWD_EARLY_ALIVE 48
#endif
constexpr int kEarlyAlive = GS_FWD_EARLY_ALIVE;
template <bool EXACT, int ILP, bool CK>
__global__ void __launch_bounds__(64)
k_rasterize_forward(int W, int H, int tiles_x, int num_tiles, const int32_t *__restrict__ order,
                    const int32_t *__restrict__ ids, const uint16_t *__restrict__ masks,
                    const int2 *__restrict__ bins,
                    const float4 *__restrict__ packed, float bg0, float bg1, float bg2,
                    const float *__restrict__ bg_dev, float *__restrict__ out_img,
                    float *__restrict__ final_Ts, int32_t *__restrict__ final_idx,
                    float *__restrict__ out_clamped, float4 *__restrict__ ckpt, int seg_shift,
                    int max_seg) {
    __shared__ SRec stage[kChunk + 1];
    __shared__ uint64_t exp_tab[EXACT ? kExpTabLds : 1];
#if GS_FWD_QWALK
    // [group][rank] -> the slot's BYTE OFFSET in stage[] (slot * 48) as a 32-bit word: the step's record address is
    // the queue word itself — no zero-extension, no multiply per step (round 6: 45 -> 43 VALU per step)
    __shared__ __attribute__((aligned(16))) uint32_t fq[4 * kChunk + 16];
    constexpr uint32_t kFqFill = (uint32_t)(kChunk * sizeof(SRec));   // the sentinel slot
    static_assert(sizeof(SRec) == 48, "the last-contributor conversion divides by 48");
#endif
    const int lane = threadIdx.x;
    int tile, qx0, qy0;
    if (!decode_wave<1>(blockIdx.x, num_tiles, tiles_x, W, H, order, tile, qx0, qy0)) return;
    GS_WLOG_T0
    if (bg_dev) {  // background handed over as a device tensor (no host copy, no sync)
        bg0 = bg_dev[0]; bg1 = bg_dev[1]; bg2 = bg_dev[2];
    }
    if (EXACT) load_exp_table(exp_tab, lane, 64);
    if (lane == 0) stage_sentinel(&stage[kChunk]);

    const int grp = lane >> 4, li = lane & 15;
    const uint32_t gsh = 8u * (uint32_t)grp;   // (the group's byte of the packed slot word: scalar walk only)
    (void)gsh;
    const int px = qx0 + 4 * (grp & 1) + (li & 3), py = qy0 + 4 * (grp >> 1) + (li >> 2);
    const bool inimg = px < W && py < H;
    const float pxf = (float)px;
    // NaN once the pixel is finished (or outside the image): a NaN row makes sigma NaN
    float pyf = inimg ? (float)py : qnan();
    float T = 1.0f, a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    // (CK) the backward's view of the state, see above.  invG only changes at a hot entry, so between two of them
    // S grows by invG times what the image's colour sum grows by:  S = sb + invG * (a - cb), rebased at every hot
    // entry (sb <- S, cb <- a, invG <- invG / g) — nothing to do on any other step, and without a hot entry
    // S = 0 + 1 * (a - 0): the image's own sums, bit for bit.  (Kept as a second set of sums on every step it cost
    // the forward 87 -> 98 us on the 6000-Gaussian training frame, from a wave's first hot entry on still 170 ->
    // 186 us on config 2's opaque Gaussians; profiles/HISTORY.md.)
    float invG = 1.0f, sb0 = 0.0f, sb1 = 0.0f, sb2 = 0.0f, cb0 = 0.0f, cb1 = 0.0f, cb2 = 0.0f;
    // called right AFTER the colour sums took the entry in (alpha > 0.99 in some lane; Tf: T in front of it):
    // S_after = S_before + c min(alpha, 0.99) Tf invG, with a_before = a - alpha Tf c
    auto rebase = [&](float alpha, float Tf, const float4 &c) {
        asm volatile("; hot entry");
        if (alpha > 0.99f) {
            const float dT = (0.99f - alpha) * Tf;
            sb0 = sb0 + invG * ((a0 - cb0) + dT * c.x);
            sb1 = sb1 + invG * ((a1 - cb1) + dT * c.y);
            sb2 = sb2 + invG * ((a2 - cb2) + dT * c.z);
            cb0 = a0; cb1 = a1; cb2 = a2;
            invG = invG * ((1.0f - 0.99f) * __builtin_amdgcn_rcpf(1.0f - alpha));
        }
    };
    auto record = [&](float first) {
        return make_float4(first, sb0 + invG * (a0 - cb0), sb1 + invG * (a1 - cb1), sb2 + invG * (a2 - cb2));
    };
    int last = -1;   // list index of the last composited entry
    int le = -1;     // ... as a slot of the current chunk (turned into an index once per chunk)
    const uint32_t *lq = nullptr;   // ... as a place in the group's queue (queue walk)
    (void)le; (void)lq;

    const int2 range = bins[tile];
    const int ox = qx0 & (GS_TILE - 1), oy = qy0 & (GS_TILE - 1);   // the quadrant's offset in its tile
    // the next chunk's entry of this lane, gathered one chunk ahead (registers, not a struct: a
    // conditionally filled aggregate ends up in scratch) — only entries whose coverage mask touches
    // one of this wave's four blocks are gathered at all (43 % of a tile's list at C2)
    float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0;
    uint32_t ntouch = 0u;
    if (range.x + lane < range.y) {
        ntouch = touch_from_mask<1>(masks[range.x + lane], ox, oy);
        if (ntouch) {
            const size_t g = (size_t)ids[range.x + lane];
            n0 = packed[3 * g + 0]; n1 = packed[3 * g + 1]; n2 = packed[3 * g + 2];
        }
    }
    // the lane's pixel inside its tile, row-major: the index of a checkpoint record
    const int pid = ((py & (GS_TILE - 1)) << 4) | (px & (GS_TILE - 1));
    if (CK) ckpt += (size_t)tile * max_seg * (GS_TILE * GS_TILE) + pid;
    for (int c0 = range.x; c0 < range.y; c0 += kChunk) {
        const uint64_t alive = __builtin_amdgcn_ballot_w64(pyf == pyf);
        if (alive == 0ull) break;
        if (CK && c0 != range.x && ((c0 - range.x) & ((1 << seg_shift) - 1)) == 0) {
            const int k = (c0 - range.x) >> seg_shift;
            if (k < max_seg && inimg)
                ckpt[(size_t)k * (GS_TILE * GS_TILE)] = record(T * invG);
        }
        __syncthreads();  // previous chunk fully consumed (single-wave workgroup: cheap)
        const uint32_t touch = ntouch;
        if (touch) {
            // (the staged C carries the "rectangle binds" flag — bit 0 of the threshold word — in its SIGN as well:
            // C > 0, the walks take |C| (a source modifier) and the one-entry walk tests the flag with ONE integer
            // compare instead of and + compare)
            stage[lane].p0 = n0;
            stage[lane].p1 = make_float4(__uint_as_float(__float_as_uint(n1.x) | (__float_as_uint(n1.z) << 31)),
                                         n1.y, n1.z, n1.w);
            stage[lane].p2 = n2;
        }
        // (a walk specialised per chunk on "no staged entry's rectangle cuts its sigma_max ellipse" — two
        // VALU and a branch less per step, as in the backward — measured 185.1 against 185.8 us: the
        // general walk is the one that runs; the switch stays for the next look.  Round 6: a rectangle of three
        // sigmas cuts the sigma_max ellipse of every Gaussian with an opacity above ~0.35 — at C2 hardly a chunk is
        // free of them)
        constexpr bool kChunkBinds = GS_FWD_CHUNK_BINDS != 0 && ILP == 1 && !CK;
        const bool chunk_binds = !kChunkBinds ||
            __builtin_amdgcn_ballot_w64(touch != 0u && (__float_as_uint(n1.z) & 1u) != 0u) != 0ull;
        const bool chunk_hot = CK && __builtin_amdgcn_ballot_w64(touch != 0u && n1.y > 0.99f) != 0ull;
        // a block whose 16 pixels are all finished walks nothing
        uint64_t m0 = (alive & 0x000000000000FFFFull) ? __builtin_amdgcn_ballot_w64((touch & 1u) != 0u) : 0ull;
        uint64_t m1 = (alive & 0x00000000FFFF0000ull) ? __builtin_amdgcn_ballot_w64((touch & 2u) != 0u) : 0ull;
        uint64_t m2 = (alive & 0x0000FFFF00000000ull) ? __builtin_amdgcn_ballot_w64((touch & 4u) != 0u) : 0ull;
        uint64_t m3 = (alive & 0xFFFF000000000000ull) ? __builtin_amdgcn_ballot_w64((touch & 8u) != 0u) : 0ull;
#if GS_FWD_QWALK
        // the four groups' lists as QUEUES of slot numbers in LDS (as backward_wave_q builds its sixteen): a step
        // is one byte read with a per-group address instead of four scalar find-first-set chains
        int nsteps = 0;
        {
            fq[lane] = kFqFill;
            fq[kChunk + lane] = kFqFill;
            fq[2 * kChunk + lane] = kFqFill;
            fq[3 * kChunk + lane] = kFqFill;
            if (lane < 16) fq[4 * kChunk + lane] = kFqFill;
#define GS_FQ(g, m)                                                                                         \
    if (__builtin_amdgcn_inverse_ballot_w64(m))                                                             \
        fq[(g) * kChunk + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)((m) >> 32),                             \
                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)(m), 0u))] =   \
            (uint32_t)(lane * (int)sizeof(SRec));
            GS_FQ(0, m0) GS_FQ(1, m1) GS_FQ(2, m2) GS_FQ(3, m3)
#undef GS_FQ
            nsteps = max(max(__builtin_popcountll(m0), __builtin_popcountll(m1)),
                         max(__builtin_popcountll(m2), __builtin_popcountll(m3)));
        }
#endif
        __syncthreads();
        GS_STAT(3, __builtin_popcountll(m0) + __builtin_popcountll(m1) + __builtin_popcountll(m2) + __builtin_popcountll(m3));
        GS_STAT(4, 1);
        ntouch = 0u;
        if (c0 + kChunk + lane < range.y) {
            ntouch = touch_from_mask<1>(masks[c0 + kChunk + lane], ox, oy);
            if (ntouch) {
                const size_t g = (size_t)ids[c0 + kChunk + lane];
                n0 = packed[3 * g + 0]; n1 = packed[3 * g + 1]; n2 = packed[3 * g + 2];
            }
        }
        auto walk = [&](auto binds_tag, auto hot_tag, auto early_tag) {
          constexpr bool BINDS = decltype(binds_tag)::value;
          constexpr bool HOT = CK && decltype(hot_tag)::value;
          constexpr bool EARLY = decltype(early_tag)::value;
          static_assert(GS_FWD_QWALK != 0, "the one-entry walk reads its queue (the scalar walk: walk2 only)");
          // One step: the group's next entry (e: the staged record's byte offset, see fq).  The walk takes the queue
          // words in PAIRS (one 8-byte LDS read and one address increment per two steps) and runs FOUR steps per
          // iteration: without a join between two steps the transmittance of a step lands in the register the step
          // before read it from (a single-step loop copied it every step), and the last contributor is remembered as
          // the queue WORD, which is in a register anyway (round 6: 44 -> ~41.5 VALU per step, profiles/HISTORY.md).
          const auto step = [&](const int e) {
            const SRec &rec = *reinterpret_cast<const SRec *>(reinterpret_cast<const char *>(stage) + e);
            const float4 q0 = rec.p0, q1 = rec.p1, q2 = rec.p2;
            const uint32_t sbits = __float_as_uint(q1.z);
            GS_STAT(0, 1);
            const float dx = q0.x - pxf, dy = q0.y - pyf;
            // sigma = 0.5f * (A*x*x + C*y*y) + B*x*y, gsplat_cpu.cpp:213-217 (same op order)
            float sg = (q0.z * dx) * dx + (__builtin_fabsf(q1.x) * dy) * dy;
            sg = 0.5f * sg;
            sg = sg + (q0.w * dx) * dy;
            // (lane predicates as 64-bit scalar masks + inverse_ballot: one compare per predicate)
            const uint64_t mbinds = BINDS ? __builtin_amdgcn_ballot_w64((int)__float_as_uint(q1.x) < 0) : 0ull;
            if (BINDS && mbinds != 0ull) {
                // rare: some group's Gaussian has a rectangle that cuts its sigma_max ellipse —
                // apply the rectangle per pixel (and turn -0.0 into +0.0, see gs_pack_splats)
                asm volatile("; rectangle binds");
                if (__builtin_amdgcn_inverse_ballot_w64(mbinds)) {
                    const uint32_t rx = __float_as_uint(q1.w), ry = __float_as_uint(q2.w);
                    const bool in = (uint32_t)px >= (rx & 0xFFFFu) && (uint32_t)px < (rx >> 16) &&
                                    (uint32_t)py >= (ry & 0xFFFFu) && (uint32_t)py < (ry >> 16);
                    sg = in ? sg + 0.0f : qnan();
                }
            }
            // 0 <= sigma <= sigma_max as ONE unsigned compare of the bit patterns
            const uint64_t mneed = __builtin_amdgcn_ballot_w64(__float_as_uint(sg) <= sbits);
            // Nobody needs the entry: 0.05 % of the steps at C2, 10 % at C3 (quadrants most of whose pixels are
            // finished).  The exit's join keeps the transmittance in two registers — a copy per step — so only the
            // walk of such chunks (EARLY) has it.
            if (EARLY && mneed == 0ull) return;
            GS_STAT(1, mneed != 0ull);
            GS_STAT(2, __builtin_popcountll(mneed));
            // every lane evaluates the exponential (masking lanes off saves no issue cycle); lanes
            // that do not need the entry — sigma possibly NaN — are discarded by `ok`
            const float vis = gs_exp<EXACT>(-sg, exp_tab);
            // gsplat_cpu.cpp:220-236: alpha = min(0.999, opacity*vis); skip if alpha < 1/255;
            // nextT = T*(1-alpha); nextT <= 1e-4 -> pixel done (Gaussian not rendered).  A skipped
            // pixel is given alpha = 0, which composites exactly nothing.
            float alpha = q1.y * vis;
            alpha = __builtin_amdgcn_fmed3f(alpha, 0.0f, 0.999f);
            bool ok = __builtin_amdgcn_inverse_ballot_w64(
                mneed & __builtin_amdgcn_ballot_w64(alpha >= (1.0f / 255.0f)));
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
            if (HOT && __builtin_amdgcn_ballot_w64(alpha > 0.99f) != 0ull) rebase(alpha, T, q2);
            T = nT;
            // the last contributor: the queue word (in a register pair for two steps anyway) — or, where the word is
            // read one step ahead, the address the walk holds: that e dies with its reads, no copy at the back edge
            le = ok ? e : le;
          };
          // four steps per iteration on two register pairs that swap roles (a pair carried across a two-step
          // iteration was copied at its back edge)
          const uint2 *myq2 = reinterpret_cast<const uint2 *>(&fq[grp * kChunk]);
          uint2 pa = myq2[0];
          int k = 0;
          for (; k + 3 < nsteps; k += 4) {
            const uint2 pb = myq2[(k >> 1) + 1];
            step((int)pa.x);
            step((int)pa.y);
            pa = myq2[(k >> 1) + 2];
            step((int)pb.x);
            step((int)pb.y);
          }
          if (k < nsteps) {   // one to three steps left (nsteps is wave-uniform)
            step((int)pa.x);
            if (k + 1 < nsteps) {
                step((int)pa.y);
                if (k + 2 < nsteps) step((int)myq2[(k >> 1) + 1].x);
            }
          }
        };
        // Two entries of a group's list per step: the second entry's sigma / exponential / alpha do not
        // depend on the first one's compositing, so the two chains overlap in ONE wave's instruction
        // stream — what the other waves of a SIMD do for a full frame and nobody does on a frame of a few
        // hundred tiles, where a lone wave pays the full latency of every dependent step (measured 170 ns
        // per step against 40 ns of issue, profiles/timeline_sweep_*_r04.json).  The compositing itself
        // stays in list order (a, then b with the transmittance a leaves), so the bits are the same.
        auto walk2 = [&](auto binds_tag, auto hot_tag) {
          constexpr bool BINDS = decltype(binds_tag)::value;
          constexpr bool HOT = CK && decltype(hot_tag)::value;
#if GS_FWD_QWALK
          // (the queue holds the group's slots in list order, the sentinel slot behind them: entries 2k and
          // 2k + 1 of the queue are this step's pair — on a frame of lone waves every instruction of a step
          // counts, and the forty scalar ones of the two mask walks were half of them)
          const uint2 *myq2 = reinterpret_cast<const uint2 *>(&fq[grp * kChunk]);
          uint2 pair_next = myq2[0];
          for (int k = 0; k < (nsteps + 1) / 2; k++) {
            const int ea = (int)pair_next.x, eb = (int)pair_next.y;   // (byte offsets: see fq)
            const SRec &ra = *reinterpret_cast<const SRec *>(reinterpret_cast<const char *>(stage) + ea);
            const SRec &rb = *reinterpret_cast<const SRec *>(reinterpret_cast<const char *>(stage) + eb);
            const float4 qa0 = ra.p0, qa1 = ra.p1, qa2 = ra.p2;
            const float4 qb0 = rb.p0, qb1 = rb.p1, qb2 = rb.p2;
            pair_next = myq2[k + 1];
#else
          uint32_t epa_next, epb_next;
          { GS_WALK_PACK(wa0_) epa_next = wa0_; }
          { GS_WALK_PACK(wb0_) epb_next = wb0_; }
          while (epa_next != kWalkDone) {   // (slot a exhausted: slot b too)
            const int ea = (int)((epa_next >> gsh) & 0xFFu), eb = (int)((epb_next >> gsh) & 0xFFu);
            const float4 qa0 = stage[ea].p0, qa1 = stage[ea].p1, qa2 = stage[ea].p2;
            const float4 qb0 = stage[eb].p0, qb1 = stage[eb].p1, qb2 = stage[eb].p2;
            { GS_WALK_PACK(wa1_) epa_next = wa1_; }
            { GS_WALK_PACK(wb1_) epb_next = wb1_; }
            asm volatile("" : "+s"(epa_next), "+s"(epb_next));
#endif
            const uint32_t sba = __float_as_uint(qa1.z), sbb = __float_as_uint(qb1.z);
            GS_STAT(0, 1);
            const float dxa = qa0.x - pxf, dya = qa0.y - pyf, dxb = qb0.x - pxf, dyb = qb0.y - pyf;
            float sga = (qa0.z * dxa) * dxa + (__builtin_fabsf(qa1.x) * dya) * dya;   // (|C|: see the staging)
            float sgb = (qb0.z * dxb) * dxb + (__builtin_fabsf(qb1.x) * dyb) * dyb;
            sga = 0.5f * sga;
            sgb = 0.5f * sgb;
            sga = sga + (qa0.w * dxa) * dya;
            sgb = sgb + (qb0.w * dxb) * dyb;
            const uint64_t mbinds = BINDS ? __builtin_amdgcn_ballot_w64(((sba | sbb) & 1u) != 0u) : 0ull;
            if (BINDS && mbinds != 0ull) {
                asm volatile("; rectangle binds");
                if (sba & 1u) {
                    const uint32_t rx = __float_as_uint(qa1.w), ry = __float_as_uint(qa2.w);
                    const bool in = (uint32_t)px >= (rx & 0xFFFFu) && (uint32_t)px < (rx >> 16) &&
                                    (uint32_t)py >= (ry & 0xFFFFu) && (uint32_t)py < (ry >> 16);
                    sga = in ? sga + 0.0f : qnan();
                }
                if (sbb & 1u) {
                    const uint32_t rx = __float_as_uint(qb1.w), ry = __float_as_uint(qb2.w);
                    const bool in = (uint32_t)px >= (rx & 0xFFFFu) && (uint32_t)px < (rx >> 16) &&
                                    (uint32_t)py >= (ry & 0xFFFFu) && (uint32_t)py < (ry >> 16);
                    sgb = in ? sgb + 0.0f : qnan();
                }
            }
            const uint64_t mna = __builtin_amdgcn_ballot_w64(__float_as_uint(sga) <= sba);
            const uint64_t mnb = __builtin_amdgcn_ballot_w64(__float_as_uint(sgb) <= sbb);
            if ((mna | mnb) == 0ull) continue;
            GS_STAT(1, 1);
            GS_STAT(2, __builtin_popcountll(mna) + __builtin_popcountll(mnb));
            const float visa = gs_exp<EXACT>(-sga, exp_tab);
            const float visb = gs_exp<EXACT>(-sgb, exp_tab);
            float aa = __builtin_amdgcn_fmed3f(qa1.y * visa, 0.0f, 0.999f);
            float ab = __builtin_amdgcn_fmed3f(qb1.y * visb, 0.0f, 0.999f);
            bool oka = __builtin_amdgcn_inverse_ballot_w64(
                mna & __builtin_amdgcn_ballot_w64(aa >= (1.0f / 255.0f)));
            bool okb = __builtin_amdgcn_inverse_ballot_w64(
                mnb & __builtin_amdgcn_ballot_w64(ab >= (1.0f / 255.0f)));
            aa = oka ? aa : 0.0f;
            ab = okb ? ab : 0.0f;
            float nTa = T * (1.0f - aa);
            float nTb = nTa * (1.0f - ab);
            if (__builtin_amdgcn_ballot_w64(nTa <= 1e-4f || nTb <= 1e-4f) != 0ull) {
                asm volatile("; pixel saturates");
                if (nTa <= 1e-4f) {          // a finishes the pixel: neither a nor b is rendered
                    pyf = qnan(); aa = 0.0f; ab = 0.0f; nTa = T; nTb = T; oka = false; okb = false;
                } else if (nTb <= 1e-4f) {   // b finishes it: a is rendered, b is not
                    pyf = qnan(); ab = 0.0f; nTb = nTa; okb = false;
                }
            }
            const float wa = aa * T;
            a0 = a0 + wa * qa2.x;
            a1 = a1 + wa * qa2.y;
            a2 = a2 + wa * qa2.z;
            if (HOT && __builtin_amdgcn_ballot_w64(aa > 0.99f) != 0ull) rebase(aa, T, qa2);
            const float wb = ab * nTa;
            a0 = a0 + wb * qb2.x;
            a1 = a1 + wb * qb2.y;
            a2 = a2 + wb * qb2.z;
            if (HOT && __builtin_amdgcn_ballot_w64(ab > 0.99f) != 0ull) rebase(ab, nTa, qb2);
            T = nTb;
#if GS_FWD_QWALK
            lq = okb ? &myq2[k + 1].x : (oka ? &myq2[k].y : lq);   // (one word past the entry's, as in walk)
#else
            le = okb ? eb : (oka ? ea : le);
#endif
          }
        };
        // (kChunkBinds is off: the general walk)  A chunk none of whose entries has an opacity above 0.99 cannot
        // hold a hot entry (alpha <= opacity): such chunks walk without looking
        if constexpr (ILP == 2) {
            if (chunk_hot) walk2(std::true_type{}, std::true_type{}); else walk2(std::true_type{}, std::false_type{});
        } else {
            // (fewer than kEarlyAlive pixels of the quadrant still open: the walk that leaves a step nobody needs)
            const bool early = __builtin_popcountll(alive) < kEarlyAlive;
            if (chunk_hot) walk(std::true_type{}, std::true_type{}, std::true_type{});
            else if (!chunk_binds) walk(std::false_type{}, std::false_type{}, std::true_type{});
            else if (early) walk(std::true_type{}, std::false_type{}, std::true_type{});
            else walk(std::true_type{}, std::false_type{}, std::false_type{});
        }
#if GS_FWD_QWALK
        // (the last composited entry as its queue word — walk: le — or one word past it — walk2: lq —; the word is
        // a stage offset, a multiple of 48 up to 63 * 48: the slot by multiply-and-shift, once per chunk)
        if (lq) last = c0 + (int)((lq[-1] * 43691u) >> 21);
        else if (ILP == 1 && le >= 0) last = c0 + (int)(((uint32_t)le * 43691u) >> 21);
        lq = nullptr;
#else
        last = le >= 0 ? c0 + le : last;
#endif
        le = -1;
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
        if (CK) ckpt[0] = record(invG);
    }
    GS_WLOG_T1(tile, range.y - range.x);
}


// ---------------------------------------------------------------------------------------------
// Round 6: the full-frame forward with the chunk COMPACTED before it is staged (k_rasterize_forward_c).
// A quadrant wave of k_rasterize_forward stages its tile's list 64 entries at a time although only the
// entries whose coverage mask touches its 8 x 8 pixels are gathered and walked — 43 % of them at C2: a
// wave runs 2.3 chunks (two barriers, four ballots, a queue build each) for every 64 entries it walks, and
// its four groups share out ~27 entries per chunk, whose longest queue sets the chunk's steps.  Here the
// touching entries are collected first — (id, list index, touch bits) appended to a ring in LDS by ballot +
// mbcnt, 64-entry windows of the list at a time, the windows' ids and masks loaded three windows ahead — and a
// chunk is the next 64 ring entries: every staged slot is walked, the four queues hold 64 entries' worth.
// Per pixel the entries arrive in list order with the same arithmetic: the image, final_Ts and final_idx
// (the LIST index, kept beside the slot) are bit for bit k_rasterize_forward's
// (tests/test_gpu_forward_compact.py).  ILP = 1, no checkpoints: full frames.  Measured in round 6: 162 -> 160 us
// at C2, 1119 -> 1191 us at C3 — selected by flag bits 23..24 = 3 only (profiles/HISTORY.md).
constexpr int kRing = 128;   // ring capacity: at most 63 left over + 64 appended
template <bool EXACT>
__global__ void __launch_bounds__(64)
k_rasterize_forward_c(int W, int H, int tiles_x, int num_tiles, const int32_t *__restrict__ order,
                      const int32_t *__restrict__ ids, const uint16_t *__restrict__ masks,
                      const int2 *__restrict__ bins, const float4 *__restrict__ packed, float bg0, float bg1,
                      float bg2, const float *__restrict__ bg_dev, float *__restrict__ out_img,
                      float *__restrict__ final_Ts, int32_t *__restrict__ final_idx,
                      float *__restrict__ out_clamped) {
    __shared__ SRec stage[kChunk + 1];
    __shared__ uint64_t exp_tab[EXACT ? kExpTabLds : 1];
    __shared__ __attribute__((aligned(16))) uint8_t fq[4 * kChunk + 16];   // [group][rank] -> slot
    __shared__ int32_t ring_id[kRing];
    __shared__ uint32_t ring_meta[kRing];      // (list index - range.x) << 4 | touch bits
    __shared__ int32_t slot_idx[kChunk];       // list index of the staged slots
    const int lane = threadIdx.x;
    int tile, qx0, qy0;
    if (!decode_wave<1>(blockIdx.x, num_tiles, tiles_x, W, H, order, tile, qx0, qy0)) return;
    if (bg_dev) { bg0 = bg_dev[0]; bg1 = bg_dev[1]; bg2 = bg_dev[2]; }
    if (EXACT) load_exp_table(exp_tab, lane, 64);
    if (lane == 0) stage_sentinel(&stage[kChunk]);
    const int grp = lane >> 4, li = lane & 15;
    const int px = qx0 + 4 * (grp & 1) + (li & 3), py = qy0 + 4 * (grp >> 1) + (li >> 2);
    const bool inimg = px < W && py < H;
    const float pxf = (float)px;
    float pyf = inimg ? (float)py : qnan();   // NaN once the pixel is finished (or outside the image)
    float T = 1.0f, a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    int last = -1, le = -1;
    const int2 range = bins[tile];
    const int ox = qx0 & (GS_TILE - 1), oy = qy0 & (GS_TILE - 1);
    // ---- the list's windows, three ahead in registers: {id, mask} of entry cursor + 64 k + lane ----
    int cursor = range.x;                      // first entry of window 0
    int32_t wid0 = 0, wid1 = 0, wid2 = 0;
    uint32_t wm0 = 0u, wm1 = 0u, wm2 = 0u;
    auto load_window = [&](int first, int32_t &id, uint32_t &m) {
        id = 0; m = 0u;
        if (first + lane < range.y) { id = ids[first + lane]; m = masks[first + lane]; }
    };
    load_window(cursor, wid0, wm0);
    load_window(cursor + kChunk, wid1, wm1);
    load_window(cursor + 2 * kChunk, wid2, wm2);
    int head = 0, count = 0;                   // ring: entries [head, head + count)
    // append windows until a chunk's worth is in the ring (or the list is exhausted)
    auto fill = [&]() {
        while (count < kChunk && cursor < range.y) {
            const uint32_t t = touch_from_mask<1>(wm0, ox, oy);   // (mask 0 beyond the list: no touch)
            const uint64_t m = __builtin_amdgcn_ballot_w64(t != 0u);
            if (t) {
                const int pos = (head + count + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))) & (kRing - 1);
                ring_id[pos] = wid0;
                ring_meta[pos] = ((uint32_t)(cursor + lane - range.x) << 4) | t;
            }
            count += __builtin_popcountll(m);
            cursor += kChunk;
            wid0 = wid1; wm0 = wm1; wid1 = wid2; wm1 = wm2;
            load_window(cursor + 2 * kChunk, wid2, wm2);
        }
    };
    // the next chunk of this lane, gathered one chunk ahead
    float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0;
    uint32_t nmeta = 0u;
    int nb = 0;
    auto take = [&]() {
        wave_sync();                            // the ring entries fill() wrote, for other lanes
        nb = min(count, kChunk);
        nmeta = 0u;
        if (lane < nb) {
            const int pos = (head + lane) & (kRing - 1);
            const size_t g = (size_t)ring_id[pos];
            nmeta = ring_meta[pos];
            n0 = packed[3 * g + 0]; n1 = packed[3 * g + 1]; n2 = packed[3 * g + 2];
        }
        head = (head + nb) & (kRing - 1);
        count -= nb;
        wave_sync();                            // (before fill() overwrites ring slots)
    };
    fill();
    take();
    while (nb > 0) {
        const uint64_t alive = __builtin_amdgcn_ballot_w64(pyf == pyf);
        if (alive == 0ull) break;
        __syncthreads();  // previous chunk fully consumed (single-wave workgroup: cheap)
        const uint32_t touch = nmeta & 15u;
        if (touch) {
            stage[lane].p0 = n0;
            stage[lane].p1 = n1;
            stage[lane].p2 = n2;
            slot_idx[lane] = (int32_t)(nmeta >> 4);
        }
        uint64_t m0 = (alive & 0x000000000000FFFFull) ? __builtin_amdgcn_ballot_w64((touch & 1u) != 0u) : 0ull;
        uint64_t m1 = (alive & 0x00000000FFFF0000ull) ? __builtin_amdgcn_ballot_w64((touch & 2u) != 0u) : 0ull;
        uint64_t m2 = (alive & 0x0000FFFF00000000ull) ? __builtin_amdgcn_ballot_w64((touch & 4u) != 0u) : 0ull;
        uint64_t m3 = (alive & 0xFFFF000000000000ull) ? __builtin_amdgcn_ballot_w64((touch & 8u) != 0u) : 0ull;
        int nsteps = 0;
        {
            reinterpret_cast<uint32_t *>(fq)[lane] = kChunk * 0x01010101u;
            if (lane < 4) reinterpret_cast<uint32_t *>(fq)[kChunk + lane] = kChunk * 0x01010101u;
#define GS_FQ(g, m)                                                                                         \
    if (__builtin_amdgcn_inverse_ballot_w64(m))                                                             \
        fq[(g) * kChunk + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)((m) >> 32),                             \
                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)(m), 0u))] =   \
            (uint8_t)lane;
            GS_FQ(0, m0) GS_FQ(1, m1) GS_FQ(2, m2) GS_FQ(3, m3)
#undef GS_FQ
            nsteps = max(max(__builtin_popcountll(m0), __builtin_popcountll(m1)),
                         max(__builtin_popcountll(m2), __builtin_popcountll(m3)));
        }
        __syncthreads();
        GS_STAT(3, __builtin_popcountll(m0) + __builtin_popcountll(m1) + __builtin_popcountll(m2) + __builtin_popcountll(m3));
        GS_STAT(4, 1);
        // the next chunk: its ring entries, then its gather — in flight under the walk
        fill();
        take();
        {
            const uint8_t *myq = &fq[grp * kChunk];
            int e_next = myq[0];
            for (int k = 0; k < nsteps; k++) {
                const int e = e_next;
                const float4 q0 = stage[e].p0, q1 = stage[e].p1, q2 = stage[e].p2;
                e_next = myq[k + 1];
                const uint32_t sbits = __float_as_uint(q1.z);
                GS_STAT(0, 1);
                const float dx = q0.x - pxf, dy = q0.y - pyf;
                // sigma = 0.5f * (A*x*x + C*y*y) + B*x*y, gsplat_cpu.cpp:213-217 (same op order)
                float sg = (q0.z * dx) * dx + (q1.x * dy) * dy;
                sg = 0.5f * sg;
                sg = sg + (q0.w * dx) * dy;
                const uint64_t mbinds = __builtin_amdgcn_ballot_w64((sbits & 1u) != 0u);
                if (mbinds != 0ull) {
                    asm volatile("; rectangle binds");
                    if (__builtin_amdgcn_inverse_ballot_w64(mbinds)) {
                        const uint32_t rx = __float_as_uint(q1.w), ry = __float_as_uint(q2.w);
                        const bool in = (uint32_t)px >= (rx & 0xFFFFu) && (uint32_t)px < (rx >> 16) &&
                                        (uint32_t)py >= (ry & 0xFFFFu) && (uint32_t)py < (ry >> 16);
                        sg = in ? sg + 0.0f : qnan();
                    }
                }
                const uint64_t mneed = __builtin_amdgcn_ballot_w64(__float_as_uint(sg) <= sbits);
                if (mneed == 0ull) continue;
                GS_STAT(1, 1);
                GS_STAT(2, __builtin_popcountll(mneed));
                const float vis = gs_exp<EXACT>(-sg, exp_tab);
                float alpha = q1.y * vis;
                alpha = __builtin_amdgcn_fmed3f(alpha, 0.0f, 0.999f);
                bool ok = __builtin_amdgcn_inverse_ballot_w64(
                    mneed & __builtin_amdgcn_ballot_w64(alpha >= (1.0f / 255.0f)));
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
                le = ok ? e : le;
            }
        }
        last = le >= 0 ? range.x + slot_idx[le] : last;
        le = -1;
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

// Sums of nine per-lane values over each 16-lane DPP row, in registers: a transposing butterfly that
// folds the row from the outside in.  The two cross-quad stages use DPP adds whose BANK MASK leaves
// part of the destination untouched, so two values share a register without any select:
//   A  r = a + row_mirror(a) everywhere, then r = b + row_mirror(b) in banks 2,3 only: lanes 0-7
//      hold eight partial sums of a, lanes 8-15 of b                       (4 pairs x 2 + 1 = 9 VALU)
//   B  s = r01 + row_half_mirror(r01), then banks 1,3 <- r23 + row_half_mirror(r23): the four banks
//      hold four partial sums each of (a, c, b, d)                                  (2 + 2 + 1 = 5)
//   C  quad_perm [1,0,3,2] with a lane-parity select folds s0 | s1 into one register, s2 alone  (4)
//   D  quad_perm [2,3,0,1] with a bit-1 select folds those two                                  (3)
// 21 VALU (round 2's first version: 28, all stages with selects), no LDS.  The mirrors make the
// result independent of the rotation direction convention.  Lane li of each row ends with the total
// of value reduce9_component(li) (or -1: nothing).  The masked adds are inline assembly (the compiler
// cannot express a partial DPP write), each group behind an s_nop 1 for the VALU-write -> DPP-read
// hazard the assembler does not see.
__device__ __forceinline__ int reduce9_component(int li) {
    const int j = li & 3, b = li >> 2;
    const int bank_value = ((b & 1) << 1) | (b >> 1);   // banks hold values 0, 2, 1, 3 (+4 for s1)
    return j == 0 ? bank_value : (j == 1 ? 4 + bank_value : (li == 2 ? 8 : -1));
}
// Stages A and B alone: they sum over the ORBIT of a lane under {row_mirror, row_half_mirror} — the four lanes
// {l, 7 - l, 8 + l, 15 - l}, l = 0..3, one in each bank of the row.  Afterwards the lane in bank b holds the
// orbit's total of value {0, 2, 1, 3}[b] in s0, of value 4 + {0, 2, 1, 3}[b] in s1 and of value 8 in s2: a
// complete nine-value reduction over FOUR-lane groups in 14 VALU — sixteen groups per wave (backward_wave_q).
__device__ __forceinline__ void orbit_reduce9(float v0, float v1, float v2, float v3, float v4, float v5,
                                              float v6, float v7, float v8, float &s0, float &s1,
                                              float &s2) {
    float r01 = v0 + dpp_f<0x140>(v0);  // row_mirror
    float r23 = v2 + dpp_f<0x140>(v2);
    float r45 = v4 + dpp_f<0x140>(v4);
    float r67 = v6 + dpp_f<0x140>(v6);
    const float r8 = v8 + dpp_f<0x140>(v8);
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %4, %4 row_mirror row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %1, %5, %5 row_mirror row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %2, %6, %6 row_mirror row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %3, %7, %7 row_mirror row_mask:0xf bank_mask:0xc"
                 : "+v"(r01), "+v"(r23), "+v"(r45), "+v"(r67)
                 : "v"(v1), "v"(v3), "v"(v5), "v"(v7));
    s0 = r01 + dpp_f<0x141>(r01);  // row_half_mirror
    s1 = r45 + dpp_f<0x141>(r45);
    s2 = r8 + dpp_f<0x141>(r8);
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                 "v_add_f32_dpp %1, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa"
                 : "+v"(s0), "+v"(s1)
                 : "v"(r23), "v"(r67));
}
__device__ __forceinline__ float row_reduce9(float v0, float v1, float v2, float v3, float v4,
                                             float v5, float v6, float v7, float v8, bool odd,
                                             bool bit1) {
    // A: fold the row's halves
    float r01 = v0 + dpp_f<0x140>(v0);  // row_mirror
    float r23 = v2 + dpp_f<0x140>(v2);
    float r45 = v4 + dpp_f<0x140>(v4);
    float r67 = v6 + dpp_f<0x140>(v6);
    const float r8 = v8 + dpp_f<0x140>(v8);
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %4, %4 row_mirror row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %1, %5, %5 row_mirror row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %2, %6, %6 row_mirror row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %3, %7, %7 row_mirror row_mask:0xf bank_mask:0xc"
                 : "+v"(r01), "+v"(r23), "+v"(r45), "+v"(r67)
                 : "v"(v1), "v"(v3), "v"(v5), "v"(v7));
    // B: fold the halves' quads
    float s0 = r01 + dpp_f<0x141>(r01);  // row_half_mirror
    float s1 = r45 + dpp_f<0x141>(r45);
    const float s2 = r8 + dpp_f<0x141>(r8);
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                 "v_add_f32_dpp %1, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa"
                 : "+v"(s0), "+v"(s1)
                 : "v"(r23), "v"(r67));
    // C, D: inside the quads
    const float t = (odd ? s1 : s0) + dpp_f<0xB1>(odd ? s0 : s1);  // quad_perm [1,0,3,2]
    const float u = s2 + dpp_f<0xB1>(s2);
    return (bit1 ? u : t) + dpp_f<0x4E>(bit1 ? t : u);            // quad_perm [2,3,0,1]
}

// The same nine sums on the MATRIX pipe (GS_BWD_MFMA): v_mfma_f32_16x16x4_f32 computes
// D[i][j] += sum_k A[i][k] B[k][j] with A taken from lane (i = l & 15, k = l >> 4) and B from lane
// (k = l >> 4, j = l & 15); D[4 (l >> 4) + r][l & 15] lands in register r of lane l.  With A = one of
// the nine per-lane values and B = the one-hot column of its component, nine accumulating
// instructions leave in D[i][c] the sum of component c over the four lanes {i, i + 16, i + 32, i + 48},
// and adding a lane's four result registers sums the rows 4 g .. 4 g + 3: lane (g = l >> 4, c = l & 15)
// ends with the total of component c over the SIXTEEN lanes {4 g + q + 16 k : q, k < 4}.  So a "group" of
// this variant is not a DPP row but one quad out of each row (mfma_group / mfma_lane_in_group), and the
// reduction costs three VALU additions instead of twenty-one DPP operations; the matrix pipe is
// otherwise idle in this kernel and runs beside the other waves' VALU work (exact fp32: each product
// is a value times one or zero, the four-term sums are fmaf chains).  A non-finite partial sum would
// spread to the other components of its (group, entry) through 0 * inf — the DPP butterfly keeps it in
// its own component; both are garbage-in cases (finite inputs give finite sums).
#ifndef GS_BWD_MFMA_CHAINS
#define GS_BWD_MFMA_CHAINS 1
#endif
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int mfma_group(int lane) { return (lane >> 2) & 3; }
__device__ __forceinline__ int mfma_lane_in_group(int lane) { return (lane & 3) | ((lane >> 4) << 2); }
__device__ __forceinline__ float mfma_reduce9(float v0, float v1, float v2, float v3, float v4,
                                              float v5, float v6, float v7, float v8,
                                              const float (&oh)[9]) {
#if GS_BWD_MFMA_CHAINS == 2
    // two accumulators (even / odd components: disjoint columns): half the dependent-latency chain
    // (nine x 40 cycles), four more additions
    f32x4_t da = {0.0f, 0.0f, 0.0f, 0.0f}, db = {0.0f, 0.0f, 0.0f, 0.0f};
    da = __builtin_amdgcn_mfma_f32_16x16x4f32(v0, oh[0], da, 0, 0, 0);
    db = __builtin_amdgcn_mfma_f32_16x16x4f32(v1, oh[1], db, 0, 0, 0);
    da = __builtin_amdgcn_mfma_f32_16x16x4f32(v2, oh[2], da, 0, 0, 0);
    db = __builtin_amdgcn_mfma_f32_16x16x4f32(v3, oh[3], db, 0, 0, 0);
    da = __builtin_amdgcn_mfma_f32_16x16x4f32(v4, oh[4], da, 0, 0, 0);
    db = __builtin_amdgcn_mfma_f32_16x16x4f32(v5, oh[5], db, 0, 0, 0);
    da = __builtin_amdgcn_mfma_f32_16x16x4f32(v6, oh[6], da, 0, 0, 0);
    db = __builtin_amdgcn_mfma_f32_16x16x4f32(v7, oh[7], db, 0, 0, 0);
    da = __builtin_amdgcn_mfma_f32_16x16x4f32(v8, oh[8], da, 0, 0, 0);
    return ((da[0] + db[0]) + (da[1] + db[1])) + ((da[2] + db[2]) + (da[3] + db[3]));
#endif
    f32x4_t d = {0.0f, 0.0f, 0.0f, 0.0f};
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(v0, oh[0], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(v1, oh[1], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(v2, oh[2], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(v3, oh[3], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(v4, oh[4], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(v5, oh[5], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(v6, oh[6], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(v7, oh[7], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(v8, oh[8], d, 0, 0, 0);
    return (d[0] + d[1]) + (d[2] + d[3]);
}

// ---------------------------------------------------------------------------------------------
// Backward.  LDS per wave: staged records 3.1 KB, ids 256 B, per-entry accumulators 9 x 65 floats:
// 5.7 KB.
constexpr int kAcc = 9;           // accumulator floats per staged entry
constexpr int kAccStride = kChunk + 1;  // component-major [9][65]: the nine components of an entry in nine banks
constexpr float kFixScale = 1099511627776.0f;  // 2^40: fixed-point scale of GS_FLAG_DETERMINISTIC
// 1: the four groups of backward_wave read their next slot from per-group queues in LDS (behind the accumulators)
// instead of walking four SGPR masks — like the forward (GS_FWD_QWALK): on the frames these kernels still serve
// (few tiles, pieces, outlying lists) a wave is alone on its SIMD and every instruction of a step counts
#ifndef GS_BWD_QWALK
#define GS_BWD_QWALK 1
#endif
constexpr int kAccFloats = kAcc * kAccStride + (GS_BWD_QWALK ? (4 * kChunk + 16) / 4 : 0);   // + the queues

// (five waves per SIMD: the compiler keeps the rare exact-exponential / rectangle paths out of the
// register budget)
#ifndef GS_BWD_WAVES
#define GS_BWD_WAVES 5
#endif
// 1: the next chunk's records are gathered into registers while the current chunk is walked (twelve
// more live registers; 0 measured faster with the mask-driven staging: 345 -> 336 us at C2)
#ifndef GS_BWD_PREFETCH
#define GS_BWD_PREFETCH 0
#endif
// 1: the per-step nine-value reduction runs on the matrix pipe (mfma_reduce9) and a group is one quad of
// each DPP row; 0: the DPP butterfly (row_reduce9), a group is a DPP row.
#ifndef GS_BWD_MFMA
#define GS_BWD_MFMA 0
#endif
// One wave of the backward: the pixels [wx0, wx0 + WW) x [wy0, wy0 + WH) of `tile` (WaveGeom<PX>).  The
// LDS arrays belong to the calling kernel (one wave per workgroup).
// A piece [lo, hi) of a tile's list for the segmented backward (SEG): front = the forward's checkpoint record
// of the state in front of entry hi (null: hi is the end of the list), final_ = record 0 (the final state).
struct ListPiece {
    int lo, hi;
    const float4 *front, *final_;
};

template <bool EXACT, bool DET, int PX, bool SEG = false>
__device__ __forceinline__ void
backward_wave(int tile, int wx0, int wy0, SRecB *__restrict__ stage, int *__restrict__ sid,
              float *__restrict__ acc, int W, int H, const int32_t *__restrict__ ids,
              const uint16_t *__restrict__ masks, const int2 *__restrict__ bins,
              const float4 *__restrict__ packed, float bg0, float bg1, float bg2,
              const float *__restrict__ bg_dev, const float *__restrict__ final_Ts,
              const int32_t *__restrict__ final_idx, const float *__restrict__ v_out,
              const float *__restrict__ v_out_alpha, const float *__restrict__ img_raw,
              float *__restrict__ gacc, unsigned long long *__restrict__ gfix,
              const ListPiece piece = ListPiece{0, 0, nullptr, nullptr}, int final_first = -1) {
    using G = WaveGeom<PX>;
    const int lane = threadIdx.x;
    if (bg_dev) {
        bg0 = bg_dev[0]; bg1 = bg_dev[1]; bg2 = bg_dev[2];
    }
#if GS_BWD_MFMA
    const int grp = mfma_group(lane), li = mfma_lane_in_group(lane);
    // which of the nine sums mfma_reduce9 leaves in this lane, and for which group (rgsh: that group's
    // byte of the packed slot word) — the lane's own pixels belong to group `grp`
    const int rcomp = (lane & 15) < kAcc ? (lane & 15) : -1;
    const uint32_t rgsh = 8u * (uint32_t)(lane >> 4);
    float onehot[kAcc];
#pragma unroll
    for (int c = 0; c < kAcc; c++) {
        onehot[c] = (lane & 15) == c ? 1.0f : 0.0f;
        asm volatile("" : "+v"(onehot[c]));   // nine live registers, not nine compares per step
    }
#else
    const int grp = lane >> 4, li = lane & 15;
    const bool odd = (lane & 1) != 0, bit1 = (lane & 2) != 0;
    const int rcomp = reduce9_component(li);   // which of the nine sums row_reduce9 leaves in this lane
#endif
    const int fj = (lane * 7282) >> 16;        // flush: lane = 9 * fj + fcomp (lane 63: fj = 7, no work)
    const int fcomp = lane == 63 ? kAcc : lane - 9 * fj;
    const uint32_t gsh = 8u * (uint32_t)grp;   // (scalar walk only)
    (void)gsh;
    const int px = wx0 + G::BW * (grp & 1) + (li % G::LW);
    const int py0 = wy0 + G::BH * (grp >> 1) + (li / G::LW);   // pixel p of the lane: row py0 + p * LH
    const float pxf = (float)px;
    // per pixel: row, transmittance being unwound, D = T_final * (v_out_alpha - bg . v_out) minus the
    // running <colour buffer, v_out> (one accumulator: the difference is what v_alpha needs),
    // cotangent, list index of the last contributor
    float pyf[PX], T[PX], D[PX], vo0[PX], vo1[PX], vo2[PX];
    int last[PX];   // list index of the pixel's last contributor; inside the chunk loop: RELATIVE to the
                    // chunk (slot t has list index hi - t: it is needed iff t >= hi - last)
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
        D[p] = Tfin * (oa - (bg0 * vo0[p] + bg1 * vo1[p] + bg2 * vo2[p]));
        if (SEG) {
            // the pixel's part of this piece of the list: nothing (its last contributor lies in front of
            // the piece), from its last contributor down (which lies inside), or — the list goes on behind
            // the piece — from the piece's last entry with the state the forward left in front of entry hi:
            // that transmittance, and the colour composited behind it (final - front) taken off D
            if (last[p] < piece.lo) {
                last[p] = -1;
            } else if (last[p] >= piece.hi) {
                const int pid = ((py & (GS_TILE - 1)) << 4) | (px & (GS_TILE - 1));
                // (records in the backward's terms, k_rasterize_forward: {T invG, S} and {invG(end), S(end)})
                const float4 cf = piece.front[pid], ce = piece.final_[pid];
                const float Gt = __builtin_amdgcn_rcpf(ce.x);   // (exactly 1 without a hot entry)
                T[p] = cf.x * Gt;
                D[p] = D[p] - Gt * ((ce.y - cf.y) * vo0[p] + (ce.z - cf.z) * vo1[p] + (ce.w - cf.w) * vo2[p]);
                last[p] = piece.hi - 1;
            }
        }
        gl = max(gl, last[p]);
    }
    // last contributor of each block (one group of lanes) and of the wave
    gl = max(gl, dpp_i<0xB1>(gl));
    gl = max(gl, dpp_i<0x4E>(gl));
#if GS_BWD_MFMA
    // a group = quad g of each of the four DPP rows
#define GS_GL(g)                                                                                        \
    max(max(__builtin_amdgcn_readlane(gl, 4 * (g)), __builtin_amdgcn_readlane(gl, 4 * (g) + 16)),      \
        max(__builtin_amdgcn_readlane(gl, 4 * (g) + 32), __builtin_amdgcn_readlane(gl, 4 * (g) + 48)))
    const int gl0 = GS_GL(0), gl1 = GS_GL(1), gl2 = GS_GL(2), gl3 = GS_GL(3);
#undef GS_GL
#else
    gl = max(gl, dpp_i<0x141>(gl));
    gl = max(gl, dpp_i<0x140>(gl));
    const int gl0 = __builtin_amdgcn_readlane(gl, 0), gl1 = __builtin_amdgcn_readlane(gl, 16);
    const int gl2 = __builtin_amdgcn_readlane(gl, 32), gl3 = __builtin_amdgcn_readlane(gl, 48);
#endif
    const int wave_last = max(max(gl0, gl1), max(gl2, gl3));
    int2 range = bins[tile];
    if (SEG) range.x = piece.lo;      // (a piece ends where the walk does: at its first entry)
    if (wave_last < range.x) return;  // (also covers empty tiles / no contributors)

    if (lane == 0) stage_sentinel(&stage[kChunk]);
#pragma unroll
    for (int i = 0; i < kAcc; i++) acc[i * kAccStride + lane] = 0.0f;

    // walk the list back to front in chunks; slot 0 of a chunk is its furthest-back entry.  Only
    // entries whose coverage mask touches one of the wave's blocks are gathered and staged.
    const int ox = wx0 & (GS_TILE - 1), oy = wy0 & (GS_TILE - 1);
#if GS_BWD_PREFETCH
    float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0;
#endif
    int ng = 0;
    uint32_t ntouch = 0u;
    if (wave_last - lane >= range.x) {
        ng = ids[wave_last - lane];
        ntouch = touch_from_mask<PX>(masks[wave_last - lane], ox, oy);
#if GS_BWD_PREFETCH
        if (ntouch) {
            n0 = packed[3 * (size_t)ng + 0]; n1 = packed[3 * (size_t)ng + 1]; n2 = packed[3 * (size_t)ng + 2];
        }
#endif
    }
#pragma unroll
    for (int p = 0; p < PX; p++) last[p] = wave_last - last[p] + kChunk;   // (no contributor: beyond any slot)
    for (int hi = wave_last; hi >= range.x; hi -= kChunk) {
#pragma unroll
        for (int p = 0; p < PX; p++) last[p] -= kChunk;   // now relative to this chunk's first slot
        wave_prio(final_first, hi - range.x + 1, wave_last - range.x + 1);
        __syncthreads();
        const uint32_t touch = ntouch;
        bool binds_t = false;   // this lane's entry needs the per-pixel rectangle test
        if (touch) {
#if !GS_BWD_PREFETCH
            // the record is gathered here, at the head of its chunk (only id and mask run one chunk
            // ahead): the other waves of the SIMD cover the latency, and twelve registers are free
            const float4 n0 = packed[3 * (size_t)ng + 0], n1 = packed[3 * (size_t)ng + 1],
                         n2 = packed[3 * (size_t)ng + 2];
#endif
            // (an opacity <= 0 never reaches 1/255: threshold +inf)
            const float tv = n1.y > 0.0f ? (1.0f / 255.0f) / n1.y : __builtin_inff();
#if GS_BWD_LOG2E
            const uint32_t sb = __float_as_uint(n1.z);
#if GS_BWD_SIGMA_THRESH
            // L' -+ band; an opacity that cannot reach 1/255 at all gets the empty interval [0, 0]
            float s_hi = 0.0f, s_lo = -1.0f;
            if (n1.y > 0.0f) {
                const float Lp = __builtin_amdgcn_logf(255.0f * n1.y);   // v_log_f32: log2
                if (Lp + kSigBand >= 0.0f) { s_hi = Lp + kSigBand; s_lo = Lp - kSigBand; }
            }
            const float sm = __uint_as_float((__float_as_uint(s_hi) & ~1u) | (sb & 1u));
            const float p1w = s_lo;
#else
            // (sigma_max keeps its flag bit; its +2e-3 of slack scales along)
            const float sm = __uint_as_float((__float_as_uint(__uint_as_float(sb & ~1u) * kLog2e) & ~1u) | (sb & 1u));
            const float p1w = tv * (1.0f + kVisBand);
#endif
            stage[lane].p0 = make_float4(n0.x, n0.y, n0.z * kLog2e, n0.w * kLog2e);
            stage[lane].p1 = make_float4(n1.x * kLog2e, n1.y, sm, p1w);
#else
            stage[lane].p0 = n0;
            stage[lane].p1 = make_float4(n1.x, n1.y, n1.z, tv * (1.0f + kVisBand));
#endif
            stage[lane].p2 = make_float4(n2.x, n2.y, n2.z, tv * (1.0f - kVisBand));
            stage[lane].p3 = make_float4(n1.w, n2.w, n0.z, n0.w);
            stage[lane].p4 = make_float4(n1.x, 0.0f, 0.0f, 0.0f);
            binds_t = (__float_as_uint(n1.z) & 1u) != 0u;
        }
        if (hi - lane >= range.x) sid[lane] = ng;
        const bool chunk_binds = __builtin_amdgcn_ballot_w64(binds_t) != 0ull;
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
#if GS_BWD_QWALK && !GS_BWD_MFMA
        uint8_t *fq = reinterpret_cast<uint8_t *>(acc + kAcc * kAccStride);   // [group][rank] -> slot
        reinterpret_cast<uint32_t *>(fq)[lane] = kChunk * 0x01010101u;
        if (lane < 4) reinterpret_cast<uint32_t *>(fq)[kChunk + lane] = kChunk * 0x01010101u;
#define GS_FQ(g, m)                                                                                         \
    if (__builtin_amdgcn_inverse_ballot_w64(m))                                                             \
        fq[(g) * kChunk + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)((m) >> 32),                             \
                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)(m), 0u))] =   \
            (uint8_t)lane;
        GS_FQ(0, m0) GS_FQ(1, m1) GS_FQ(2, m2) GS_FQ(3, m3)
#undef GS_FQ
        const int nsteps = max(max(__builtin_popcountll(m0), __builtin_popcountll(m1)),
                               max(__builtin_popcountll(m2), __builtin_popcountll(m3)));
#endif
        __syncthreads();
        GS_STAT(11, __builtin_popcountll(m0) + __builtin_popcountll(m1) + __builtin_popcountll(m2) + __builtin_popcountll(m3));
        GS_STAT(12, 1);
        ntouch = 0u;
        if (hi - kChunk - lane >= range.x) {
            ng = ids[hi - kChunk - lane];
            ntouch = touch_from_mask<PX>(masks[hi - kChunk - lane], ox, oy);
#if GS_BWD_PREFETCH
            if (ntouch) {
                n0 = packed[3 * (size_t)ng + 0]; n1 = packed[3 * (size_t)ng + 1]; n2 = packed[3 * (size_t)ng + 2];
            }
#endif
        }
        bool flushed_any = false;  // wave-uniform
        auto walk = [&](auto binds_tag) {
            constexpr bool BINDS = decltype(binds_tag)::value;
#if GS_BWD_QWALK && !GS_BWD_MFMA
            const uint8_t *myq = &fq[grp * kChunk];
            int e_next = myq[0];
            for (int k = 0; k < nsteps; k++) {
                const int e = e_next;
                const float4 q0 = stage[e].p0, q1 = stage[e].p1, q2 = stage[e].p2;
                e_next = myq[k + 1];
#else
            uint32_t ep_next;
            { GS_WALK_PACK(ep0_) ep_next = ep0_; }
            while (ep_next != kWalkDone) {
                const uint32_t ep = ep_next;
                const int e = (int)((ep >> gsh) & 0xFFu);
                const float4 q0 = stage[e].p0, q1 = stage[e].p1, q2 = stage[e].p2;
                { GS_WALK_PACK(ep1_) ep_next = ep1_; asm volatile("" : "+s"(ep_next)); }
#endif
                const uint32_t sbits = __float_as_uint(q1.z);
                GS_STAT(8, 1);
                const float dx = q0.x - pxf;
                const float Adxdx = (q0.z * dx) * dx, Bdx = q0.w * dx;
                const float hAdxdx = 0.5f * Adxdx, hC = 0.5f * q1.x;
                // rectangle test data of the rare entries whose rectangle cuts the sigma_max ellipse
                const bool any_binds = BINDS && __builtin_amdgcn_ballot_w64((sbits & 1u) != 0u) != 0ull;
                // (-0 is the identity of the addition: the first live pass adds to nothing)
                float su = -0.0f, suy = -0.0f, suyy = -0.0f, gr = -0.0f, gg = -0.0f, gb = -0.0f;
                uint64_t anym = 0ull;  // lanes that needed the entry in some pass
    #pragma unroll
                for (int p = 0; p < PX; p++) {
                    const float dy = q0.y - pyf[p];
                    // 0.5 (A dx^2 + C dy^2) + B dx dy with the halving folded into the per-step factors
                    // (a power of two commutes with every rounding: the same bits as halving the sum)
                    float sg = fmaf(hC * dy, dy, hAdxdx);
                    sg = fmaf(Bdx, dy, sg);
                    if (any_binds) {
                        asm volatile("; rectangle binds");
                        if (sbits & 1u) {
                            // decide exactly like the forward: its op order for sigma, rectangle applied
                            const float4 q3 = stage[e].p3;
                            float se = ((q3.z * dx) * dx) + (stage[e].p4.x * dy) * dy;
                            se = 0.5f * se;
                            se = se + (q3.w * dx) * dy;
                            const uint32_t rx = __float_as_uint(q3.x), ry = __float_as_uint(q3.y);
                            const uint32_t pyu = (uint32_t)(py0 + p * G::LH);
                            const bool in = (uint32_t)px >= (rx & 0xFFFFu) && (uint32_t)px < (rx >> 16) &&
                                            pyu >= (ry & 0xFFFFu) && pyu < (ry >> 16);
                            sg = in ? (GS_BWD_LOG2E ? (se + 0.0f) * kLog2e : se + 0.0f) : qnan();
                        }
                    }
                    // lane masks are combined as 64-bit scalars (ballot of each compare, s_and) and turned
                    // back into a lane predicate with inverse_ballot: a ballot of `a && b` costs a
                    // v_cndmask + v_cmp pair per use
                    const uint64_t mneed = __builtin_amdgcn_ballot_w64(e >= last[p]) &
                                           __builtin_amdgcn_ballot_w64(__float_as_uint(sg) <= sbits);
                    if (mneed == 0ull) continue;
                    anym |= mneed;
                    GS_STAT(9, 1);
                    GS_STAT(10, __builtin_popcountll(mneed));
                    // vis = exp(-sigma), alpha = min(0.99, opacity * vis), gsplat_cpu.cpp:337-338.  Lanes
                    // that do not need the entry keep whatever the exponential makes of their sigma
                    // (possibly inf or NaN) until `ok` discards it: they end with vis = alpha = 0
#if GS_BWD_LOG2E
                    float vis = __builtin_amdgcn_exp2f(-sg);   // sg = sigma log2(e)
#else
                    float vis = __expf(-sg);
#endif
                    uint64_t mok;
                    if (EXACT) {
                        // the forward's >= 1/255 decision, taken against the entry's thresholds (SRecB) — on
                        // sigma' (GS_BWD_SIGMA_THRESH: mneed already holds "possibly", q1.w is "certainly")
                        // or on vis; inside the band the forward's own arithmetic is redone — its sigma,
                        // the exact exponential, alpha = o * vis — and decides
#if GS_BWD_SIGMA_THRESH
                        const uint64_t bhi = __builtin_amdgcn_ballot_w64(sg <= q1.w);
                        const uint64_t blo = ~0ull;
#else
                        const uint64_t bhi = __builtin_amdgcn_ballot_w64(vis >= q1.w);
                        const uint64_t blo = __builtin_amdgcn_ballot_w64(vis >= q2.w);
#endif
                        mok = mneed & bhi;
                        if (GS_BWD_SIGMA_THRESH ? mok != mneed : bhi != blo) {   // some lane sits in the band
                            asm volatile("; threshold ambiguous");
                            bool in = false;
                            if (__builtin_amdgcn_inverse_ballot_w64(mneed & blo & ~bhi)) {
                                const float4 q3 = stage[e].p3;
                                float se = ((q3.z * dx) * dx) + (stage[e].p4.x * dy) * dy;
                                se = 0.5f * se;
                                se = se + (q3.w * dx) * dy;
                                vis = expf_glibc_cmem(-se);
                                in = q1.y * vis >= (1.0f / 255.0f);
                            }
                            mok |= __builtin_amdgcn_ballot_w64(in);
                        }
                    } else {
                        mok = mneed & __builtin_amdgcn_ballot_w64(q1.y * vis >= (1.0f / 255.0f));
                    }
                    vis = __builtin_amdgcn_inverse_ballot_w64(mok) ? vis : 0.0f;
                    const float alpha = fminf(q1.y * vis, 0.99f);   // 0 where the entry does not contribute
                    // ra = 1 / (1 - alpha) by the hardware reciprocal (1 ulp; exactly 1 for alpha = 0): T
                    // is a running product either way — a correctly rounded quotient would make its
                    // error 1 instead of 1.5 ulp per entry
                    const float om = 1.0f - alpha;
                    const float ra = __builtin_amdgcn_rcpf(om);
                    T[p] = T[p] * ra;  // transmittance in front of this Gaussian
                    const float fac = alpha * T[p];
                    gr = fmaf(fac, vo0[p], gr);
                    gg = fmaf(fac, vo1[p], gg);
                    gb = fmaf(fac, vo2[p], gb);
                    // cv = <colour, v_out>;  v_alpha = T*cv + ra*(T_final*w - <buffer, v_out>) = T*cv + ra*D
                    const float cv = fmaf(q2.z, vo2[p], fmaf(q2.y, vo1[p], q2.x * vo0[p]));
                    const float v_alpha = fmaf(T[p], cv, ra * D[p]);
                    D[p] = fmaf(-fac, cv, D[p]);
                    // u = vis * v_alpha (= d/d opacity); v_sigma = -opacity * u is applied at the flush
                    const float u = vis * v_alpha;
                    const float uy = u * dy;
                    su += u;
                    suy += uy;
                    suyy = fmaf(uy, dy, suyy);
                }
                if (anym == 0ull) continue;
                // ---- the nine sums over the group's 16 lanes: lane c of the row ends up with total c ----
                const float ux = su * dx;
#if GS_BWD_MFMA
                const float r = mfma_reduce9(ux, suy, ux * dx, suy * dx, suyy, gr, gg, gb, su, onehot);
                const int er = (int)((ep >> rgsh) & 0xFFu);   // the entry of the group this lane reports
                if (rcomp >= 0 && r != 0.0f && er < kChunk)  // (a group without work has nothing to add)
                    __hip_atomic_fetch_add(&acc[rcomp * kAccStride + er], r, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
#else
                const float r = row_reduce9(ux, suy, ux * dx, suy * dx, suyy, gr, gg, gb, su, odd, bit1);
                if (rcomp >= 0 && r != 0.0f && e < kChunk)  // (a group without work has nothing to add)
                    __hip_atomic_fetch_add(&acc[rcomp * kAccStride + e], r, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
                flushed_any = true;
            }
        };
        if (chunk_binds) walk(std::true_type{}); else walk(std::false_type{});
        if (!flushed_any) continue;
        // ---- flush: moments -> gradient components (once per entry), then one atomic lane per
        //      (entry, component): the nine lanes of an entry hit ONE 64-byte record ----
        wave_sync();
        // (only slots staged THIS chunk: an unstaged slot has zero sums, and the stale record it may still
        // hold — possibly one with a non-finite conic — must not be multiplied into them: 0 * inf = NaN
        // would be flushed into whichever Gaussian the slot's id names now)
        if (touch != 0u) {
            const float Ux = acc[0 * kAccStride + lane], Uy = acc[1 * kAccStride + lane];
            const float Uxx = acc[2 * kAccStride + lane], Uxy = acc[3 * kAccStride + lane];
            const float Uyy = acc[4 * kAccStride + lane];
            const float mo = -stage[lane].p1.y;          // v_sigma = -opacity * u
            const float A = stage[lane].p3.z, B = stage[lane].p3.w, C = stage[lane].p4.x;
            acc[0 * kAccStride + lane] = mo * fmaf(A, Ux, B * Uy);   // v_x: v_sigma * (A dx + B dy)
            acc[1 * kAccStride + lane] = mo * fmaf(B, Ux, C * Uy);   // v_y: v_sigma * (B dx + C dy)
            acc[2 * kAccStride + lane] = 0.5f * mo * Uxx;            // v_A  (gsplat_cpu.cpp:361-363)
            acc[3 * kAccStride + lane] = 0.5f * mo * Uxy;            // v_B
            acc[4 * kAccStride + lane] = 0.5f * mo * Uyy;            // v_C
        }
        wave_sync();
        // record-major: lanes 9 j .. 9 j + 8 carry the nine components of entry 7 i + j (j < 7, lane
        // 63 idle), so the nine lanes of an entry hit ONE 64-byte record in one instruction and the
        // only per-lane constants are (j, component): the entry of iteration i is an immediate offset
        // away.  (Enumerating k = 64 i + lane -> (k / 9, k % 9) instead needs 18 loop-invariant
        // addresses, which the compiler hoists and -- at 96 VGPRs -- spills: +235 MB of scratch
        // traffic per launch, profiles/r02_pmc.json.)
#pragma unroll
        for (int i = 0; i < (kChunk + 6) / 7; i++) {
            const int ent = 7 * i + fj;
            if (fcomp < kAcc && (7 * i + 6 < kChunk || ent < kChunk)) {
                const float v = acc[fcomp * kAccStride + ent];
                if (v != 0.0f) {
                    const size_t o = (size_t)sid[ent] * kGradRec + fcomp;
                    if (DET)
                        // round to nearest (a truncating convert biases every sum towards zero) and
                        // saturate at +-2^62 (a float beyond int64 is undefined in the convert): the
                        // quantum is 2^-40 = 9.1e-13 per flushed partial sum, |sums| up to 2^22
                        atomicAdd(gfix + o, (unsigned long long)__float2ll_rn(
                                                fminf(fmaxf(v * kFixScale, -4.6e18f), 4.6e18f)));
                    else
                        atomicAdd(gacc + o, v);
                }
            }
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < kAcc; i++) acc[i * kAccStride + lane] = 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// Round 5: SIXTEEN GROUPS OF FOUR LANES ("Q geometry").  A wave is still one tile with four pixels per
// lane, but a group is FOUR lanes that own one 4 x 4-pixel block (a lane = one column of the block, its four
// pixels the block's rows), and every one of the sixteen groups walks ITS OWN list — the entries of the staged
// chunk whose coverage-mask bit for that block is set: sixteen different Gaussians per instruction stream.
//   * why: with 8 x 8 blocks a Gaussian covers 16 - 17 of the 64 pixel slots of a pass (26 % live lanes,
//     profiles/work_stats_r04_c2.json); a 4 x 4 block it reaches it covers to 52 %, and all four passes of a
//     step are always needed.  scripts/sim_bwd_geometry.py (the same lists rebuilt on the host, checked against
//     the instrumented kernel's counters): 0.30 steps per list entry at 71 % group fill against 0.47 steps at
//     3.65 passes — 0.68 x the VALU instructions; measured 1.53e8 -> 1.04e8 (profiles/r05/).
//   * the walk cannot be scalar any more (sixteen s_ff1 chains per step): per chunk the wave builds sixteen
//     QUEUES of slot numbers in LDS — one ballot per block bit, trimmed by the block's last contributor, ranks by
//     v_mbcnt, a byte store per (block, entry) pair — and a step is one ds_read_u8 with a per-group address;
//     the queues are pre-filled with the sentinel slot, so an exhausted group reads the NaN record by itself.
//     The step count of a chunk is the longest queue (scalar: sixteen s_bcnt1 + s_max).
//   * the nine sums of a step are reduced over the group's four lanes by the first two stages of the row
//     butterfly alone (orbit_reduce9, 14 VALU: the groups are the orbits {l, 7-l, 8+l, 15-l} of row_mirror and
//     row_half_mirror, one lane in each DPP bank).
//   * NO LDS ATOMICS.  The first version added the group's sums to per-entry accumulators with ds_add_f32 like
//     backward_wave does: 144 atomic lanes per step, and an LDS float atomic costs ~2.2 LDS cycles PER LANE
//     (SQ_LDS_IDX_ACTIVE 1.2e8 -> 2.3e8, SQ_WAIT_INST_LDS 2.2e7 -> 4.5e8: 442 us against 294, LDS-bound,
//     profiles/r05/q64_ldsatomic_*).  The second stored every (block, entry) pair's sums in a record of its own
//     and let lane t add up entry t's records after the walk: no LDS wait any more, but 600 VALU per chunk of
//     queue / gather bookkeeping and 13 KB of LDS (three waves per SIMD): 298 us (profiles/r05/q192_pairstore_*).
//     Now: PLAIN read-add-write on the entry's accumulator record, made safe by a CLAIM — two groups collide
//     only if they take the same entry in the same step; every group writes its number to tag[entry] (one byte
//     store), reads it back, and the group whose number stuck does its read-add-write; the others go round
//     again (LDS operations of a wave execute in order, the lanes in lockstep: exactly one winner per entry
//     and round).  One round almost always.
// Everything per pixel — thresholds, the exact re-evaluation inside the band, the recurrences — is
// backward_wave's, statement for statement; flush and gradient records are unchanged.
struct __attribute__((aligned(16))) SRecQ {
    float4 p0, p1, p2;   // {x y A' B' | C' o s_hi s_lo | r g b C}   (A' = A log2 e ...; C unscaled for the rare paths)
};
constexpr int kQChunk = 64;
// floats per accumulator record: {c0 c4 | c1 c5 | c2 c6 | c3 c7 | c8 -}.  (Round 6: 48-byte records — the stride of
// the staged records, the claim word inside: one multiply per step addresses all three, four VALU less — measured
// SLOWER, 0.2411 against 0.2342 ms at C2, 1.616 against 1.601 at C3: with a stride of twelve dwords the records of
// slots e and e + 8 share their banks, with ten dwords e and e + 16, and the claim's accesses are what collides.)
constexpr int kAccRec = 10;
// dword of component c in an accumulator record
__device__ __forceinline__ int acc_dword(int c) { return c < 8 ? 2 * (c & 3) + (c >> 2) : 8; }
// accumulator copies: 1, or 2 = one per checkerboard colour of the blocks — neighbouring blocks hold the same
// Gaussians at nearly the same queue ranks and meet in the claim; with a copy per colour the model
// (scripts/sim_bwd_geometry.py: claim_rounds) gives 1.89 rounds per step instead of 2.73, for 2.9 KB more LDS per wave
#ifndef GS_BWDQ_COPIES
#define GS_BWDQ_COPIES 1
#endif
constexpr int kQCopies = GS_BWDQ_COPIES;
// (Round 6: a wave of this kernel walks 2.6 .. 3.0 list entries per us whether two or four waves share its SIMD
// — one wave gets a quarter of a SIMD's issue slots at most — and a C2 launch spends its last 40 % with two waves per
// SIMD: scripts/wave_timeline.py, profiles/r06/wave_timeline_*.json.  A walk software-pipelined by one step — the
// next staged record read a step ahead, the sums of step k claimed and added under the passes of step k + 1, the four
// passes' threshold decisions behind ONE branch, 128 VGPRs — was built, bit-identical, and SLOWER: 0.262 against
// 0.247 ms at C2, 1.80 against 1.65 ms at C3; the oldest wave of a SIMD walked 2.99 instead of 3.04 entries per
// us: what a wave waits for is not its LDS round trips but its own issue slots, and the pipelined walk issues more
// instructions.  profiles/HISTORY.md.)
// the claim word.  (A byte-sized tag array kept QLds at 8128 bytes — TWENTY waves per CU instead of eighteen — and
// launch_bounds(64, 4) gave sixteen; measured in round 6, same box, three interleaved bench lines each: 0.2484 /
// 0.2476 and 0.2461 / 0.2453 against 0.2456 / 0.2478 ms at C2: the kernel does not react to 16 .. 20 waves per CU.)
typedef unsigned int qtag_t;
// GS_BWDQ_PK: the four pixel passes of a step as two PACKED ones (v_pk_add / mul / fma_f32 on the rows 2 j, 2 j + 1).
// A wave is bound by the instructions it issues (round 6, scripts/wave_timeline.py): 44 packed instructions replace
// 88 plain ones per step; a saturated SIMD pays ~1.6 plain ones for a packed one (profiles/valu_calib_r06.json).
#ifndef GS_BWDQ_PK
#define GS_BWDQ_PK 1
#endif
typedef float v2f __attribute__((ext_vector_type(2)));
// waves per SIMD the q kernels are compiled for: the packed passes want a few more registers than 96 (no gain or loss
// from 16 against 18 .. 20 resident waves per CU: measured, profiles/HISTORY.md)
#ifndef GS_BWDQ_WAVES
#define GS_BWDQ_WAVES (GS_BWDQ_PK ? 4 : GS_BWD_WAVES)
#endif
struct QLds {
    SRecQ stage[kQChunk + 1];
    float4 rare[kQChunk + 1];                          // {rx, ry, A, B}: rectangle words, unscaled conic (rare paths, flush)
    alignas(16) uint8_t queue[16 * kQChunk + 16];      // [block][rank] -> slot; (+16: the read one step ahead)
    alignas(16) float acc[kQCopies][(kQChunk + 1) * kAccRec];    // per-entry sums (entry-major)
    int sid[kQChunk];
    qtag_t tag[kQCopies][kQChunk + 1];                 // the claim: which block adds to an entry this round
};

template <bool EXACT, bool DET>
__device__ __forceinline__ void
backward_wave_q(int tile, int tx0, int ty0, QLds &lds, int W, int H, const int32_t *__restrict__ ids,
                const uint16_t *__restrict__ masks, const int2 *__restrict__ bins,
                const float4 *__restrict__ packed, float bg0, float bg1, float bg2,
                const float *__restrict__ bg_dev, const float *__restrict__ final_Ts,
                const int32_t *__restrict__ final_idx, const float *__restrict__ v_out,
                const float *__restrict__ v_out_alpha, const float *__restrict__ img_raw,
                float *__restrict__ gacc, unsigned long long *__restrict__ gfix, int final_first) {
    static_assert((GS_BWD_LOG2E & GS_BWD_SIGMA_THRESH) != 0, "the Q walk is written for the sigma' thresholds");
    constexpr int CH = kQChunk;
    constexpr int PX = 4;
    const int lane = threadIdx.x;
    if (bg_dev) {
        bg0 = bg_dev[0]; bg1 = bg_dev[1]; bg2 = bg_dev[2];
    }
    // lane -> (block, column): row of lanes = block row; the orbit index = block column; DPP bank = column
    const int brow = lane >> 4, l16 = lane & 15, bank = l16 >> 2;
    const int bcol = (bank & 1) ? 3 - (l16 & 3) : (l16 & 3);
    const int grp = 4 * brow + bcol;                    // == the block's bit in the coverage mask
    const int c0 = ((bank & 1) << 1) | (bank >> 1);     // the component orbit_reduce9 leaves in s0 (s1: 4 + c0)
    const int fj = (lane * 7282) >> 16;                 // flush: lane = 9 * fj + fcomp (lane 63: no work)
    const int fcomp = lane == 63 ? kAcc : lane - 9 * fj;
    const int fdw = acc_dword(fcomp < kAcc ? fcomp : 0);
    const int px = tx0 + 4 * bcol + bank;
    const int py0 = ty0 + 4 * brow;                     // pixel p of the lane: row py0 + p
    const float pxf = (float)px;
    float pyf[PX], T[PX], D[PX], vo0[PX], vo1[PX], vo2[PX];
    int last[PX];
    int gl = -1;
#pragma unroll
    for (int p = 0; p < PX; p++) {
        const int py = py0 + p;
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
            if (img_raw) {  // backward of the fused clamp_max(rgb, 1)
                if (!(img_raw[3 * pix + 0] <= 1.0f)) vo0[p] = 0.0f;
                if (!(img_raw[3 * pix + 1] <= 1.0f)) vo1[p] = 0.0f;
                if (!(img_raw[3 * pix + 2] <= 1.0f)) vo2[p] = 0.0f;
            }
            oa = v_out_alpha ? v_out_alpha[pix] : 0.0f;
        }
        pyf[p] = (float)py;
        T[p] = Tfin;
        D[p] = Tfin * (oa - (bg0 * vo0[p] + bg1 * vo1[p] + bg2 * vo2[p]));
        gl = max(gl, last[p]);
    }
    // last contributor of each block (its four lanes: the orbit) and of the wave
    gl = max(gl, dpp_i<0x140>(gl));
    gl = max(gl, dpp_i<0x141>(gl));
    int sgl[16];
#pragma unroll
    for (int g = 0; g < 16; g++) sgl[g] = __builtin_amdgcn_readlane(gl, 16 * (g >> 2) + (g & 3));
    int wave_last = sgl[0];
#pragma unroll
    for (int g = 1; g < 16; g++) wave_last = max(wave_last, sgl[g]);
    const int2 range = bins[tile];
    if (wave_last < range.x) return;

    if (lane == 0) {
        stage_sentinel(&lds.stage[CH]);
        lds.rare[CH] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
#if GS_BWDQ_PK
    // per-pixel state of rows 2 j, 2 j + 1 side by side (even-aligned register pairs: packed fp32 operands)
    static_assert(PX == 4, "two packed passes");
    v2f pyf2[2], T2[2], D2[2], vo0_2[2], vo1_2[2], vo2_2[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        pyf2[j] = v2f{pyf[2 * j], pyf[2 * j + 1]};
        T2[j] = v2f{T[2 * j], T[2 * j + 1]};
        D2[j] = v2f{D[2 * j], D[2 * j + 1]};
        vo0_2[j] = v2f{vo0[2 * j], vo0[2 * j + 1]};
        vo1_2[j] = v2f{vo1[2 * j], vo1[2 * j + 1]};
        vo2_2[j] = v2f{vo2[2 * j], vo2[2 * j + 1]};
    }
#endif
    const int copy = kQCopies > 1 ? ((brow + bcol) & 1) : 0;
    float2 *myrec = reinterpret_cast<float2 *>(&lds.acc[0][lane * kAccRec]);   // (8-byte aligned: 40-byte records)
    float2 *myrec1 = reinterpret_cast<float2 *>(&lds.acc[kQCopies - 1][lane * kAccRec]);
#pragma unroll
    for (int i = 0; i < kAccRec / 2; i++) {
        myrec[i] = make_float2(0.0f, 0.0f);
        if (kQCopies > 1) myrec1[i] = make_float2(0.0f, 0.0f);
    }

    int ng = 0;
    uint32_t nmask = 0u;
    if (wave_last - lane >= range.x) { ng = ids[wave_last - lane]; nmask = masks[wave_last - lane]; }
#pragma unroll
    for (int p = 0; p < PX; p++) last[p] = wave_last - last[p] + CH;   // (no contributor: beyond any slot)
    const uint8_t *myq = &lds.queue[grp * CH];
    for (int hi = wave_last; hi >= range.x; hi -= CH) {
#pragma unroll
        for (int p = 0; p < PX; p++) last[p] -= CH;   // now relative to this chunk's first slot
        wave_prio(final_first, hi - range.x + 1, wave_last - range.x + 1);
        __syncthreads();
        const uint32_t msk = nmask;
        bool binds_t = false;
        if (msk) {
            const float4 n0 = packed[3 * (size_t)ng + 0], n1 = packed[3 * (size_t)ng + 1],
                         n2 = packed[3 * (size_t)ng + 2];
            const uint32_t sb = __float_as_uint(n1.z);
            // L' -+ band; an opacity that cannot reach 1/255 at all gets the empty interval [0, 0]
            float s_hi = 0.0f, s_lo = -1.0f;
            if (n1.y > 0.0f) {
                const float Lp = __builtin_amdgcn_logf(255.0f * n1.y);   // v_log_f32: log2
                if (Lp + kSigBand >= 0.0f) { s_hi = Lp + kSigBand; s_lo = Lp - kSigBand; }
            }
            const float sm = __uint_as_float((__float_as_uint(s_hi) & ~1u) | (sb & 1u));
            lds.stage[lane].p0 = make_float4(n0.x, n0.y, n0.z * kLog2e, n0.w * kLog2e);
            lds.stage[lane].p1 = make_float4(n1.x * kLog2e, n1.y, sm, s_lo);
            lds.stage[lane].p2 = make_float4(n2.x, n2.y, n2.z, n1.x);
            lds.rare[lane] = make_float4(n1.w, n2.w, n0.z, n0.w);
            binds_t = (sb & 1u) != 0u;
        }
        if (hi - lane >= range.x) lds.sid[lane] = ng;
        const bool chunk_binds = __builtin_amdgcn_ballot_w64(binds_t) != 0ull;
        // ---- the sixteen queues ----
        {
            const uint32_t fill = CH * 0x01010101u;   // the sentinel slot in every byte
            uint4 *qf = reinterpret_cast<uint4 *>(lds.queue);
            qf[lane] = make_uint4(fill, fill, fill, fill);
            if (lane == 0) qf[64] = make_uint4(fill, fill, fill, fill);
        }
        int nsteps = 0;
#pragma unroll
        for (int g = 0; g < 16; g++) {
            uint64_t m = __builtin_amdgcn_ballot_w64((msk & (1u << g)) != 0u);
            // slot t has list index hi - t: needed by the block only if hi - t <= its last contributor
            const int d = hi - sgl[g];
            if (d > 0) m = d >= 64 ? 0ull : (m & ~((1ull << d) - 1ull));
            if (__builtin_amdgcn_inverse_ballot_w64(m)) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                lds.queue[g * CH + rank] = (uint8_t)lane;
            }
            const int cnt = __builtin_popcountll(m);
            GS_STAT(11, cnt);
            nsteps = max(nsteps, cnt);
        }
        GS_STAT(12, 1);
        __syncthreads();
        nmask = 0u;
        if (hi - CH - lane >= range.x) { ng = ids[hi - CH - lane]; nmask = masks[hi - CH - lane]; }
        if (nsteps == 0) continue;
        auto walk = [&](auto binds_tag) {
            constexpr bool BINDS = decltype(binds_tag)::value;
            int e_next = myq[0];
            for (int k = 0; k < nsteps; k++) {
                const int e = e_next;
                const float4 q0 = lds.stage[e].p0, q1 = lds.stage[e].p1, q2 = lds.stage[e].p2;
                e_next = myq[k + 1];
                // the claim's first round: issued here, looked at after the passes
                // (an exhausted group has nothing to add; its dx is NaN.  The predicate lives as a scalar mask from here
                // to the claim: as a bool it crossed the passes in a VGPR, two VALU per step to put it there and back)
                const uint64_t actm = __builtin_amdgcn_ballot_w64(e < CH);
                qtag_t *mytag = &lds.tag[copy][e];
                if (__builtin_amdgcn_inverse_ballot_w64(actm))
                    __hip_atomic_store(mytag, (qtag_t)grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t sbits = __float_as_uint(q1.z);
                GS_STAT(8, 1);
                const float dx = q0.x - pxf;
                const float Adxdx = (q0.z * dx) * dx, Bdx = q0.w * dx;
                const float hAdxdx = 0.5f * Adxdx, hC = 0.5f * q1.x;
                // (a scalar mask, not a bool: as a bool it crossed the passes in a VGPR — v_cndmask / v_cmp per step)
                const uint64_t bindm = BINDS ? __builtin_amdgcn_ballot_w64((sbits & 1u) != 0u) : 0ull;
#if !GS_BWDQ_PK
                const bool any_binds = bindm != 0ull;
#endif
#if !GS_BWDQ_PK
                float su = -0.0f, suy = -0.0f, suyy = -0.0f, gr = -0.0f, gg = -0.0f, gb = -0.0f;
#endif
                const int won = (int)__hip_atomic_load(mytag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                float *rec = &lds.acc[copy][e * kAccRec];
                float2 *r2 = reinterpret_cast<float2 *>(rec + 2 * c0);
#if GS_BWDQ_PK
                // The four pixel passes as TWO packed ones (rows 2 j and 2 j + 1 in the halves of 64-bit register
                // pairs): every add / multiply / fma of a pass is one v_pk_*_f32 for two rows, the compares, the
                // exponential, the reciprocal and the selects stay per row.  Same operations per pixel; the six sums
                // over the block's rows are formed as (row 0 + row 2) + (row 1 + row 3).
                v2f su2 = {-0.0f, -0.0f}, suy2 = su2, suyy2 = su2, gr2 = su2, gg2 = su2, gb2 = su2;
                // sigma' of all four rows in front of the per-row decisions: in the block that loaded the record the
                // per-entry scalars fold into the packed instructions' op_sel (behind the branches the compiler
                // copied them into register pairs: four v_mov per step)
                v2f dy2[PX / 2], sg2[PX / 2];
#pragma unroll
                for (int j = 0; j < PX / 2; j++) {
                    dy2[j] = q0.y - pyf2[j];
                    sg2[j] = __builtin_elementwise_fma(hC * dy2[j], dy2[j], (v2f)(hAdxdx));
                    sg2[j] = __builtin_elementwise_fma((v2f)(Bdx), dy2[j], sg2[j]);
                    asm volatile("" : "+v"(sg2[j]), "+v"(dy2[j]));   // (stays here: the machine sinker moved it back)
                }
#pragma unroll
                for (int j = 0; j < PX / 2; j++) {
                    const v2f dy = dy2[j];
                    const v2f sg = sg2[j];
                    float vish[2];
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int p = 2 * j + h;
                        const float dyh = h ? dy.y : dy.x;
                        float sgh = h ? sg.y : sg.x;
                        uint64_t bm = bindm;
                        asm volatile("" : "+s"(bm));   // (every row tests the mask itself: one s_cmp; a shared i1 was
                                                       //  inverted through a VGPR)
                        if (bm != 0ull) {
                            asm volatile("; rectangle binds");
                            if (__builtin_amdgcn_inverse_ballot_w64(bindm)) {
                                // decide exactly like the forward: its op order for sigma, rectangle applied
                                const float4 q3 = lds.rare[e];
                                float se = ((q3.z * dx) * dx) + (q2.w * dyh) * dyh;
                                se = 0.5f * se;
                                se = se + (q3.w * dx) * dyh;
                                const uint32_t rx = __float_as_uint(q3.x), ry = __float_as_uint(q3.y);
                                const uint32_t pyu = (uint32_t)(py0 + p);
                                const bool in = (uint32_t)px >= (rx & 0xFFFFu) && (uint32_t)px < (rx >> 16) &&
                                                pyu >= (ry & 0xFFFFu) && pyu < (ry >> 16);
                                sgh = in ? (se + 0.0f) * kLog2e : qnan();
                            }
                        }
                        const uint64_t mneed = __builtin_amdgcn_ballot_w64(e >= last[p]) &
                                               __builtin_amdgcn_ballot_w64(__float_as_uint(sgh) <= sbits);
                        GS_STAT(9, 1);
                        GS_STAT(10, __builtin_popcountll(mneed));
                        float vis = __builtin_amdgcn_exp2f(-sgh);   // sg = sigma log2(e)
                        uint64_t mok;
                        if (EXACT) {
                            const uint64_t bhi = __builtin_amdgcn_ballot_w64(sgh <= q1.w);
                            mok = mneed & bhi;
                            if (mok != mneed) {   // some lane sits in the band
                                asm volatile("; threshold ambiguous");
                                bool in = false;
                                if (__builtin_amdgcn_inverse_ballot_w64(mneed & ~bhi)) {
                                    const float4 q3 = lds.rare[e];
                                    float se = ((q3.z * dx) * dx) + (q2.w * dyh) * dyh;
                                    se = 0.5f * se;
                                    se = se + (q3.w * dx) * dyh;
                                    vis = expf_glibc_cmem(-se);
                                    in = q1.y * vis >= (1.0f / 255.0f);
                                }
                                mok |= __builtin_amdgcn_ballot_w64(in);
                            }
                        } else {
                            mok = mneed & __builtin_amdgcn_ballot_w64(q1.y * vis >= (1.0f / 255.0f));
                        }
                        vish[h] = __builtin_amdgcn_inverse_ballot_w64(mok) ? vis : 0.0f;
                    }
                    const v2f vis = {vish[0], vish[1]};
                    v2f alpha = q1.y * vis;
                    alpha.x = fminf(alpha.x, 0.99f);
                    alpha.y = fminf(alpha.y, 0.99f);
                    const v2f om = 1.0f - alpha;
                    const v2f ra = {__builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y)};
                    T2[j] = T2[j] * ra;
                    const v2f fac = alpha * T2[j];
                    gr2 = __builtin_elementwise_fma(fac, vo0_2[j], gr2);
                    gg2 = __builtin_elementwise_fma(fac, vo1_2[j], gg2);
                    gb2 = __builtin_elementwise_fma(fac, vo2_2[j], gb2);
                    const v2f cv = __builtin_elementwise_fma(
                        (v2f)(q2.z), vo2_2[j], __builtin_elementwise_fma((v2f)(q2.y), vo1_2[j], q2.x * vo0_2[j]));
                    const v2f v_alpha = __builtin_elementwise_fma(T2[j], cv, ra * D2[j]);
                    D2[j] = __builtin_elementwise_fma(-fac, cv, D2[j]);
                    const v2f u = vis * v_alpha;
                    const v2f uy = u * dy;
                    su2 += u;
                    suy2 += uy;
                    suyy2 = __builtin_elementwise_fma(uy, dy, suyy2);
                }
                const float su = su2.x + su2.y, suy = suy2.x + suy2.y, suyy = suyy2.x + suyy2.y;
                const float gr = gr2.x + gr2.y, gg = gg2.x + gg2.y, gb = gb2.x + gb2.y;
#else
#pragma unroll
                for (int p = 0; p < PX; p++) {
                    const float dy = q0.y - pyf[p];
                    float sg = fmaf(hC * dy, dy, hAdxdx);
                    sg = fmaf(Bdx, dy, sg);
                    if (any_binds) {
                        asm volatile("; rectangle binds");
                        if (sbits & 1u) {
                            // decide exactly like the forward: its op order for sigma, rectangle applied
                            const float4 q3 = lds.rare[e];
                            float se = ((q3.z * dx) * dx) + (q2.w * dy) * dy;
                            se = 0.5f * se;
                            se = se + (q3.w * dx) * dy;
                            const uint32_t rx = __float_as_uint(q3.x), ry = __float_as_uint(q3.y);
                            const uint32_t pyu = (uint32_t)(py0 + p);
                            const bool in = (uint32_t)px >= (rx & 0xFFFFu) && (uint32_t)px < (rx >> 16) &&
                                            pyu >= (ry & 0xFFFFu) && pyu < (ry >> 16);
                            sg = in ? (se + 0.0f) * kLog2e : qnan();
                        }
                    }
                    const uint64_t mneed = __builtin_amdgcn_ballot_w64(e >= last[p]) &
                                           __builtin_amdgcn_ballot_w64(__float_as_uint(sg) <= sbits);
                    // (no "nobody needs this pass" branch: with sixteen Gaussians in flight it is never taken, and
                    // without it the four passes schedule as one block: 243 -> 240 us)
                    GS_STAT(9, 1);
                    GS_STAT(10, __builtin_popcountll(mneed));
                    float vis = __builtin_amdgcn_exp2f(-sg);   // sg = sigma log2(e)
                    uint64_t mok;
                    if (EXACT) {
                        const uint64_t bhi = __builtin_amdgcn_ballot_w64(sg <= q1.w);
                        mok = mneed & bhi;
                        if (mok != mneed) {   // some lane sits in the band
                            asm volatile("; threshold ambiguous");
                            bool in = false;
                            if (__builtin_amdgcn_inverse_ballot_w64(mneed & ~bhi)) {
                                const float4 q3 = lds.rare[e];
                                float se = ((q3.z * dx) * dx) + (q2.w * dy) * dy;
                                se = 0.5f * se;
                                se = se + (q3.w * dx) * dy;
                                vis = expf_glibc_cmem(-se);
                                in = q1.y * vis >= (1.0f / 255.0f);
                            }
                            mok |= __builtin_amdgcn_ballot_w64(in);
                        }
                    } else {
                        mok = mneed & __builtin_amdgcn_ballot_w64(q1.y * vis >= (1.0f / 255.0f));
                    }
                    vis = __builtin_amdgcn_inverse_ballot_w64(mok) ? vis : 0.0f;
                    const float alpha = fminf(q1.y * vis, 0.99f);
                    const float om = 1.0f - alpha;
                    const float ra = __builtin_amdgcn_rcpf(om);
                    T[p] = T[p] * ra;
                    const float fac = alpha * T[p];
                    gr = fmaf(fac, vo0[p], gr);
                    gg = fmaf(fac, vo1[p], gg);
                    gb = fmaf(fac, vo2[p], gb);
                    const float cv = fmaf(q2.z, vo2[p], fmaf(q2.y, vo1[p], q2.x * vo0[p]));
                    const float v_alpha = fmaf(T[p], cv, ra * D[p]);
                    D[p] = fmaf(-fac, cv, D[p]);
                    const float u = vis * v_alpha;
                    const float uy = u * dy;
                    su += u;
                    suy += uy;
                    suyy = fmaf(uy, dy, suyy);
                }
#endif
                const float ux = su * dx;
                float s0, s1, s2;
                orbit_reduce9(ux, suy, ux * dx, suy * dx, suyy, gr, gg, gb, su, s0, s1, s2);
                // ---- add to the entry's record: plain read-add-write by the group that holds the claim ----
                // (lane masks as scalars: the loop's exit test is one s_cmp)
                uint64_t winm = actm & __builtin_amdgcn_ballot_w64(won == grp);
                uint64_t pendm = actm & ~winm;
                float2 a01;
                float a8;
                asm volatile("" : "=v"(a01.x), "=v"(a01.y), "=v"(a8));   // (defined, without three v_mov per step)
                bool loaded = false;   // (wave-uniform)
                for (;;) {
                    if (__builtin_amdgcn_inverse_ballot_w64(winm)) {
                        if (!loaded) { a01 = *r2; a8 = rec[8]; }
                        a01.x += s0; a01.y += s1; a8 += s2;   // (the group's four lanes hold the same s2: same store)
                        *r2 = a01;
                        rec[8] = a8;
                    }
                    if (pendm == 0ull) break;
                    // another round for the groups that lost: the claim and the record — behind the winners'
                    // stores — in ONE LDS round trip (242 -> 236 us at C2; reading the record of the FIRST round
                    // before the passes, to take its round trip off the end of the step, measured 252 us: every
                    // active group then reads, not only the winners)
                    asm volatile("; claim lost: another round");
                    // The losers re-read records the winners have just stored with PLAIN LDS accesses: ordered by
                    // program order and the in-order LDS pipe on the hardware, but a data race under the memory
                    // model — a wavefront-scope release / acquire pair (no instruction on gfx950) keeps the
                    // compiler from forwarding or hoisting across it.
                    wave_sync();
                    int tagv;
                    asm volatile("" : "=v"(tagv));   // (defined; the lanes that do not read it are masked below)
                    if (__builtin_amdgcn_inverse_ballot_w64(pendm)) {
                        __hip_atomic_store(mytag, (qtag_t)grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        tagv = (int)__hip_atomic_load(mytag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        a01 = *r2;
                        a8 = rec[8];
                    }
                    loaded = true;
                    // (the compare outside the masked block: as a bool set inside it went through a VGPR)
                    winm = pendm & __builtin_amdgcn_ballot_w64(tagv == grp);
                    pendm &= ~winm;
                }
            }
        };
        if (chunk_binds) walk(std::true_type{}); else walk(std::false_type{});
        // ---- flush: moments -> gradient components (once per entry), then one atomic lane per
        //      (entry, component): the nine lanes of an entry hit ONE 64-byte record ----
        wave_sync();
        if (msk != 0u) {   // (only slots staged THIS chunk, see backward_wave)
            float *r = &lds.acc[0][lane * kAccRec];
            if (kQCopies > 1) {   // the two colours' sums of this entry
#pragma unroll
                for (int i = 0; i < kAccRec / 2; i++) {
                    const float2 a = myrec[i], b = myrec1[i];
                    myrec[i] = make_float2(a.x + b.x, a.y + b.y);
                }
            }
            const float Ux = r[0], Uy = r[2], Uxx = r[4], Uxy = r[6], Uyy = r[1];
            const float mo = -lds.stage[lane].p1.y;          // v_sigma = -opacity * u
            const float A = lds.rare[lane].z, B = lds.rare[lane].w, C = lds.stage[lane].p2.w;
            r[0] = mo * fmaf(A, Ux, B * Uy);   // v_x: v_sigma * (A dx + B dy)
            r[2] = mo * fmaf(B, Ux, C * Uy);   // v_y
            r[4] = 0.5f * mo * Uxx;            // v_A  (gsplat_cpu.cpp:361-363)
            r[6] = 0.5f * mo * Uxy;            // v_B
            r[1] = 0.5f * mo * Uyy;            // v_C
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < (CH + 6) / 7; i++) {
            const int ent = 7 * i + fj;
            if (fcomp < kAcc && (7 * i + 6 < CH || ent < CH)) {
                const float v = lds.acc[0][ent * kAccRec + fdw];
                if (v != 0.0f) {
                    const size_t o = (size_t)lds.sid[ent] * kGradRec + fcomp;
                    if (DET)
                        atomicAdd(gfix + o, (unsigned long long)__float2ll_rn(
                                                fminf(fmaxf(v * kFixScale, -4.6e18f), 4.6e18f)));
                    else
                        atomicAdd(gacc + o, v);
                }
            }
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < kAccRec / 2; i++) {
            myrec[i] = make_float2(0.0f, 0.0f);
            if (kQCopies > 1) myrec1[i] = make_float2(0.0f, 0.0f);
        }
    }
}

// One geometry for every tile (flag bits 21..22: measurements, the wave-geometry tests).
template <bool EXACT, bool DET, int PX>
// (four pixels per lane at 96 VGPRs keep eight values in scratch, touched once per CHUNK, not per step:
// 356 us against 363 us with four waves per SIMD and none)
__global__ void __launch_bounds__(64, GS_BWD_WAVES)
k_rasterize_backward(int W, int H, int tiles_x, int num_tiles, const int32_t *__restrict__ order,
                     const int32_t *__restrict__ ids, const uint16_t *__restrict__ masks,
                     const int2 *__restrict__ bins,
                     const float4 *__restrict__ packed, float bg0, float bg1, float bg2,
                     const float *__restrict__ bg_dev, const float *__restrict__ final_Ts,
                     const int32_t *__restrict__ final_idx, const float *__restrict__ v_out,
                     const float *__restrict__ v_out_alpha, const float *__restrict__ img_raw,
                     float *__restrict__ gacc, unsigned long long *__restrict__ gfix, int final_first) {
    __shared__ SRecB stage[kChunk + 1];
    __shared__ int sid[kChunk];
    __shared__ float acc[kAccFloats];
    int tile, wx0, wy0;
    if (!decode_wave<PX>(blockIdx.x, num_tiles, tiles_x, W, H, order, tile, wx0, wy0)) return;
    backward_wave<EXACT, DET, PX>(tile, wx0, wy0, stage, sid, acc, W, H, ids, masks, bins, packed, bg0, bg1, bg2,
                                  bg_dev, final_Ts, final_idx, v_out, v_out_alpha, img_raw, gacc, gfix,
                                  ListPiece{0, 0, nullptr, nullptr}, final_first);
}

// The default: four pixels per lane — one wave per tile — amortise the per-step reduction best, but leave
// a tile's whole list to ONE wave; a tile whose list is much longer than the others (clustered scenes: a
// few thousand entries) would finish long after the rest of the frame.  Such tiles — list longer than
// `long_len`, among the first kLongSlots of the longest-first order — are taken by FOUR waves with one
// pixel per lane instead (8 x 8 quadrants, a fourth of the steps each), in the same launch: the first
// 4 * kLongSlots workgroups look at those slots and leave unless the tile is long, the workgroup that owns
// the tile with four pixels per lane leaves if it is.  Both read the same bins and the same threshold.
constexpr int kLongSlots = 256;
template <bool EXACT, bool DET>
__global__ void __launch_bounds__(64, GS_BWD_WAVES)
k_rasterize_backward_mixed(int W, int H, int tiles_x, int num_tiles, int long_len,
                           const int32_t *__restrict__ order, const int32_t *__restrict__ ids,
                           const uint16_t *__restrict__ masks, const int2 *__restrict__ bins,
                           const float4 *__restrict__ packed, float bg0, float bg1, float bg2,
                           const float *__restrict__ bg_dev, const float *__restrict__ final_Ts,
                           const int32_t *__restrict__ final_idx, const float *__restrict__ v_out,
                           const float *__restrict__ v_out_alpha, const float *__restrict__ img_raw,
                           float *__restrict__ gacc, unsigned long long *__restrict__ gfix, int final_first) {
    __shared__ SRecB stage[kChunk + 1];
    __shared__ int sid[kChunk];
    __shared__ float acc[kAccFloats];
    const int b = blockIdx.x;
    if (b < 4 * kLongSlots) {
        const int slot = b >> 2, part = b & 3;
        if (slot >= num_tiles || !order) return;   // (without the longest-first order: no long path)
        const int tile = order[slot];
        const int2 r = bins[tile];
        if (r.y - r.x <= long_len) return;
        const int wx0 = (tile % tiles_x) * GS_TILE + 8 * (part & 1), wy0 = (tile / tiles_x) * GS_TILE + 8 * (part >> 1);
        if (wx0 >= W || wy0 >= H) return;
        backward_wave<EXACT, DET, 1>(tile, wx0, wy0, stage, sid, acc, W, H, ids, masks, bins, packed, bg0, bg1,
                                     bg2, bg_dev, final_Ts, final_idx, v_out, v_out_alpha, img_raw, gacc, gfix,
                                     ListPiece{0, 0, nullptr, nullptr}, final_first);
        return;
    }
    // (decode_wave<4> on the remaining workgroups; the slot decides whether the tile was taken above)
    const int bb = b - 4 * kLongSlots;
    const int slot = ((bb >> 3) << 3) + (bb & 7);
    if (slot >= num_tiles) return;
    const int tile = order ? order[slot] : xcd_swizzle(slot, num_tiles);
    if (slot < kLongSlots && order) {
        const int2 r = bins[tile];
        if (r.y - r.x > long_len) return;
    }
    const int wx0 = (tile % tiles_x) * GS_TILE, wy0 = (tile / tiles_x) * GS_TILE;
    backward_wave<EXACT, DET, 4>(tile, wx0, wy0, stage, sid, acc, W, H, ids, masks, bins, packed, bg0, bg1, bg2,
                                 bg_dev, final_Ts, final_idx, v_out, v_out_alpha, img_raw, gacc, gfix,
                                 ListPiece{0, 0, nullptr, nullptr}, final_first);
}
// The full-frame default since round 5: every tile one wave of sixteen four-lane groups (backward_wave_q).
template <bool EXACT, bool DET>
__global__ void __launch_bounds__(64, GS_BWDQ_WAVES)
k_rasterize_backward_q(int W, int H, int tiles_x, int num_tiles, const int32_t *__restrict__ order,
                       const int32_t *__restrict__ ids, const uint16_t *__restrict__ masks,
                       const int2 *__restrict__ bins, const float4 *__restrict__ packed, float bg0, float bg1,
                       float bg2, const float *__restrict__ bg_dev, const float *__restrict__ final_Ts,
                       const int32_t *__restrict__ final_idx, const float *__restrict__ v_out,
                       const float *__restrict__ v_out_alpha, const float *__restrict__ img_raw,
                       float *__restrict__ gacc, unsigned long long *__restrict__ gfix, int final_first) {
    __shared__ QLds lds;
    int tile, wx0, wy0;
    if (!decode_wave<4>(blockIdx.x, num_tiles, tiles_x, W, H, order, tile, wx0, wy0)) return;
    GS_WLOG_T0
    backward_wave_q<EXACT, DET>(tile, wx0, wy0, lds, W, H, ids, masks, bins, packed, bg0, bg1, bg2, bg_dev,
                                final_Ts, final_idx, v_out, v_out_alpha, img_raw, gacc, gfix, final_first);
    GS_WLOG_T1(tile, bins[tile].y - bins[tile].x);
}
// ... and with the few outlying lists of a frame taken by four waves of one pixel per lane each, exactly as in
// k_rasterize_backward_mixed (one LDS allocation serves either layout).
struct ClassicLds {
    SRecB stage[kChunk + 1];
    int sid[kChunk];
    float acc[kAccFloats];
};
template <bool EXACT, bool DET>
__global__ void __launch_bounds__(64, GS_BWDQ_WAVES)
k_rasterize_backward_q_mixed(int W, int H, int tiles_x, int num_tiles, int long_len,
                             const int32_t *__restrict__ order, const int32_t *__restrict__ ids,
                             const uint16_t *__restrict__ masks, const int2 *__restrict__ bins,
                             const float4 *__restrict__ packed, float bg0, float bg1, float bg2,
                             const float *__restrict__ bg_dev, const float *__restrict__ final_Ts,
                             const int32_t *__restrict__ final_idx, const float *__restrict__ v_out,
                             const float *__restrict__ v_out_alpha, const float *__restrict__ img_raw,
                             float *__restrict__ gacc, unsigned long long *__restrict__ gfix, int final_first) {
    constexpr size_t kBytes = sizeof(QLds) > sizeof(ClassicLds) ? sizeof(QLds) : sizeof(ClassicLds);
    __shared__ __attribute__((aligned(16))) unsigned char raw[kBytes];
    const int b = blockIdx.x;
    if (b < 4 * kLongSlots) {
        const int slot = b >> 2, part = b & 3;
        if (slot >= num_tiles || !order) return;   // (without the longest-first order: no long path)
        const int tile = order[slot];
        const int2 r = bins[tile];
        if (r.y - r.x <= long_len) return;
        const int wx0 = (tile % tiles_x) * GS_TILE + 8 * (part & 1), wy0 = (tile / tiles_x) * GS_TILE + 8 * (part >> 1);
        if (wx0 >= W || wy0 >= H) return;
        ClassicLds &c = *reinterpret_cast<ClassicLds *>(raw);
        backward_wave<EXACT, DET, 1>(tile, wx0, wy0, c.stage, c.sid, c.acc, W, H, ids, masks, bins, packed, bg0,
                                     bg1, bg2, bg_dev, final_Ts, final_idx, v_out, v_out_alpha, img_raw, gacc,
                                     gfix, ListPiece{0, 0, nullptr, nullptr}, final_first);
        return;
    }
    const int bb = b - 4 * kLongSlots;
    const int slot = ((bb >> 3) << 3) + (bb & 7);
    if (slot >= num_tiles) return;
    const int tile = order ? order[slot] : xcd_swizzle(slot, num_tiles);
    if (slot < kLongSlots && order) {
        const int2 r = bins[tile];
        if (r.y - r.x > long_len) return;
    }
    const int wx0 = (tile % tiles_x) * GS_TILE, wy0 = (tile / tiles_x) * GS_TILE;
    backward_wave_q<EXACT, DET>(tile, wx0, wy0, *reinterpret_cast<QLds *>(raw), W, H, ids, masks, bins, packed,
                                bg0, bg1, bg2, bg_dev, final_Ts, final_idx, v_out, v_out_alpha, img_raw, gacc,
                                gfix, final_first);
}
// Frames of few tiles (the reduced resolutions a training run starts with, small captures): four waves per
// tile leave most of the chip's wave slots empty, and a lone wave issues one instruction every four cycles
// whatever it is — the launch lasts as long as the LONGEST list takes one wave (170 - 200 ns per entry,
// profiles/timeline_sweep_*_r04.json).  The backward, unlike the forward, can start anywhere in a list if
// it is handed the state in front of that entry: the forward stores it (a checkpoint record per pixel every
// 1 << seg_shift entries, k_rasterize_forward), and here every piece of every list is a wave of its own —
// tile-major (the pieces of a tile on one XCD), longest list first.  The pieces of a Gaussian's gradient meet
// in its record like the tiles' always did.  (The reference walks a tile's list in one workgroup,
// backward.cu:217-353; its transmittance recurrence is the one of backward_wave.)
template <bool EXACT, bool DET, int PX>
__global__ void __launch_bounds__(64, GS_BWD_WAVES)
k_rasterize_backward_seg(int W, int H, int tiles_x, int num_tiles, int seg_shift, int max_seg,
                         const float4 *__restrict__ ckpt, const int32_t *__restrict__ order,
                         const int32_t *__restrict__ ids, const uint16_t *__restrict__ masks,
                         const int2 *__restrict__ bins, const float4 *__restrict__ packed, float bg0,
                         float bg1, float bg2, const float *__restrict__ bg_dev,
                         const float *__restrict__ final_Ts, const int32_t *__restrict__ final_idx,
                         const float *__restrict__ v_out, const float *__restrict__ v_out_alpha,
                         const float *__restrict__ img_raw, float *__restrict__ gacc,
                         unsigned long long *__restrict__ gfix) {
    __shared__ SRecB stage[kChunk + 1];
    __shared__ int sid[kChunk];
    __shared__ float acc[kAccFloats];
    // block -> (XCD x, k): k = (slot / 8, piece, part of the tile), like decode_wave<PX> with PER_TILE * max_seg
    // parts per tile
    using G = WaveGeom<PX>;
    const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int per_tile = G::PER_TILE * max_seg;
    const int part = k % per_tile, slot = ((k / per_tile) << 3) + x;
    if (slot >= num_tiles) return;
    const int tile = order ? order[slot] : xcd_swizzle(slot, num_tiles);
    const int seg = part / G::PER_TILE, sub = part % G::PER_TILE;
    const int2 r = bins[tile];
    const int lo = r.x + (seg << seg_shift);
    if (lo >= r.y) return;
    // (the last piece the buffer has a record for takes whatever is left of a list that outgrew the plan)
    const bool tail = seg == max_seg - 1 || lo + (1 << seg_shift) >= r.y;
    const int hi = tail ? r.y : lo + (1 << seg_shift);
    constexpr int PX_COLS = GS_TILE / G::WW;
    const int wx0 = (tile % tiles_x) * GS_TILE + G::WW * (sub % PX_COLS);
    const int wy0 = (tile / tiles_x) * GS_TILE + G::WH * (sub / PX_COLS);
    if (wx0 >= W || wy0 >= H) return;
    const float4 *rec = ckpt + (size_t)tile * max_seg * (GS_TILE * GS_TILE);
    const ListPiece piece{lo, hi, tail ? nullptr : rec + (size_t)(seg + 1) * (GS_TILE * GS_TILE), rec};
    backward_wave<EXACT, DET, PX, true>(tile, wx0, wy0, stage, sid, acc, W, H, ids, masks, bins, packed, bg0, bg1,
                                       bg2, bg_dev, final_Ts, final_idx, v_out, v_out_alpha, img_raw, gacc,
                                       gfix, piece);
}
#undef GS_WALK_STEP
#undef GS_WALK_PACK


// Splits the 64-byte gradient records into the four tensors the operator surface returns
// (rasterize_gaussians.cpp:113-124): v_xy[N,2] v_conic[N,3] v_colors[N,3] v_opacity[N].
__global__ void __launch_bounds__(256)
k_unpack_grads(int N, const float4 *__restrict__ gacc, const float4 *__restrict__ packed_logit,
               float *__restrict__ v_xy, float *__restrict__ v_conic, float *__restrict__ v_colors,
               float *__restrict__ v_opacity) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float4 a = gacc[(kGradRec / 4) * (size_t)n + 0], b = gacc[(kGradRec / 4) * (size_t)n + 1];
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
                                (lane & 2) != 0);
    const int c = reduce9_component(li);
    if (c >= 0) out[((size_t)blockIdx.x * 4 + (lane >> 4)) * 9 + c] = r;
}

// Test hook: mfma_reduce9 on given values.  in [blocks, 9, 64] -> out [blocks, 4, 9] (group, value),
// group g = the lanes {4 g + q + 16 k}.
__global__ void __launch_bounds__(64) k_debug_mfma_reduce9(const float *__restrict__ in,
                                                           float *__restrict__ out) {
    const int lane = threadIdx.x;
    const float *p = in + (size_t)blockIdx.x * 9 * 64;
    float v[9], oh[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        v[i] = p[i * 64 + lane];
        oh[i] = (lane & 15) == i ? 1.0f : 0.0f;
    }
    const float r = mfma_reduce9(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], oh);
    if ((lane & 15) < 9) out[((size_t)blockIdx.x * 4 + (lane >> 4)) * 9 + (lane & 15)] = r;
}

// GS_FLAG_DETERMINISTIC: 64-bit fixed-point sums -> the float records
__global__ void __launch_bounds__(256)
k_fixed_to_records(int64_t n, const long long *__restrict__ fix, float *__restrict__ rec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rec[i] = (float)((double)fix[i] * (1.0 / 1099511627776.0));
}

}  // namespace gs


extern "C" int gs_debug_expf(int64_t n, const float *x, float *y, uint32_t flags,
                             gs_stream_t stream) {
    if (n < 0) return GS_ERR_INVALID_ARGUMENT;
    if (n == 0) return GS_OK;
    if (!x || !y) return GS_ERR_INVALID_ARGUMENT;
    int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (flags & GS_FLAG_FAST_EXP)
        GS_LAUNCH(gs::k_debug_expf<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           n, x, y);
    else
        GS_LAUNCH(gs::k_debug_expf<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
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
#ifdef GS_WAVELOG
// the per-workgroup log of the LAST forward / backward_q launch: n records of four words (see GS_WLOG_T1)
extern "C" int gs_debug_wavelog(unsigned long long *host, int n) {
    GS_HIP_CHECK(hipDeviceSynchronize());
    if (!host || n < 0 || n > gs::kWaveLog) return GS_ERR_INVALID_ARGUMENT;
    GS_HIP_CHECK(hipMemcpyFromSymbol(host, HIP_SYMBOL(gs::g_wavelog), (size_t)n * 32));
    return GS_OK;
}
#endif


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


extern "C" size_t gs_rasterize_backward_workspace_bytes(int N) {
    return N > 0 ? (size_t)N * gs::kGradRec * sizeof(float) : 0;
}


extern "C" int gs_debug_row_reduce9(int blocks, const float *in, float *out, gs_stream_t stream) {
    if (blocks < 0) return GS_ERR_INVALID_ARGUMENT;
    if (blocks == 0) return GS_OK;
    if (!in || !out) return GS_ERR_INVALID_ARGUMENT;
    GS_LAUNCH(gs::k_debug_row_reduce9, dim3(blocks), dim3(64), 0, (hipStream_t)stream, in, out);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" int gs_debug_group_reduce9(int blocks, const float *in, float *out, int mfma,
                                      gs_stream_t stream) {
    if (blocks < 0) return GS_ERR_INVALID_ARGUMENT;
    if (blocks == 0) return GS_OK;
    if (!in || !out) return GS_ERR_INVALID_ARGUMENT;
    if (mfma)
        GS_LAUNCH(gs::k_debug_mfma_reduce9, dim3(blocks), dim3(64), 0, (hipStream_t)stream, in, out);
    else
        GS_LAUNCH(gs::k_debug_row_reduce9, dim3(blocks), dim3(64), 0, (hipStream_t)stream, in, out);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" int gs_debug_backward_uses_mfma(void) { return GS_BWD_MFMA; }

extern "C" size_t gs_rasterize_backward_workspace_bytes_det(int N) {
    // float records + the 64-bit fixed-point accumulators of GS_FLAG_DETERMINISTIC
    return N > 0 ? (size_t)N * gs::kGradRec * (sizeof(float) + sizeof(long long)) : 0;
}

// Wave slots of the chip (256 CUs x 4 SIMDs x 5 waves).  A frame whose tiles cannot fill them with four / two
// waves each (1280 / 2560 tiles) is composited by lone waves: the backward gives every tile four / two waves
// — or, with checkpoints, every piece of every list a wave — and the forward takes two entries per step up
// to 2560 tiles (measured: 640x480 119 -> 105 us, 1008x756 114 -> 103 us at 20 000 Gaussians, level at
// 1504x1000; pieces: 752x500 234 -> 171 us, 1008x756 290 -> 282 us, 1504x1000 330 -> 581 us).
constexpr int kWaveSlots = 5120;
// one-wave workgroups of a kernel the chip holds at once (occupancy x CUs; asked once per kernel)
template <class K>
static int resident_slots(K kernel) {
    static const int slots = [&] {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, 0) != hipSuccess || per_cu <= 0)
            return kWaveSlots;
        return per_cu * cus;
    }();
    return slots;
}

static bool checkpoint_args_ok(const void *checkpoints, size_t checkpoint_bytes, int32_t seg_len,
                               int32_t max_segments, int tiles) {
    if (!checkpoints) return true;
    if (seg_len < 64 || (seg_len & (seg_len - 1)) != 0 || max_segments < 2) return false;
    if ((uintptr_t)checkpoints & 15u) return false;
    return checkpoint_bytes >= (size_t)tiles * (size_t)max_segments * GS_TILE * GS_TILE * sizeof(float4);
}

// Pixels per lane of the pieces' waves, by the frame's tile count (measured, 64-entry pieces, SfM-like scenes:
// 30 tiles 25 / 31 / 44 us with 1 / 2 / 4; 432 tiles 48 / 45 / 54 at 6000 and 68 / 55 / 63 at 20 000 Gaussians;
// 1200 tiles 133 / 94 / 82; 1504 tiles 171 / 113 / 99; 3024 tiles 282 / 170 / 128).
static int piece_pixels_per_lane(int tiles) { return tiles <= 128 ? 1 : tiles <= 1024 ? 2 : 4; }

extern "C" int gs_rasterize_checkpoint_plan(int W, int H, const int32_t *list_stats, int32_t *seg_len,
                                            int32_t *max_segments, size_t *bytes) {
    if (W <= 0 || H <= 0 || !seg_len || !max_segments || !bytes) return GS_ERR_INVALID_ARGUMENT;
    *seg_len = 0; *max_segments = 0; *bytes = 0;
    const int tiles = ((W + GS_TILE - 1) / GS_TILE) * ((H + GS_TILE - 1) / GS_TILE);
    // Where pieces pay (SfM-like scenes, one-pass -> pieces, backward kernel alone): 432 tiles 141 -> 45 us,
    // 1504 tiles 234 -> 99, 3024 tiles 288 -> 128, 5922 tiles 402 -> 315; a full 1080p frame (8160 tiles) gains
    // 10 % on a scene with a long tail of lists and loses 5 % on BASELINE config 2's even one (no tail to cut,
    // and its opaque Gaussians make the forward look for hot entries in most chunks: 170 -> 178 us).
    constexpr int kMaxTiles = 6144;
    if (tiles > kMaxTiles || !list_stats || list_stats[0] <= 0) return GS_OK;
    const int64_t longest = list_stats[1], mean = ((int64_t)list_stats[0] + tiles - 1) / tiles;
    // pieces of one chunk (64 entries): 6000 Gaussians at 384x288 47.6 us, against 58.9 / 75.3 / 126.8 us with
    // 128 / 256 / 512 and 141 us in one pass; 96x72: 25 / 40 / 65 / 107 / 263 us (profiles/HISTORY.md)
    int64_t len = 64;
    if (longest <= 2 * len) return GS_OK;   // nothing worth cutting
    // a frame whose one-pass launch (four / two / one wave per tile) comes near the chip's wave slots gains only by
    // its tail — lists far beyond the mean.  Even lists (uniform synthetic scenes, opaque Gaussians), one pass ->
    // pieces: 752x500 5659 -> 5525 /s, 1008x756 3713 -> 3678 and 1295 -> 1318 /s, 1504x1000 2473 -> 2310 /s.
    constexpr int kAlwaysTiles = 960;
    if (tiles > kAlwaysTiles && longest < 4 * mean) return GS_OK;
    // the longest list of the frame the statistics come from + a quarter: a list that outgrows it is
    // finished by its last piece (slower, correct).  Every (tile, piece) is a workgroup, most of which look at
    // their tile's list and leave: longer pieces before the launch exceeds a quarter of a million of them.
    auto pieces = [&](int64_t l) { return (longest + longest / 4 + l - 1) / l + 1; };
    // ... and before the records (4 KiB per (tile, piece), allocated per frame by the callers and held until the
    // backward) exceed 512 MiB (64-entry pieces measured best wherever they fit: a 1008 x 756 frame
    // with a 2000-entry list takes 508 MB).  A frame that needs more even at 8192-entry pieces is walked in one pass.
    constexpr int64_t kMaxBytes = 512ll << 20;
    constexpr int64_t kRecord = GS_TILE * GS_TILE * sizeof(float4);
    while ((len < 1024 && (int64_t)tiles * pieces(len) > (1 << 18)) ||
           (len < 8192 && (int64_t)tiles * pieces(len) * kRecord > kMaxBytes))
        len *= 2;
    const int64_t segs = std::min<int64_t>(pieces(len), 4096);
    if ((int64_t)tiles * segs * kRecord > kMaxBytes) return GS_OK;
    *seg_len = (int32_t)len;
    *max_segments = (int32_t)segs;
    *bytes = (size_t)tiles * (size_t)segs * kRecord;
    return GS_OK;
}

extern "C" int gs_rasterize_forward_ckpt(int W, int H, const int32_t *gaussian_ids_sorted,
                                         const uint16_t *block_masks, const int32_t *tile_bins,
                                         const float *packed, const float *background, float *out_img,
                                         float *final_Ts, int32_t *final_idx, float *out_img_clamped,
                                         const int32_t *list_stats, const int32_t *tile_order,
                                         uint32_t flags, void *checkpoints, size_t checkpoint_bytes,
                                         int32_t seg_len, int32_t max_segments, gs_stream_t stream) {
    GS_TRACE("gs_rasterize_forward");
    if (W <= 0 || H <= 0) return GS_ERR_INVALID_ARGUMENT;
    if ((flags & GS_FLAG_CLAMP_IMAGE) && !out_img_clamped) return GS_ERR_INVALID_ARGUMENT;
    float *clamped = (flags & GS_FLAG_CLAMP_IMAGE) ? out_img_clamped : nullptr;
    if (W > 65535 || H > 65535) return GS_ERR_UNSUPPORTED;
    if (!tile_bins || !background || !out_img || !final_Ts || !final_idx)
        return GS_ERR_INVALID_ARGUMENT;
    if (gaussian_ids_sorted && !block_masks) return GS_ERR_INVALID_ARGUMENT;
    if ((uintptr_t)packed & 15u) return GS_ERR_INVALID_ARGUMENT;
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    const int tiles = tiles_x * tiles_y;
    if (!checkpoint_args_ok(checkpoints, checkpoint_bytes, seg_len, max_segments, tiles))
        return GS_ERR_INVALID_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    const int2 *bins = reinterpret_cast<const int2 *>(tile_bins);
    const float4 *pk = reinterpret_cast<const float4 *>(packed);
    const int units = gs::WaveGeom<1>::PER_TILE * 8 * ((tiles + 7) / 8);  // see decode_wave
    const float *bg_dev = gs::on_device(background) ? background : nullptr;
    const float bg0 = bg_dev ? 0.f : background[0], bg1 = bg_dev ? 0.f : background[1],
                bg2 = bg_dev ? 0.f : background[2];
    float4 *ck = static_cast<float4 *>(checkpoints);
    const int seg_shift = ck ? __builtin_ctz((unsigned)seg_len) : 0;
    // entries per step: two on a frame of lone waves (flag bits 23..24 force 1 / 2: measurements, tests)
    int ilp = 2 * tiles <= kWaveSlots ? 2 : 1;
    if (((flags >> 23) & 3u) != 0u) ilp = (int)((flags >> 23) & 3u) == 2 ? 2 : 1;
    gs::ev_before(s);
#define GS_FWD_LAUNCH3(EX, IL, CK)                                                                       \
    GS_LAUNCH((gs::k_rasterize_forward<EX, IL, CK>), dim3(units), dim3(64), 0, s, W, H, tiles_x, tiles,   \
              tile_order, gaussian_ids_sorted, block_masks, bins, pk, bg0, bg1, bg2, bg_dev, out_img,     \
              final_Ts, final_idx, clamped, ck, seg_shift, (int)max_segments)
#define GS_FWD_LAUNCH(EX, IL)                                                                            \
    do {                                                                                                 \
        if (ck) GS_FWD_LAUNCH3(EX, IL, true); else GS_FWD_LAUNCH3(EX, IL, false);                        \
    } while (0)
    // flag bits 23..24 = 3: the forward that compacts its chunks (k_rasterize_forward_c, round 6) on full frames
    // without checkpoints — measured 162 -> 160 us at C2 and 1119 -> 1191 us at C3 (a quadrant touches 43 % of a
    // tile's list at C2, two thirds at C3: little to compact, and the ring costs): NOT the default; kept for the
    // bit-for-bit test of the two (tests/test_gpu_forward_compact.py) and the next look
#define GS_FWD_LAUNCH_C(EX)                                                                               \
    GS_LAUNCH((gs::k_rasterize_forward_c<EX>), dim3(units), dim3(64), 0, s, W, H, tiles_x, tiles, tile_order, \
              gaussian_ids_sorted, block_masks, bins, pk, bg0, bg1, bg2, bg_dev, out_img, final_Ts, final_idx,  \
              clamped)
    const bool compact = ilp == 1 && !ck && (((flags >> 23) & 3u) == 3u || GS_FWD_COMPACT);
    if (compact) {
        if (flags & GS_FLAG_FAST_EXP) GS_FWD_LAUNCH_C(false); else GS_FWD_LAUNCH_C(true);
    } else if (flags & GS_FLAG_FAST_EXP) {
        if (ilp == 2) GS_FWD_LAUNCH(false, 2); else GS_FWD_LAUNCH(false, 1);
    } else {
        if (ilp == 2) GS_FWD_LAUNCH(true, 2); else GS_FWD_LAUNCH(true, 1);
    }
#undef GS_FWD_LAUNCH_C
#undef GS_FWD_LAUNCH
#undef GS_FWD_LAUNCH3
    gs::ev_after(s);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" int gs_rasterize_forward(int W, int H, const int32_t *gaussian_ids_sorted,
                                    const uint16_t *block_masks, const int32_t *tile_bins,
                                    const float *packed,
                                    const float *background, float *out_img, float *final_Ts,
                                    int32_t *final_idx, float *out_img_clamped,
                                    const int32_t *list_stats, const int32_t *tile_order,
                                    uint32_t flags, gs_stream_t stream) {
    return gs_rasterize_forward_ckpt(W, H, gaussian_ids_sorted, block_masks, tile_bins, packed, background,
                                     out_img, final_Ts, final_idx, out_img_clamped, list_stats, tile_order,
                                     flags, nullptr, 0, 0, 0, stream);
}

extern "C" int gs_rasterize_backward_ckpt(int W, int H, int N, const int32_t *gaussian_ids_sorted,
                                          const uint16_t *block_masks, const int32_t *tile_bins,
                                          const float *packed, const float *background,
                                          const float *final_Ts, const int32_t *final_idx,
                                          const float *v_out, const float *v_out_alpha,
                                          const float *out_img, float *v_xy, float *v_conic,
                                          float *v_colors, float *v_opacity, void *workspace,
                                          size_t workspace_bytes, const int32_t *list_stats,
                                          const int32_t *tile_order, uint32_t flags,
                                          const void *checkpoints, size_t checkpoint_bytes,
                                          int32_t seg_len, int32_t max_segments, gs_stream_t stream) {
    GS_TRACE("gs_rasterize_backward");
    if (W <= 0 || H <= 0 || N < 0) return GS_ERR_INVALID_ARGUMENT;
    if ((flags & GS_FLAG_CLAMP_IMAGE) && !out_img) return GS_ERR_INVALID_ARGUMENT;
    const float *img_raw = (flags & GS_FLAG_CLAMP_IMAGE) ? out_img : nullptr;
    if (W > 65535 || H > 65535) return GS_ERR_UNSUPPORTED;
    if (N == 0) return GS_OK;
    const bool keep_records = (flags & GS_FLAG_KEEP_RECORDS) != 0u;
    const bool det = (flags & GS_FLAG_DETERMINISTIC) != 0u;
    if (!tile_bins || !background || !final_Ts || !final_idx || !v_out || !workspace)
        return GS_ERR_INVALID_ARGUMENT;
    if (gaussian_ids_sorted && !block_masks) return GS_ERR_INVALID_ARGUMENT;
    if (!keep_records && (!v_xy || !v_conic || !v_colors || !v_opacity)) return GS_ERR_INVALID_ARGUMENT;
    if (((uintptr_t)packed & 15u) || ((uintptr_t)workspace & 63u)) return GS_ERR_INVALID_ARGUMENT;
    const size_t rec_bytes = gs_rasterize_backward_workspace_bytes(N);
    if (workspace_bytes < (det ? gs_rasterize_backward_workspace_bytes_det(N) : rec_bytes))
        return GS_ERR_WORKSPACE;
    const int tiles_x = (W + GS_TILE - 1) / GS_TILE, tiles_y = (H + GS_TILE - 1) / GS_TILE;
    const int tiles = tiles_x * tiles_y;
    if (!checkpoint_args_ok(checkpoints, checkpoint_bytes, seg_len, max_segments, tiles))
        return GS_ERR_INVALID_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    const int2 *bins = reinterpret_cast<const int2 *>(tile_bins);
    const float4 *pk = reinterpret_cast<const float4 *>(packed);
    float *gacc = static_cast<float *>(workspace);
    unsigned long long *gfix =
        det ? reinterpret_cast<unsigned long long *>(static_cast<char *>(workspace) + rec_bytes) : nullptr;
    if (det)
        GS_HIP_CHECK(hipMemsetAsync(gfix, 0, (size_t)N * gs::kGradRec * sizeof(long long), s));
    else if (!(flags & GS_FLAG_RECORDS_ZEROED))
    {
        gs::timeline_before(s);
        GS_HIP_CHECK(hipMemsetAsync(gacc, 0, rec_bytes, s));
        gs::timeline_after("memset(gradient records)", s);
    }
    // Wave geometry of the backward (WaveGeom): four pixels per lane — one wave per tile, 8 x 8 blocks —
    // amortise the per-step reduction best (338 against 378 / 500 us with two / one at C2), and the
    // default launch gives the few tiles whose list is far longer than the others four waves with one
    // pixel per lane (k_rasterize_backward_mixed).  "Far longer": beyond 2 x the mean list and 512
    // entries (a wave slot works through tiles / slots ~ 1.6 mean lists at 1080p: a list much longer
    // than that finishes after everything else), from the scan's {M, longest list} of the previous frame when the caller has it (a stale
    // value costs time, not correctness: both halves of the launch read the same threshold).  Flag bits
    // 21..22 select 1 / 2 / 4 pixels per lane for every tile (measurements, tests).
    int px_per_lane = 0;   // 0: mixed
    int long_len = 512;
    if (list_stats && list_stats[0] > 0) {
        const int64_t mean_len = ((int64_t)list_stats[0] + tiles - 1) / tiles;
        long_len = (int)std::min<int64_t>(std::max<int64_t>(2 * mean_len, 512), 1 << 30);
        // no list was that long in the frame the statistics come from: the plain launch (the mixed one
        // costs 256 workgroups that look and leave, 1.5 us at C2)
        if (list_stats[1] <= long_len) px_per_lane = 4;
    }
    // a small frame does not fill the chip with one wave per tile (256 CUs x 4 SIMDs x 5 waves = 5120 slots;
    // the quarter-resolution frames a training run starts with have a few dozen tiles): more waves per
    // tile then cost 1.1 / 1.5 x the work and take a half / a quarter of the time
    if (4 * tiles <= kWaveSlots) px_per_lane = 1;
    else if (2 * tiles <= kWaveSlots) px_per_lane = 2;
    if (((flags >> 21) & 3u) != 0u) px_per_lane = 1 << (((flags >> 21) & 3u) - 1u);
    // Every tile one wave with four pixels per lane: since round 5 that wave is sixteen four-lane groups
    // (backward_wave_q: 286 -> 244 us at BASELINE config 2, level at config 3).  Flag bits 25..26: 1 = that
    // geometry for every frame (tests), 2 = the four-group kernels of rounds 2 - 4 (measurements).
    // ... where the footprints are SMALL: what the sixteen groups buy is live lanes on 4 x 4 blocks that a Gaussian
    // covers half of; splats of tens of pixels (SfM-initialised scenes at the start of a run) fill whole tiles,
    // every entry then has sixteen (block, entry) pairs colliding on its record, and the four-group kernels are
    // 1.3 ... 1.5 x faster (100 000 Gaussians at 1504 x 1000: 126 against 163 us; the pieces with sixteen groups,
    // built and measured: 188 -> 240 us at 1008 x 756 — not kept).  The statistic at hand is the list entries per
    // Gaussian, M / N, of the frame the list statistics come from: 2.2 at BASELINE config 2, 3.8 at config 3,
    // 15 ... 35 on those scenes; sixteen groups up to kQMaxEntriesPerGaussian.  The threshold is the measured
    // crossover (round 6, scripts/sweep_bwd_selection.py, profiles/r06/bwd_selection_sweep.json: 500 000 Gaussians
    // at 1080p, splat size swept, both kernels on the same lists — sixteen-group time / four-group time at M / N =
    // 2.2: 0.81, 2.7: 0.95, 3.4: 0.96, 4.2: 1.02, 5.1: 1.04, 6.0: 1.11, 8.8: 1.13, 12: 1.12); rounds 5's "6" stood
    // on four points.  A frame without statistics (the first one, or after a reset) takes the four-group kernels:
    // 178 instead of 144 us at M / N = 2.2, once.
    constexpr int64_t kQMaxEntriesPerGaussian = 4;
    const int qsel = (int)((flags >> 25) & 3u);
    const bool small_footprints = list_stats && list_stats[0] > 0 &&
                                  (int64_t)list_stats[0] <= kQMaxEntriesPerGaussian * (int64_t)N;
    const bool use_q = !checkpoints && ((flags >> 21) & 3u) == 0u && qsel != 2 &&
                       (qsel == 1 || ((px_per_lane == 4 || px_per_lane == 0) && small_footprints));
    const bool q_mixed = use_q && px_per_lane == 0;
    const float4 *ck = static_cast<const float4 *>(checkpoints);
    const int seg_shift = ck ? __builtin_ctz((unsigned)seg_len) : 0;
    // pieces: pixels per lane by the tile count unless the flag bits say otherwise
    const int seg_px = ((flags >> 21) & 3u) != 0u ? px_per_lane : piece_pixels_per_lane(tiles);
    const int units = ck ? (4 / seg_px) * max_segments * 8 * ((tiles + 7) / 8)
                    : use_q ? (q_mixed ? 4 * gs::kLongSlots : 0) + 8 * ((tiles + 7) / 8)
                    : px_per_lane == 0 ? 4 * gs::kLongSlots + 8 * ((tiles + 7) / 8)
                                       : (4 / px_per_lane) * 8 * ((tiles + 7) / 8);  // PER_TILE waves per tile
    const float *bg_dev = gs::on_device(background) ? background : nullptr;
    const float bg0 = bg_dev ? 0.f : background[0], bg1 = bg_dev ? 0.f : background[1],
                bg2 = bg_dev ? 0.f : background[2];
    // first workgroup of the launch's final residents (wave_prio): grid - slots; a launch of less than two rounds of
    // waves has nothing to stagger — everybody equalises
#define GS_FINAL_FIRST(K) std::max(units - resident_slots(K), 0)
    gs::ev_before(s);
#define GS_BWD_LAUNCH3(EX, DT, PXN)                                                                       \
    GS_LAUNCH((gs::k_rasterize_backward<EX, DT, PXN>), dim3(units), dim3(64), 0, s, W, H, tiles_x, \
                       tiles, tile_order, gaussian_ids_sorted, block_masks, bins, pk, bg0, bg1, bg2,      \
                       bg_dev, final_Ts, final_idx, v_out, v_out_alpha, img_raw, gacc, gfix,              \
                       GS_FINAL_FIRST((gs::k_rasterize_backward<EX, DT, PXN>)))
#define GS_SEG_LAUNCH3(EX, DT, PXN)                                                                       \
    GS_LAUNCH((gs::k_rasterize_backward_seg<EX, DT, PXN>), dim3(units), dim3(64), 0, s, W, H, tiles_x,     \
              tiles, seg_shift, (int)max_segments, ck, tile_order, gaussian_ids_sorted, block_masks, bins, \
              pk, bg0, bg1, bg2, bg_dev, final_Ts, final_idx, v_out, v_out_alpha, img_raw, gacc, gfix)
#define GS_Q_LAUNCH3(EX, DT)                                                                              \
    do {                                                                                                  \
        if (q_mixed)                                                                                      \
            GS_LAUNCH((gs::k_rasterize_backward_q_mixed<EX, DT>), dim3(units), dim3(64), 0, s, W, H,      \
                      tiles_x, tiles, long_len, tile_order, gaussian_ids_sorted, block_masks, bins, pk,   \
                      bg0, bg1, bg2, bg_dev, final_Ts, final_idx, v_out, v_out_alpha, img_raw, gacc,      \
                      gfix, GS_FINAL_FIRST((gs::k_rasterize_backward_q_mixed<EX, DT>)));                  \
        else                                                                                              \
            GS_LAUNCH((gs::k_rasterize_backward_q<EX, DT>), dim3(units), dim3(64), 0, s, W, H, tiles_x,   \
                      tiles, tile_order, gaussian_ids_sorted, block_masks, bins, pk, bg0, bg1, bg2,       \
                      bg_dev, final_Ts, final_idx, v_out, v_out_alpha, img_raw, gacc, gfix,               \
                      GS_FINAL_FIRST((gs::k_rasterize_backward_q<EX, DT>)));                              \
    } while (0)
#define GS_BWD_LAUNCH(EX, DT)                                                                              \
    do {                                                                                                   \
        if (use_q) GS_Q_LAUNCH3(EX, DT);                                                                   \
        else if (ck && seg_px == 1) GS_SEG_LAUNCH3(EX, DT, 1);                                             \
        else if (ck && seg_px == 2) GS_SEG_LAUNCH3(EX, DT, 2);                                             \
        else if (ck) GS_SEG_LAUNCH3(EX, DT, 4);                                                            \
        else if (px_per_lane == 1) GS_BWD_LAUNCH3(EX, DT, 1);                                              \
        else if (px_per_lane == 2) GS_BWD_LAUNCH3(EX, DT, 2);                                              \
        else if (px_per_lane == 4) GS_BWD_LAUNCH3(EX, DT, 4);                                              \
        else                                                                                               \
            GS_LAUNCH((gs::k_rasterize_backward_mixed<EX, DT>), dim3(units), dim3(64), 0, s, W, H, \
                               tiles_x, tiles, long_len, tile_order, gaussian_ids_sorted, block_masks, bins, \
                               pk, bg0, bg1, bg2, bg_dev, final_Ts, final_idx, v_out, v_out_alpha, img_raw, \
                               gacc, gfix, GS_FINAL_FIRST((gs::k_rasterize_backward_mixed<EX, DT>)));      \
    } while (0)
    if (det) {
        if (flags & GS_FLAG_FAST_EXP) GS_BWD_LAUNCH(false, true); else GS_BWD_LAUNCH(true, true);
    } else {
        if (flags & GS_FLAG_FAST_EXP) GS_BWD_LAUNCH(false, false); else GS_BWD_LAUNCH(true, false);
    }
#undef GS_BWD_LAUNCH3
#undef GS_FINAL_FIRST
#undef GS_Q_LAUNCH3
#undef GS_SEG_LAUNCH3
#undef GS_BWD_LAUNCH
    gs::ev_after(s);
    GS_LAUNCH_CHECK();
    if (det) {
        const int64_t n = (int64_t)N * gs::kGradRec;
        GS_LAUNCH(gs::k_fixed_to_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n,
                           reinterpret_cast<const long long *>(gfix), gacc);
        GS_LAUNCH_CHECK();
    }
    if (keep_records) return GS_OK;  // the 64-byte records go straight to gs_gaussian_backward
    GS_LAUNCH(gs::k_unpack_grads, dim3((N + 255) / 256), dim3(256), 0, s, N,
                       reinterpret_cast<const float4 *>(gacc),
                       (flags & GS_FLAG_LOGIT_OPACITY) ? pk : nullptr, v_xy, v_conic, v_colors,
                       v_opacity);
    GS_LAUNCH_CHECK();
    return GS_OK;
}

extern "C" int gs_rasterize_backward(int W, int H, int N, const int32_t *gaussian_ids_sorted,
                                     const uint16_t *block_masks, const int32_t *tile_bins,
                                     const float *packed,
                                     const float *background, const float *final_Ts,
                                     const int32_t *final_idx, const float *v_out,
                                     const float *v_out_alpha, const float *out_img, float *v_xy,
                                     float *v_conic, float *v_colors, float *v_opacity,
                                     void *workspace, size_t workspace_bytes,
                                     const int32_t *list_stats, const int32_t *tile_order,
                                     uint32_t flags, gs_stream_t stream) {
    return gs_rasterize_backward_ckpt(W, H, N, gaussian_ids_sorted, block_masks, tile_bins, packed,
                                      background, final_Ts, final_idx, v_out, v_out_alpha, out_img, v_xy,
                                      v_conic, v_colors, v_opacity, workspace, workspace_bytes, list_stats,
                                      tile_order, flags, nullptr, 0, 0, 0, stream);
}
