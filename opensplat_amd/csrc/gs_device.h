// gs_device.h — device-side helpers shared by the gfx950 kernels of libgsplat_hip.so.
//
// Everything here is written for CDNA4 wave64: cross-lane sums go through DPP row operations
// (no LDS traffic, unlike __shfl which lowers to ds_bpermute), rows are combined with
// v_readlane, and the exponential is a bit-exact re-statement of glibc's expf so that the
// alpha / transmittance thresholds of the compositing loops fall exactly where they fall in
// OpenSplat's CPU rasterizer (rasterizer/gsplat-cpu/gsplat_cpu.cpp:220,:337).
//
// The whole library is compiled with -ffp-contract=off: a*b+c is two roundings unless written
// as fmaf()/fma() explicitly, which is what makes the per-pixel arithmetic reproduce the CPU
// reference (built for baseline x86-64, no FMA) bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gsplat_hip.h"

namespace gs {

constexpr int kWave = 64;

// ---- host-side error plumbing ---------------------------------------------------------------
void set_hip_error(hipError_t e, const char *what);

// RAII ROCTX range around an entry point (gs_api.hip); a no-op unless GSPLAT_ROCTX is set.
struct TraceRange {
    explicit TraceRange(const char *name);
    ~TraceRange();
    TraceRange(const TraceRange &) = delete;
    TraceRange &operator=(const TraceRange &) = delete;

private:
    bool on_;
};
#define GS_TRACE(name) ::gs::TraceRange gs_trace_range_(name)

#define GS_HIP_CHECK(expr)                                                                       \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            ::gs::set_hip_error(_e, #expr);                                                      \
            return GS_ERR_HIP;                                                                   \
        }                                                                                        \
    } while (0)

#define GS_LAUNCH_CHECK() GS_HIP_CHECK(hipGetLastError())

// Every kernel launch of the library goes through GS_LAUNCH: when the calling thread has armed the
// kernel timeline (gs_debug_timeline, include/gsplat_hip.h) a HIP event is recorded on the launch stream
// right before and right after the launch, under the kernel's name — bench.py's per-kernel durations come
// from its OWN run that way (an instrumented pass outside the timed region; a record costs ~5 us of stream
// time).  Unarmed (always, outside that pass) the cost is one thread-local load and a branch.
void timeline_before(hipStream_t s);
void timeline_after(const char *name, hipStream_t s);
#define GS_LAUNCH(kernel, grid, block, lds, stream, ...)                               \
    do {                                                                               \
        ::gs::timeline_before(stream);                                                 \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);             \
        ::gs::timeline_after(#kernel, stream);                                         \
    } while (0)

// Small read-only arguments (background colour, camera position) may be handed over as host OR
// device memory: a device tensor is then read by the kernel itself instead of being copied to the
// host first (which would synchronise the stream).
inline bool on_device(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // plain host memory: not an error
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// ---- DPP cross-lane primitives --------------------------------------------------------------
// dpp_ctrl encodings (gfx9): quad_perm = p0|p1<<2|p2<<4|p3<<6; 0x140 row_mirror;
// 0x141 row_half_mirror.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(
        __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

__device__ __forceinline__ int wave_max_i(int v) {
    v = max(v, dpp_i<0xB1>(v));
    v = max(v, dpp_i<0x4E>(v));
    v = max(v, dpp_i<0x141>(v));
    v = max(v, dpp_i<0x140>(v));
    int s0 = __builtin_amdgcn_readlane(v, 0), s1 = __builtin_amdgcn_readlane(v, 16);
    int s2 = __builtin_amdgcn_readlane(v, 32), s3 = __builtin_amdgcn_readlane(v, 48);
    return max(max(s0, s1), max(s2, s3));
}

// ---- glibc-compatible expf ------------------------------------------------------------------
// Table of 2^(i/32) (bits, minus i<<47) and polynomial of the expf that glibc >= 2.27 ships
// (sysdeps/ieee754/flt-32/e_expf.c, N = 32, computed in double).  Checked exhaustively against
// this image's libm over every float in [-87, 0]: 1 118 699 521 inputs, one mismatch
// (x = -0x1.f8cbb2p+5), none in the range the rasterizer evaluates (sigma in [0, 5.6]).
// Only valid for |x| < 87 (no overflow / subnormal handling) — the callers guarantee it.
static __device__ __constant__ const uint64_t kExp2fTab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

// `tab` points at kExpTabLds entries in LDS: the 32-entry table replicated 8 times, so that the
// index is simply the low BYTE of ki (one SDWA shift) and per-lane reads are one ds_read_b64.
//
// Operation order: the polynomial is evaluated with fused multiply-adds.  glibc ships both a
// contracted (x86-64 FMA ifunc) and an uncontracted build of this routine; an exhaustive host run
// over every float in [-6, 0] (1.09e9 inputs) shows the fused, the unfused and this image's libm
// results are bit-identical there — the compositing kernels only evaluate [-5.55, 0] — and
// tests/test_gpu_ops_and_edges.py::test_device_expf_is_bit_exact_with_host_libm re-checks the
// device against the GPU box's libm.
constexpr int kExpTabLds = 256;

__device__ __forceinline__ void load_exp_table(uint64_t *tab_lds, int tid, int nthreads) {
    for (int i = tid; i < kExpTabLds; i += nthreads) tab_lds[i] = kExp2fTab[i & 31];
}

__device__ __forceinline__ float expf_glibc(float x, const uint64_t *tab) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32.0;
    const double Shift = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0;
    const double C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0;
    const double C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    double z = InvLn2N * (double)x;
    double kd = z + Shift;
    const uint32_t ki = (uint32_t)__double_as_longlong(kd);  // low word: round(z) mod 2^32
    kd -= Shift;
    double r = z - kd;
    // s = 2^(k/32): table bits + (ki << 47) only touches the high word
    const uint64_t t = tab[ki & 0xFFu];
    const uint32_t hi = (uint32_t)(t >> 32) + (ki << 15);
    const double s = __hiloint2double((int)hi, (int)(uint32_t)t);
    // p = fma(C0, r, C1) in the three-address form: the compiler picks the two-address v_fmac_f64 and
    // pays a v_mov_b64 of C1 in front of it on every evaluation
    double p;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "s"(C0), "v"(r), "v"(C1));
    double r2 = r * r;
    double y = fma(C2, r, 1.0);
    y = fma(p, r2, y);
    y = y * s;
    return (float)y;
}

// Same result from the table in constant memory (per-lane global loads): for the backward's rare
// near-threshold re-evaluation, which is not worth 2 KiB of LDS per wave.
__device__ __forceinline__ float expf_glibc_cmem(float x) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32.0;
    const double Shift = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0;
    const double C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0;
    const double C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    double z = InvLn2N * (double)x;
    double kd = z + Shift;
    const uint32_t ki = (uint32_t)__double_as_longlong(kd);
    kd -= Shift;
    double r = z - kd;
    const uint64_t t = kExp2fTab[ki & 31u];
    const uint32_t hi = (uint32_t)(t >> 32) + (ki << 15);
    const double s = __hiloint2double((int)hi, (int)(uint32_t)t);
    double p = fma(C0, r, C1);
    double r2 = r * r;
    double y = fma(C2, r, 1.0);
    y = fma(p, r2, y);
    y = y * s;
    return (float)y;
}

template <bool EXACT>
__device__ __forceinline__ float gs_exp(float x, const uint64_t *tab) {
    if constexpr (EXACT) {
        return expf_glibc(x, tab);
    } else {
        return __expf(x);
    }
}

// ---- per-Gaussian pixel rectangle (CPU oracle semantics) -------------------------------------
// rows/cols the CPU rasterizer visits for a Gaussian centred at (gx, gy) with 2-D covariance
// diagonal (cxx, cyy): gsplat_cpu.cpp:167-168 (3*sqrt(cov)), :201-204 (floor/ceil, +-2, clip).
// Returned as half-open ranges clipped to the image; empty rectangles have x1 <= x0 or y1 <= y0.
struct PixRect {
    int x0, x1, y0, y1;
};

__device__ __forceinline__ int f2i_sat(float v) {
    // float -> int with saturation (the reference's static_cast<int> is UB out of range; clamp
    // first so that far-away Gaussians get a well-defined, clipped rectangle)
    v = fminf(fmaxf(v, -1.0e9f), 1.0e9f);
    return (int)v;
}

__device__ __forceinline__ PixRect pixel_rect(float gx, float gy, float cxx, float cyy, int W,
                                              int H) {
    float sqx = 3.0f * sqrtf(cxx);
    float sqy = 3.0f * sqrtf(cyy);
    PixRect r;
    r.y0 = max(0, f2i_sat(floorf(gy - sqy)) - 2);
    r.y1 = min(H, f2i_sat(ceilf(gy + sqy)) + 2);
    r.x0 = max(0, f2i_sat(floorf(gx - sqx)) - 2);
    r.x1 = min(W, f2i_sat(ceilf(gx + sqx)) + 2);
    // NaN centres/covariances: comparisons with NaN are false -> fminf/fmaxf return the
    // non-NaN operand -> saturated, rectangle ends up empty or clipped; never out of range.
    return r;
}

__device__ __forceinline__ int rect_tiles(const PixRect &r) {
    if (r.x1 <= r.x0 || r.y1 <= r.y0) return 0;
    int tx0 = r.x0 / GS_TILE, tx1 = (r.x1 + GS_TILE - 1) / GS_TILE;
    int ty0 = r.y0 / GS_TILE, ty1 = (r.y1 + GS_TILE - 1) / GS_TILE;
    return (tx1 - tx0) * (ty1 - ty0);
}

// ---- coverage of a tile's sixteen 4x4-pixel blocks by one Gaussian ----------------------------
// block_mask16: bit 4*r + c is set if the block at columns [4c, 4c+3], rows [4r, 4r+3] of the 16x16
// tile with pixel origin (tx0, ty0) may hold a pixel that the compositing kernels composite for this
// packed record — a pixel inside the record's rectangle whose sigma = 0.5 (A u^2 + C v^2) + B u v
// (u, v = pixel - centre) does not exceed sigma_max (= ln(255 opacity) + margin: alpha >= 1/255,
// gsplat_cpu.cpp:213-222).  A SUPERSET of those blocks is always valid (the kernels evaluate every
// pixel of a block they visit); a missing one would change the image, so every step is conservative:
//   * the rectangle alone (block columns x block rows it touches) is the fallback whenever the
//     conic is not a comfortably conditioned ellipse (A, C > 0, det > 1e-3 A C) or the centre is not
//     finite;
//   * otherwise each block row is a slab v in [vlo, vhi] (its pixel rows, clipped to the rectangle
//     and to the ellipse's own v range); the ellipse's u-extent over the slab is attained where the
//     unconstrained extremum v* = -+B u_max / C is clamped into the slab (the right / left boundary
//     u(v) = (-B v +- sqrt(s2 A - det v^2)) / A is concave / convex in v); s2 = 2 sigma_max inflated
//     by 2e-3 (relative) + 1e-3, extents by 1e-3 u_max + 2e-3 pixels: orders of magnitude above the
//     fp32 error of the kernels' sigma for det > 1e-3 A C (det itself is then good to 1e-4), far
//     below a pixel.
// Measured on the benchmark scenes (sampled against per-pixel evaluation): 7.56 blocks per Gaussian
// against 7.54 exactly covered and 8.83 touched by the rectangle at C2; 20.3 / 20.2 / 25.0 at C3; no
// covered block missed (tests/test_gpu_block_masks.py checks that on the device).
struct EllipseRows {
    bool trust;
    float gx, gy, B, det, s2A, vmax, vsr, rA, m;
};
__device__ __forceinline__ EllipseRows ellipse_rows(float gx, float gy, float A, float B, float C,
                                                    uint32_t smax_bits) {
    EllipseRows e;
    const float smax = __uint_as_float(smax_bits & ~1u);
    const float det = A * C - B * B;
    e.trust = smax >= 0.0f && A > 0.0f && C > 0.0f && det > 1.0e-3f * (A * C) && det < 3.0e38f &&
              fabsf(gx) < 1.0e6f && fabsf(gy) < 1.0e6f;
    e.gx = gx; e.gy = gy; e.B = B; e.det = det;
    e.s2A = e.vmax = e.vsr = e.rA = e.m = 0.0f;
    if (e.trust) {
        const float s2 = 2.0f * smax * 1.002f + 1.0e-3f;
        const float rdet = __builtin_amdgcn_rcpf(det) * 1.0001f;   // (1-ulp reciprocal: rounded up)
        const float umax = __builtin_amdgcn_sqrtf(s2 * C * rdet) * 1.0001f;
        e.vmax = __builtin_amdgcn_sqrtf(s2 * A * rdet) * 1.0001f;
        e.vsr = -B * umax * __builtin_amdgcn_rcpf(C);
        e.rA = __builtin_amdgcn_rcpf(A);
        e.s2A = s2 * A;
        e.m = umax * 1.0e-3f + 2.0e-3f;
    }
    return e;
}
// pixel columns [clo, chi] the ellipse can reach on the pixel rows ylo .. yhi (image coordinates,
// inclusive); false if it misses the slab.  Only for e.trust.
__device__ __forceinline__ bool ellipse_row_extent(const EllipseRows &e, int ylo, int yhi, int &clo,
                                                   int &chi) {
    const float vlo = fmaxf((float)ylo - e.gy, -e.vmax);
    const float vhi = fminf((float)yhi - e.gy, e.vmax);
    if (vlo > vhi) return false;   // the slab misses the ellipse
    const float vr = __builtin_amdgcn_fmed3f(e.vsr, vlo, vhi);
    const float vl = __builtin_amdgcn_fmed3f(-e.vsr, vlo, vhi);
    const float dR = fmaxf(e.s2A - e.det * vr * vr, 0.0f);
    const float dL = fmaxf(e.s2A - e.det * vl * vl, 0.0f);
    const float uR = (__builtin_amdgcn_sqrtf(dR) - e.B * vr) * e.rA;
    const float uL = (-__builtin_amdgcn_sqrtf(dL) - e.B * vl) * e.rA;
    // (|u| 1e-4 covers the relative error of the 1-ulp reciprocal / square roots as well)
    clo = f2i_sat(ceilf(e.gx + (uL - (e.m + 1.0e-4f * fabsf(uL)))));
    chi = f2i_sat(floorf(e.gx + (uR + (e.m + 1.0e-4f * fabsf(uR)))));
    return true;
}

__device__ __forceinline__ uint32_t block_mask16(float gx, float gy, float A, float B, float C,
                                                 uint32_t smax_bits, uint32_t rx, uint32_t ry,
                                                 int tx0, int ty0) {
    // rectangle, tile-local, inclusive, clipped to the tile
    int x0 = (int)(rx & 0xFFFFu) - tx0, x1 = (int)(rx >> 16) - 1 - tx0;
    int y0 = (int)(ry & 0xFFFFu) - ty0, y1 = (int)(ry >> 16) - 1 - ty0;
    x0 = max(x0, 0); x1 = min(x1, GS_TILE - 1);
    y0 = max(y0, 0); y1 = min(y1, GS_TILE - 1);
    if (x1 < x0 || y1 < y0) return 0u;
    const uint32_t cols_rect = ((2u << (x1 >> 2)) - 1u) & ~((1u << (x0 >> 2)) - 1u);
    const EllipseRows e = ellipse_rows(gx, gy, A, B, C, smax_bits);
    uint32_t mask = 0u;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int ylo = max(4 * r, y0), yhi = min(4 * r + 3, y1);
        if (ylo > yhi) continue;
        uint32_t cm = cols_rect;
        if (e.trust) {
            int clo, chi;
            if (!ellipse_row_extent(e, ty0 + ylo, ty0 + yhi, clo, chi)) continue;
            clo = max(clo - tx0, x0);
            chi = min(chi - tx0, x1);
            if (clo > chi) continue;
            cm = ((2u << (chi >> 2)) - 1u) & ~((1u << (clo >> 2)) - 1u);
        }
        mask |= cm << (4 * r);
    }
    return mask;
}

// The same coverage computed ONCE PER GAUSSIAN instead of once per (tile, Gaussian): a table of the
// block-column extent of each 4-pixel block row the Gaussian's rectangle spans —
//   base = first block row | first block column << 16      (image / 4; kNoRowTable: no table)
//   r0..r3 = sixteen rows x one byte: (first column - base's) | (last column - base's) << 4;
//                first > last (0x0F) = row not reached
// — from which the mask of any tile is assembled with a few integer operations (mask_from_rows): the
// per-entry work of the binning drops from ~300 to ~90 VALU instructions.  Rectangles spanning more
// than 16 block rows or 16 block columns (64 x 64 pixels) get no table; their entries fall back to
// block_mask16.
constexpr uint32_t kNoRowTable = 0xFFFFFFFFu;
constexpr int kRowTableRows = 16;
struct RowTable {   // (named words, not an array: an array member sends the whole table to scratch memory)
    uint32_t base;
    uint32_t r0, r1, r2, r3;
};
__device__ __forceinline__ RowTable block_rows_table(float gx, float gy, float A, float B, float C,
                                                     uint32_t smax_bits, uint32_t rx, uint32_t ry) {
    const int x0 = (int)(rx & 0xFFFFu), x1 = (int)(rx >> 16) - 1;   // inclusive
    const int y0 = (int)(ry & 0xFFFFu), y1 = (int)(ry >> 16) - 1;
    RowTable w = {kNoRowTable, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu};
    if (x1 < x0 || y1 < y0) return w;
    const int br0 = y0 >> 2, br1 = y1 >> 2, bc0 = x0 >> 2, bc1 = x1 >> 2;
    if (br1 - br0 >= kRowTableRows || bc1 - bc0 >= 16) return w;
    const EllipseRows e = ellipse_rows(gx, gy, A, B, C, smax_bits);
    uint32_t rows0 = 0x0F0F0F0Fu, rows1 = 0x0F0F0F0Fu, rows2 = 0x0F0F0F0Fu, rows3 = 0x0F0F0F0Fu;
    for (int k = 0; k <= br1 - br0; k++) {   // (per-lane trip count: 2.8 on average at C2)
        const int ylo = max(4 * (br0 + k), y0), yhi = min(4 * (br0 + k) + 3, y1);
        int clo = x0, chi = x1;
        bool hit = true;
        if (e.trust) {
            hit = ellipse_row_extent(e, ylo, yhi, clo, chi);
            clo = max(clo, x0);
            chi = min(chi, x1);
            hit = hit && clo <= chi;
        }
        if (hit) {
            const uint32_t byte = (uint32_t)((clo >> 2) - bc0) | ((uint32_t)((chi >> 2) - bc0) << 4);
            const int sh = (k & 3) * 8;
            const uint32_t clr = ~(0xFFu << sh), set = byte << sh;
            if ((k >> 2) == 0) rows0 = (rows0 & clr) | set;
            else if ((k >> 2) == 1) rows1 = (rows1 & clr) | set;
            else if ((k >> 2) == 2) rows2 = (rows2 & clr) | set;
            else rows3 = (rows3 & clr) | set;
        }
    }
    w.base = (uint32_t)br0 | ((uint32_t)bc0 << 16);
    w.r0 = rows0; w.r1 = rows1; w.r2 = rows2; w.r3 = rows3;
    return w;
}
// mask of the tile at tile coordinates (tx, ty) from a row table; only for w.base != kNoRowTable
__device__ __forceinline__ uint32_t mask_from_rows(const RowTable w, int tx, int ty) {
    // (by value, words copied to scalars: a select among the members of a struct behind a reference is a
    // select among ADDRESSES, which pins the table in scratch memory)
    const uint32_t r0 = w.r0, r1 = w.r1, r2 = w.r2, r3 = w.r3;
    const int br0 = (int)(w.base & 0xFFFFu), bc0 = (int)(w.base >> 16);
    const int dc = bc0 - 4 * tx;   // table columns -> tile-local block columns
    uint32_t mask = 0u;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int k = 4 * ty + r - br0;
        const int q = k >> 2;
        const uint32_t word = q == 0 ? r0 : (q == 1 ? r1 : (q == 2 ? r2 : r3));
        const uint32_t byte = (k >= 0 && k < kRowTableRows) ? ((word >> ((k & 3) * 8)) & 0xFFu) : 0x0Fu;
        const int lo = max((int)(byte & 15u) + dc, 0), hi = min((int)(byte >> 4) + dc, 3);
        // (an unreached row has first = 15 > last = 0; clipping to the tile keeps first > last)
        const uint32_t cm = ((2u << max(hi, 0)) - 1u) & ~((1u << lo) - 1u);
        mask |= ((byte & 15u) <= (byte >> 4) && lo <= hi) ? (cm << (4 * r)) : 0u;
    }
    return mask;
}

// XCD-aware block -> tile mapping: consecutive blocks round-robin over the 8 XCDs, so give each
// XCD a contiguous band of tiles (neighbouring tiles share Gaussians -> shared L2 lines).
__device__ __forceinline__ int xcd_swizzle(int block, int num_blocks) {
    constexpr int kXcd = 8;
    int per = num_blocks / kXcd;  // tiles in the evenly divisible part
    int body = per * kXcd;
    if (block >= body) return block;  // ragged tail: identity
    return (block % kXcd) * per + (block / kXcd);
}

}  // namespace gs
