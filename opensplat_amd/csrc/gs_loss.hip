// gs_loss.hip — Model::mainLoss (model.cpp:780-784) and its backward, fused: L1 (model.cpp:54-56)
// + SSIM with the reference's 11x11 window (ssim.cpp:7-45), value and gradient w.r.t. the rendered
// image in two tiled kernels + a one-workgroup finalisation.  C ABI: include/gsplat_train.h.
//
// The reference builds the loss from six grouped conv2d calls and ~25 element-wise torch ops, each
// a full pass over [1,3,H,W] tensors, then autograd replays them backwards.  Here:
//
//   k_ssim_maps   one workgroup per 32x22 output tile: stages the (32+10)x(22+10) halo of both
//                 images (all three channels, exactly as they lie in the HWC rows) in LDS, runs
//                 the separable window over the four quantities {x, y, xx + yy, xy} (only the SUM of
//                 the second moments enters the map and its derivatives) — horizontal pass LDS->LDS
//                 with four outputs per thread, vertical pass LDS->registers with three outputs per
//                 thread — evaluates the SSIM map and its three partial derivatives
//                 (w.r.t. mu2, E[yy], E[xy]) and writes those planar; per-workgroup partial sums
//                 of the SSIM map and of |gt - rendered|;
//   k_ssim_grad   the transposed (flipped-window) convolution of the three derivative maps, again
//                 separable through LDS, combined with d mu2/dy = 1, d yy/dy = 2y, d xy/dy = x and
//                 the L1 sign term; output staged in LDS so that the HWC rows are written coalesced;
//   k_loss_finalize  sums the partials in fp64 and writes {mainLoss, l1, ssim}.
//
// Algorithmic bytes per pixel: 24 read + 108 written by the first kernel, 108 + 24 read + 12
// written by the second.  Measured on MI355X the first is VALU-bound (the window's
// pair structure, conv_pairs, cuts the multiply-adds per output from eleven to six), the second
// latency-bound (next channel's halo prefetched into registers); tiles are numbered so that each
// XCD works on a contiguous band of the image and halo re-reads hit its L2.
#include <math.h>

#include "gs_device.h"
#include "../../include/gsplat_train.h"

namespace gs {

constexpr int kWin = GS_SSIM_WINDOW, kRad = kWin / 2;
constexpr int kTW = 32, kTH = 22;                        // output tile
constexpr int kHW = kTW + 2 * kRad, kHH = kTH + 2 * kRad;  // halo: 42 x 32
constexpr int kHPitch = kTW + 1;                         // 33
constexpr int kMapPitch = kHW + 1;                       // 43
constexpr int kOutPitch = 3 * kTW + 1;                   // 97
constexpr int kLossThreads = 256;
static_assert(kHH * (kTW / 4) == kLossThreads, "horizontal pass: one item per thread");
static_assert(kTH <= 3 * (kLossThreads / kTW), "vertical pass: three rows per thread");
static_assert(kHH % 4 == 0 && kTH % 2 == 0 && 3 * kHW <= 128 && kHW <= 64 && kLossThreads == 256,
              "row-pair / row-quad loads");

// The reference's window has the shape {a, b,b, c,c, d,d, e,e, f,f} (floor((i - 11) / 2) takes every
// value but -6 twice, ssim.cpp:42): one single weight and five PAIR weights.  A pass then needs one
// add per input (pair sums, shared by all outputs of a thread) + six multiply-adds per output
// instead of eleven.
struct Window { float w0; float p[5]; };

// out[j] = sum_d w[d] in[j + d]  (FLIP: the transposed pass, sum_d w[10 - d] in[j + d])
template <bool FLIP, int NOUT>
static __device__ __forceinline__ void conv_pairs(const Window &w, const float (&in)[NOUT + 10],
                                                  float (&out)[NOUT]) {
    float ps[NOUT + 9];
#pragma unroll
    for (int k = 0; k < NOUT + 9; k++) ps[k] = in[k] + in[k + 1];
#pragma unroll
    for (int j = 0; j < NOUT; j++) {
        float s = w.w0 * in[FLIP ? j + 10 : j];
#pragma unroll
        for (int m = 0; m < 5; m++)
            s = fmaf(FLIP ? w.p[4 - m] : w.p[m], ps[FLIP ? j + 2 * m : j + 1 + 2 * m], s);
        out[j] = s;
    }
}

// ssim.cpp:39-45 on the host, normalised with libtorch's summation order (see gs_ssim_window).
static void host_window(float *g) {
    const float sigma = 1.5f;
    for (int i = 0; i < kWin; i++) {
        const float a = floorf((float)(i - kWin) / 2.0f);
        g[i] = expf(-(powf(a, 2.0f)) / (2.0f * sigma * sigma));
    }
    float acc[8];
    for (int j = 0; j < 8; j++) acc[j] = g[j];
    for (int j = 8; j < kWin; j++) acc[j - 8] += g[j];
    for (int j = 0; j < 4; j++) acc[j] += acc[j + 4];
    for (int j = 0; j < 2; j++) acc[j] += acc[j + 2];
    const float sum = acc[0] + acc[1];
    for (int i = 0; i < kWin; i++) g[i] = g[i] / sum;
}

static bool make_window(Window &w) {
    float g[kWin];
    host_window(g);
    w.w0 = g[0];
    for (int m = 0; m < 5; m++) {
        if (g[1 + 2 * m] != g[2 + 2 * m]) return false;  // cannot happen: same exponent argument
        w.p[m] = g[1 + 2 * m];
    }
    return true;
}

// Tiles are numbered so that each XCD (blocks round-robin over the 8 XCDs) works on a contiguous
// band of the image: the halos neighbouring tiles share are then hits in that XCD's L2.
static __device__ __forceinline__ void tile_origin(int tiles_x, int &tile, int &x0, int &y0) {
    tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    x0 = tx * kTW;
    y0 = ty * kTH;
}

// Sum of `v` over the workgroup (256 threads); result valid in thread 0.
static __device__ __forceinline__ float block_sum(float v, float *red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    float r = 0.0f;
    if (threadIdx.x == 0) r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kLossThreads)
k_ssim_maps(int W, int H, int tiles_x, Window win, const float *__restrict__ rendered,
            const float *__restrict__ gt, float *__restrict__ maps /* [3 maps][3 ch][H][W] */,
            float2 *__restrict__ partial /* {ssim sum, l1 sum} per workgroup */) {
    __shared__ float rx[kHH][kMapPitch];  // gt       (img1, ssim.cpp:8), ONE channel at a time
    __shared__ float ry[kHH][kMapPitch];  // rendered (img2, ssim.cpp:9)
    __shared__ float hb[4][kHH][kHPitch];  // window sums of x, y, xx + yy, xy
    __shared__ float red[4];
    const int tid = threadIdx.x;
    int tile, x0, y0;
    tile_origin(tiles_x, tile, x0, y0);
    const size_t P = (size_t)W * H;

    // Halo of ONE channel at a time (28 KB of LDS instead of 49: five workgroups per CU instead of
    // three — the kernel is latency-bound): lane -> halo column, four rows per iteration; the
    // stride-3 reads of the three channel passes touch the same lines (L1 / L2 hits), and the next
    // channel's values are fetched into registers while this channel is convolved.
    const int lc = tid & 63, lrr = tid >> 6;
    const int lgx = x0 - kRad + lc;
    const bool lcol_ok = lc < kHW && lgx >= 0 && lgx < W;
    float pa[kHH / 4], pb[kHH / 4];
    auto fetch = [&](int ch) {
#pragma unroll
        for (int it = 0; it < kHH / 4; it++) {
            const int gy = y0 - kRad + 4 * it + lrr;
            pa[it] = 0.0f;
            pb[it] = 0.0f;
            if (lcol_ok && gy >= 0 && gy < H) {
                const size_t o = ((size_t)gy * W + lgx) * 3 + ch;
                pa[it] = gt[o];
                pb[it] = rendered[o];
            }
        }
    };
    fetch(0);

    float ssim_sum = 0.0f, l1_sum = 0.0f;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;  // ssim.cpp:26-27
    for (int ch = 0; ch < 3; ch++) {
        if (lc < kHW) {
#pragma unroll
            for (int it = 0; it < kHH / 4; it++) {
                rx[4 * it + lrr][lc] = pa[it];
                ry[4 * it + lrr][lc] = pb[it];
            }
        }
        __syncthreads();  // (also: the previous channel's vertical pass is done with hb)
        if (ch < 2) fetch(ch + 1);
        {   // horizontal pass: thread = (halo row, group of four output columns)
            const int row = tid % kHH, c0 = 4 * (tid / kHH);
            float xv[kWin + 3], yv[kWin + 3], in[kWin + 3], o[4];
#pragma unroll
            for (int k = 0; k < kWin + 3; k++) {
                xv[k] = rx[row][c0 + k];
                yv[k] = ry[row][c0 + k];
            }
            conv_pairs<false, 4>(win, xv, o);
#pragma unroll
            for (int j = 0; j < 4; j++) hb[0][row][c0 + j] = o[j];
            conv_pairs<false, 4>(win, yv, o);
#pragma unroll
            for (int j = 0; j < 4; j++) hb[1][row][c0 + j] = o[j];
            // only the SUM of the two second moments enters the SSIM map and its derivatives
#pragma unroll
            for (int k = 0; k < kWin + 3; k++) in[k] = fmaf(xv[k], xv[k], yv[k] * yv[k]);
            conv_pairs<false, 4>(win, in, o);
#pragma unroll
            for (int j = 0; j < 4; j++) hb[2][row][c0 + j] = o[j];
#pragma unroll
            for (int k = 0; k < kWin + 3; k++) in[k] = xv[k] * yv[k];
            conv_pairs<false, 4>(win, in, o);
#pragma unroll
            for (int j = 0; j < 4; j++) hb[3][row][c0 + j] = o[j];
        }
        __syncthreads();
        {   // vertical pass: thread = (column, group of three output rows)
            const int c = tid % kTW, r0 = 3 * (tid / kTW);
            float acc[4][3];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float in[kWin + 2];
#pragma unroll
                for (int k = 0; k < kWin + 2; k++) in[k] = (r0 + k < kHH) ? hb[q][r0 + k][c] : 0.0f;
                conv_pairs<false, 3>(win, in, acc[q]);
            }
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int r = r0 + j, gx = x0 + c, gy = y0 + r;
                if (r < kTH && gx < W && gy < H) {
                    const float m1 = acc[0][j], m2 = acc[1][j];
                    const float mu1Sq = m1 * m1, mu2Sq = m2 * m2, mu1mu2 = m1 * m2;
                    const float s12 = acc[3][j] - mu1mu2;
                    const float A1 = 2.0f * mu1mu2 + C1, A2 = 2.0f * s12 + C2;
                    // sigma1Sq + sigma2Sq = (E[xx] + E[yy]) - mu1^2 - mu2^2
                    const float B1 = mu1Sq + mu2Sq + C1, B2 = (acc[2][j] - mu1Sq - mu2Sq) + C2;
                    // reciprocals: hardware estimate + one Newton step (|rel. error| < 2^-22; four IEEE
                    // divisions per pixel-channel were 15 % of this VALU-bound kernel's instructions)
                    float r1 = __builtin_amdgcn_rcpf(B1), r2 = __builtin_amdgcn_rcpf(B2);
                    r1 = fmaf(fmaf(-B1, r1, 1.0f), r1, r1);
                    r2 = fmaf(fmaf(-B2, r2, 1.0f), r2, r2);
                    const float invB = r1 * r2;
                    const float S = (A1 * A2) * invB;  // ssim.cpp:29
                    ssim_sum += S;
                    const float xc = rx[r + kRad][c + kRad];
                    const float yc = ry[r + kRad][c + kRad];
                    l1_sum += fabsf(xc - yc);  // model.cpp:55
                    // dS/dmu2, dS/dE[yy], dS/dE[xy] with the other window sums held fixed
                    const size_t o = (size_t)ch * P + (size_t)gy * W + gx;
                    maps[o] = 2.0f * m1 * (A2 - A1) * invB - 2.0f * m2 * S * (r1 - r2);
                    maps[3 * P + o] = -S * r2;
                    maps[6 * P + o] = 2.0f * A1 * invB;
                }
            }
        }
        __syncthreads();
    }
    const float a = block_sum(ssim_sum, red);
    const float b = block_sum(l1_sum, red);
    if (tid == 0) partial[tile] = make_float2(a, b);
}

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kLossThreads)
k_ssim_grad(int W, int H, int tiles_x, Window win, const float *__restrict__ rendered,
            const float *__restrict__ gt, const float *__restrict__ maps, float c_l1, float c_ssim,
            float *__restrict__ v_rendered) {
    __shared__ float raw[3][kHH][kMapPitch];
    __shared__ float hb[3][kHH][kHPitch];
    __shared__ float outb[kTH][kOutPitch];  // result tile, written back as coalesced HWC rows
    const int tid = threadIdx.x;
    int tile, x0, y0;
    tile_origin(tiles_x, tile, x0, y0);
    const size_t P = (size_t)W * H;

    const int oc = tid & 127, orr = tid >> 7;
    const bool ocol_ok = oc < 3 * kTW && x0 * 3 + oc < 3 * W;
    // halo of the three derivative maps: lane -> (map, column), two rows per iteration
    const int lq = (tid & 127) / kHW, lc = (tid & 127) - lq * kHW, lrr = tid >> 7;
    const int lgx = x0 - kRad + lc;
    const bool lcol_ok = lq < 3 && lgx >= 0 && lgx < W;

    // The next channel's halo is fetched into registers while this channel is being convolved:
    // the kernel is latency-bound (three dependent global -> LDS -> barrier rounds per tile).
    float pre[kHH / 2];
    auto fetch = [&](int ch) {
#pragma unroll
        for (int it = 0; it < kHH / 2; it++) {
            const int gy = y0 - kRad + 2 * it + lrr;
            pre[it] = 0.0f;
            if (lcol_ok && gy >= 0 && gy < H)
                pre[it] = maps[(size_t)(3 * lq + ch) * P + (size_t)gy * W + lgx];
        }
    };
    fetch(0);
    for (int ch = 0; ch < 3; ch++) {
        if (lq < 3) {
#pragma unroll
            for (int it = 0; it < kHH / 2; it++) raw[lq][2 * it + lrr][lc] = pre[it];
        }
        // (also orders the previous channel's vertical reads of hb before this channel's writes)
        __syncthreads();
        if (ch < 2) fetch(ch + 1);
        {   // horizontal pass with the flipped window (transposed convolution)
            const int row = tid % kHH, c0 = 4 * (tid / kHH);
#pragma unroll
            for (int q = 0; q < 3; q++) {
                float in[kWin + 3], o[4];
#pragma unroll
                for (int k = 0; k < kWin + 3; k++) in[k] = raw[q][row][c0 + k];
                conv_pairs<true, 4>(win, in, o);
#pragma unroll
                for (int j = 0; j < 4; j++) hb[q][row][c0 + j] = o[j];
            }
        }
        __syncthreads();
        {
            const int c = tid % kTW, r0 = 3 * (tid / kTW);
            float acc[3][3];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                float in[kWin + 2];
#pragma unroll
                for (int k = 0; k < kWin + 2; k++) in[k] = (r0 + k < kHH) ? hb[q][r0 + k][c] : 0.0f;
                conv_pairs<true, 3>(win, in, acc[q]);
            }
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int r = r0 + j, gx = x0 + c, gy = y0 + r;
                if (r < kTH) {
                    float v = 0.0f;
                    if (gx < W && gy < H) {
                        // (stride-3 reads of the two images: the three channel passes of a wave
                        // touch the same lines, so two of three are L1 / L2 hits; staging the tile
                        // in LDS instead costs 17 KB and a workgroup of occupancy — measured slower)
                        const size_t o = ((size_t)gy * W + gx) * 3 + ch;
                        const float xc = gt[o], yc = rendered[o];
                        // d ssim_map-sum / d y[q]
                        const float dS = acc[0][j] + 2.0f * yc * acc[1][j] + xc * acc[2][j];
                        // d|gt - r|/dr = -sign(gt - r), sign(0) = 0 (torch::abs backward)
                        const float d = xc - yc;
                        const float sg = d > 0.0f ? -1.0f : (d < 0.0f ? 1.0f : 0.0f);
                        v = c_l1 * sg - c_ssim * dS;
                    }
                    outb[r][c * 3 + ch] = v;
                }
            }
        }
    }
    __syncthreads();
    // HWC rows of the tile: 96 contiguous floats each
#pragma unroll
    for (int it = 0; it < kTH / 2; it++) {
        const int r = 2 * it + orr, gy = y0 + r;
        if (ocol_ok && gy < H) v_rendered[(size_t)gy * W * 3 + x0 * 3 + oc] = outb[r][oc];
    }
}

// --ssim-weight 0: L1 only, one streaming pass
__global__ void __launch_bounds__(kLossThreads)
k_l1_loss(int64_t n, const float *__restrict__ rendered, const float *__restrict__ gt, float c_l1,
          float *__restrict__ v_rendered, float2 *__restrict__ partial) {
    __shared__ float red[4];
    float sum = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float d = gt[i] - rendered[i];
        sum += fabsf(d);
        if (v_rendered) v_rendered[i] = c_l1 * (d > 0.0f ? -1.0f : (d < 0.0f ? 1.0f : 0.0f));
    }
    const float a = block_sum(sum, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = make_float2(0.0f, a);
}

__global__ void __launch_bounds__(kLossThreads)
k_loss_finalize(int nparts, const float2 *__restrict__ partial, double inv, float ssim_weight,
                int have_ssim, float *__restrict__ loss) {
    __shared__ double sa[kLossThreads], sb[kLossThreads];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nparts; i += kLossThreads) {
        a += (double)partial[i].x;
        b += (double)partial[i].y;
    }
    sa[threadIdx.x] = a;
    sb[threadIdx.x] = b;
    __syncthreads();
    for (int off = kLossThreads / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            sa[threadIdx.x] += sa[threadIdx.x + off];
            sb[threadIdx.x] += sb[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double ssim = have_ssim ? sa[0] * inv : 0.0, l1 = sb[0] * inv;
        const double w = (double)ssim_weight;
        loss[0] = (float)((1.0 - w) * l1 + (have_ssim ? w * (1.0 - ssim) : 0.0));
        loss[1] = (float)l1;
        loss[2] = (float)ssim;
    }
}

struct LossLayout {
    int tiles_x, tiles_y, nparts;
    size_t maps_off, partial_off, total;
    LossLayout(int W, int H) {
        tiles_x = (W + kTW - 1) / kTW;
        tiles_y = (H + kTH - 1) / kTH;
        nparts = tiles_x * tiles_y > 1024 ? tiles_x * tiles_y : 1024;
        maps_off = 0;
        partial_off = (((size_t)9 * W * H * sizeof(float)) + 255) & ~(size_t)255;
        total = partial_off + (size_t)nparts * sizeof(float2);
    }
};

}  // namespace gs

extern "C" int gs_ssim_window(float *g) {
    if (!g) return GS_ERR_INVALID_ARGUMENT;
    gs::host_window(g);
    return GS_OK;
}

extern "C" size_t gs_loss_workspace_bytes(int W, int H) {
    if (W <= 0 || H <= 0) return 0;
    return gs::LossLayout(W, H).total;
}

extern "C" int gs_main_loss(int W, int H, const float *rendered, const float *gt, float ssim_weight,
                            float grad_scale, float *loss, float *v_rendered, void *workspace,
                            size_t workspace_bytes, gs_stream_t stream) {
    using namespace gs;
    if (W <= 0 || H <= 0 || !rendered || !gt || !loss || !workspace) return GS_ERR_INVALID_ARGUMENT;
    if (W > 65535 || H > 65535) return GS_ERR_UNSUPPORTED;
    const LossLayout L(W, H);
    if (workspace_bytes < L.total) return GS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    float *maps = (float *)(ws + L.maps_off);
    float2 *partial = (float2 *)(ws + L.partial_off);
    const double inv = 1.0 / (3.0 * (double)W * (double)H);  // .mean() over [1,3,H,W] / [H,W,3]
    const float c_l1 = (float)((double)grad_scale * (1.0 - (double)ssim_weight) * inv);
    const float c_ssim = (float)((double)grad_scale * (double)ssim_weight * inv);
    int nparts;
    if (ssim_weight != 0.0f) {
        Window win;
        if (!make_window(win)) return GS_ERR_UNSUPPORTED;
        nparts = L.tiles_x * L.tiles_y;
        const dim3 grid(nparts);
        GS_LAUNCH(k_ssim_maps, grid, dim3(kLossThreads), 0, s, W, H, L.tiles_x, win,
                           rendered, gt, maps, partial);
        GS_LAUNCH_CHECK();
        if (v_rendered) {
            GS_LAUNCH(k_ssim_grad, grid, dim3(kLossThreads), 0, s, W, H, L.tiles_x, win,
                               rendered, gt, maps, c_l1, c_ssim, v_rendered);
            GS_LAUNCH_CHECK();
        }
    } else {
        nparts = 1024;
        GS_LAUNCH(k_l1_loss, dim3(nparts), dim3(kLossThreads), 0, s,
                           (int64_t)3 * W * H, rendered, gt, c_l1, v_rendered, partial);
        GS_LAUNCH_CHECK();
    }
    GS_LAUNCH(k_loss_finalize, dim3(1), dim3(kLossThreads), 0, s, nparts, partial, inv,
                       ssim_weight, ssim_weight != 0.0f ? 1 : 0, loss);
    GS_LAUNCH_CHECK();
    return GS_OK;
}
