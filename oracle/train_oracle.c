/* TEST INFRASTRUCTURE ONLY — never linked into, imported by, or called from the product path.
 *
 * Plain-C restatement of the training-step pieces of SURVEY.md §8 row f2:
 *   SSIM::gaussian / createWindow / eval     ssim.cpp:7-45
 *   l1                                        model.cpp:54-56
 *   Model::mainLoss                           model.cpp:780-784
 *   torch::optim::Adam::step                  libtorch 2.10 (third-party, not in the reference tree;
 *                                             model.cpp:61-66 constructs it with AdamOptions(lr) only:
 *                                             betas (0.9, 0.999), eps 1e-8, no weight decay, no
 *                                             amsgrad); algorithm restated from its documentation
 *   OptimScheduler::getLearningRate           optim_scheduler.cpp:4-7
 * Pinned against the reference's own objects (oracle/_ref, ref_train_shim.cpp) by
 * tests/test_train_oracle.py and against tests/golden/train_*.npz.
 *
 * The loss gradient is the hand-derived reverse mode of the same op graph libtorch autograd
 * differentiates (opensplat.cpp:160-161).  Convolutions are the direct 2-D, zero-padded
 * cross-correlations conv2d computes, with the reference's (asymmetric!) window.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define WS 11 /* model.hpp:32: ssim(11, 3) */

/* ssim.cpp:39-45: gauss[i] = exp(-(floor((i - windowSize) / 2))^2 / (2 sigma^2)), normalised.
 * NB the argument is floor((i - 11) / 2) = -6,-5,-5,-4,-4,-3,-3,-2,-2,-1,-1 — not the symmetric
 * (i - 5) of the original pytorch-ssim; the window's mass sits at its far end.  Reproduced. */
void orc_ssim_gaussian(float sigma, float *g /* [11] */) {
    for (int i = 0; i < WS; i++) {
        const float a = floorf((float)(i - WS) / 2.0f);
        g[i] = expf(-(powf(a, 2.0f)) / (2.0f * sigma * sigma));
    }
    /* gauss.sum(): ATen's vectorised float sum — eight lanes, the three-element tail added to
     * lanes 0..2, then a halving tree.  (Pinned by experiment: a plain left-to-right sum is one
     * ulp off and changes nine of the eleven weights; tests/test_train_oracle.py.) */
    float acc[8];
    for (int j = 0; j < 8; j++) acc[j] = g[j];
    for (int j = 8; j < WS; j++) acc[j - 8] += g[j];
    for (int j = 0; j < 4; j++) acc[j] += acc[j + 4];
    for (int j = 0; j < 2; j++) acc[j] += acc[j + 2];
    const float sum = acc[0] + acc[1];
    for (int i = 0; i < WS; i++) g[i] = g[i] / sum;
}

/* ssim.cpp:33-37: _2DWindow = _1DWindow.mm(_1DWindow.t()) */
void orc_ssim_window(float *w2 /* [11*11] */) {
    float g[WS];
    orc_ssim_gaussian(1.5f, g);
    for (int i = 0; i < WS; i++)
        for (int j = 0; j < WS; j++) w2[i * WS + j] = g[i] * g[j];
}

/* conv2d(img, window, padding = 5, groups = channel) of one HWC channel, ssim.cpp:16-24 */
static void conv_channel(int W, int H, const float *src /* [H,W] */, const float *w2, float *dst) {
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float acc = 0.0f;
            for (int i = 0; i < WS; i++) {
                const int yy = y + i - WS / 2;
                if (yy < 0 || yy >= H) continue;
                for (int j = 0; j < WS; j++) {
                    const int xx = x + j - WS / 2;
                    if (xx < 0 || xx >= W) continue;
                    acc += w2[i * WS + j] * src[(size_t)yy * W + xx];
                }
            }
            dst[(size_t)y * W + x] = acc;
        }
}

/* transpose of conv_channel: dst[q] += sum_p g[p] w2[q - p + 5] */
static void conv_channel_transposed(int W, int H, const float *g, const float *w2, float *dst) {
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float acc = 0.0f;
            for (int i = 0; i < WS; i++) {
                const int py = y - i + WS / 2;
                if (py < 0 || py >= H) continue;
                for (int j = 0; j < WS; j++) {
                    const int px = x - j + WS / 2;
                    if (px < 0 || px >= W) continue;
                    acc += w2[i * WS + j] * g[(size_t)py * W + px];
                }
            }
            dst[(size_t)y * W + x] = acc;
        }
}

/* Model::mainLoss (model.cpp:780-784) and its gradient w.r.t. the rendered image.
 * rendered, gt: [H,W,3].  loss3 = { mainLoss, l1, ssim }.  v_rendered may be NULL. */
int orc_main_loss(int W, int H, const float *rendered, const float *gt, float ssim_weight,
                  float *loss3, float *v_rendered) {
    const size_t P = (size_t)W * H;
    float w2[WS * WS];
    orc_ssim_window(w2);
    float *buf = (float *)malloc(sizeof(float) * P * 13);
    if (!buf) return -1;
    float *x = buf, *y = x + P, *t = y + P, *mu1 = t + P, *mu2 = mu1 + P, *e11 = mu2 + P,
          *e22 = e11 + P, *e12 = e22 + P, *gm = e12 + P, *g22 = gm + P, *g12 = g22 + P,
          *c0 = g12 + P, *c1 = c0 + P;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f; /* ssim.cpp:26-27 */
    double ssim_sum = 0.0, l1_sum = 0.0;
    const double inv = 1.0 / (3.0 * (double)P);
    for (int ch = 0; ch < 3; ch++) {
        /* img1 = gt, img2 = rendered (ssim.cpp:8-9) */
        for (size_t p = 0; p < P; p++) {
            x[p] = gt[3 * p + ch];
            y[p] = rendered[3 * p + ch];
            l1_sum += fabs((double)(x[p] - y[p])); /* model.cpp:55 */
        }
        conv_channel(W, H, x, w2, mu1);
        conv_channel(W, H, y, w2, mu2);
        for (size_t p = 0; p < P; p++) t[p] = x[p] * x[p];
        conv_channel(W, H, t, w2, e11);
        for (size_t p = 0; p < P; p++) t[p] = y[p] * y[p];
        conv_channel(W, H, t, w2, e22);
        for (size_t p = 0; p < P; p++) t[p] = x[p] * y[p];
        conv_channel(W, H, t, w2, e12);
        for (size_t p = 0; p < P; p++) {
            const float m1 = mu1[p], m2 = mu2[p];
            const float mu1Sq = m1 * m1, mu2Sq = m2 * m2, mu1mu2 = m1 * m2;
            const float s1 = e11[p] - mu1Sq, s2 = e22[p] - mu2Sq, s12 = e12[p] - mu1mu2;
            const float A1 = 2.0f * mu1mu2 + C1, A2 = 2.0f * s12 + C2;
            const float B1 = mu1Sq + mu2Sq + C1, B2 = s1 + s2 + C2;
            const float S = (A1 * A2) / (B1 * B2); /* ssim.cpp:29 */
            ssim_sum += (double)S;
            /* d S / d {mu2, E[y^2], E[xy]} with the other conv outputs held fixed */
            const float invB = 1.0f / (B1 * B2);
            gm[p] = 2.0f * m1 * (A2 - A1) * invB - 2.0f * m2 * S * (1.0f / B1 - 1.0f / B2);
            g22[p] = -S / B2;
            g12[p] = 2.0f * A1 * invB;
        }
        if (v_rendered) {
            conv_channel_transposed(W, H, gm, w2, c0);
            conv_channel_transposed(W, H, g22, w2, c1);
            conv_channel_transposed(W, H, g12, w2, t);
            for (size_t p = 0; p < P; p++) {
                const double dS = (double)c0[p] + 2.0 * (double)y[p] * (double)c1[p] +
                                  (double)x[p] * (double)t[p];
                /* d|gt - r|/dr = -sign(gt - r), sign(0) = 0 (torch::abs backward) */
                const float d = x[p] - y[p];
                const double sg = d > 0.0f ? -1.0 : (d < 0.0f ? 1.0 : 0.0);
                v_rendered[3 * p + ch] =
                    (float)((1.0 - (double)ssim_weight) * sg * inv - (double)ssim_weight * dS * inv);
            }
        }
    }
    const double ssim = ssim_sum * inv, l1 = l1_sum * inv;
    loss3[0] = (float)((1.0 - (double)ssim_weight) * l1 + (double)ssim_weight * (1.0 - ssim));
    loss3[1] = (float)l1;
    loss3[2] = (float)ssim;
    free(buf);
    return 0;
}

/* One torch::optim::Adam step (defaults as constructed at model.cpp:61-66), `step` 1-based:
 *   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g g;
 *   denom = sqrt(v) / sqrt(1 - b2^step) + eps;  p += (-(lr / (1 - b1^step)) m) / denom
 * Both moments come out bit-identical to libtorch's.  The parameter does in all but ~0.03 % of
 * the elements per step, where it is one ulp off: libtorch's vectorised CPU sqrt (AVX-512 build in
 * this image) is not correctly rounded (0.66 % of random inputs differ from IEEE sqrt, measured);
 * sqrtf here is.
 * bias corrections in double as libtorch computes them, tensor arithmetic in float. */
void orc_adam_step(int64_t n, float *p, const float *g, float *m, float *v, double lr, double b1,
                   double b2, double eps_d, int64_t step) {
    const double bc1 = 1.0 - pow(b1, (double)step);
    const double bc2 = 1.0 - pow(b2, (double)step);
    const float step_size = (float)(lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float beta1 = (float)b1, beta2 = (float)b2, eps = (float)eps_d;
    const float omb1 = (float)(1.0 - b1), omb2 = (float)(1.0 - b2);
    for (int64_t i = 0; i < n; i++) {
        /* ATen's CPU kernels fuse the multiply-add of add_(., alpha) / addcmul_ (pinned by
         * experiment against libtorch 2.10, tests/test_train_oracle.py: bit-equal states) */
        m[i] = fmaf(omb1, g[i], m[i] * beta1);          /* mul_(b1).add_(g, 1 - b1)            */
        v[i] = fmaf(omb2 * g[i], g[i], v[i] * beta2);   /* mul_(b2).addcmul_(g, g, 1 - b2)     */
        const float denom = sqrtf(v[i]) / bc2_sqrt + eps;
        p[i] = p[i] + ((-step_size) * m[i]) / denom;    /* addcdiv_(m, denom, -step_size)      */
    }
}

/* optim_scheduler.cpp:4-7 */
float orc_sched_lr(float lr_init, float lr_final, int max_steps, int step) {
    float t = (float)step / (float)max_steps;
    t = t < 1.0f ? t : 1.0f;
    t = t > 0.0f ? t : 0.0f;
    return expf(logf(lr_init) * (1.0f - t) + logf(lr_final) * t);
}
