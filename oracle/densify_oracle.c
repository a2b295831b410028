/* TEST INFRASTRUCTURE ONLY — never linked into, imported by, or called from the product path.
 *
 * Plain-C restatement of Model::afterTrain's tensor work:
 *   per-iteration statistics          model.cpp:317-337
 *   split / duplicate / cull          model.cpp:345-458
 *   optimiser-state surgery           model.cpp:253-309 (addToOptimizer / removeFromOptimizer)
 *   quatToRotMat                      tensor_math.cpp:5-28
 *   alpha reset                       model.cpp:464-466
 * as sequential loops that build the new Gaussian set in the reference's order:
 *   cat({originals, split samples [sample-major], duplicates}).index({~culls}).
 * Pinned against the same statements run under libtorch with the reference's own quatToRotMat
 * (oracle/ref_train_shim.cpp: ref_densify_stats / ref_densify_refine) by tests/test_densify_oracle.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void orc_densify_stats(int N, const float *xys_grad, const int32_t *radii, int last_height,
                       int last_width, int first, float *gnorm, float *vis, float *m2d) {
    const float max_side = (float)(last_height > last_width ? last_height : last_width);
    for (int n = 0; n < N; n++) {
        const float gx = xys_grad[2 * n], gy = xys_grad[2 * n + 1];
        const float norm = sqrtf(gx * gx + gy * gy);
        const int visible = radii[n] > 0;
        if (first) { /* :321-323, :329-331 */
            gnorm[n] = norm;
            vis[n] = 1.0f;
            m2d[n] = 0.0f;
        } else if (visible) { /* :325-326 */
            vis[n] = vis[n] + 1.0f;
            gnorm[n] = norm + gnorm[n];
        }
        if (visible) { /* :333-336 */
            const float v = (float)radii[n] / max_side;
            m2d[n] = m2d[n] > v ? m2d[n] : v;
        }
    }
}

static float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
static float max3(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)); }

/* row lengths of means, scales, quats, opacities, featuresDc, featuresRest */
static void row_lens(int K, int *len) {
    len[0] = 3; len[1] = 3; len[2] = 4; len[3] = 1; len[4] = 3; len[5] = (K - 1) * 3;
}

/* counts: [0] nSplits, [1] new N, [2] nDups, [3] cull count.  samples [2*nSplits, 3] (may be NULL
 * when only counts[0] is wanted).  Outputs have capacity 4N rows. */
int orc_densify_refine(int N, int K, const float *const *params, const float *const *exp_avg,
                       const float *const *exp_avg_sq, const float *gnorm, const float *vis,
                       const float *m2d, int last_width, int last_height, float grad_thresh,
                       float size_thresh, int check_screen, float split_screen, int cull_huge,
                       const float *samples, float *const *out_params, float *const *out_exp_avg,
                       float *const *out_exp_avg_sq, int32_t *counts) {
    const float *means = params[0], *scales = params[1], *quats = params[2], *opac = params[3];
    const float max_side = (float)(last_width > last_height ? last_width : last_height);
    const float cull_alpha = 0.1f, cull_scale = 0.5f, cull_screen = 0.15f, size_fac = 1.6f;
    unsigned char *split = (unsigned char *)calloc((size_t)N, 1), *dup = (unsigned char *)calloc((size_t)N, 1);
    int *split_rank = (int *)malloc(sizeof(int) * (size_t)N);
    if (!split || !dup || !split_rank) return -1;
    int n_splits = 0, n_dups = 0;
    for (int n = 0; n < N; n++) {
        const float avg = (gnorm[n] / vis[n]) * 0.5f * max_side; /* :346 */
        const int high = avg > grad_thresh;
        const float size = max3(expf(scales[3 * n]), expf(scales[3 * n + 1]), expf(scales[3 * n + 2]));
        int s = size > size_thresh;                       /* :350 */
        if (check_screen) s = s || (m2d[n] > split_screen); /* :352-353 */
        s = s && high;                                    /* :355 */
        const int d = (size <= size_thresh) && high;      /* :378-379 */
        split[n] = (unsigned char)s;
        dup[n] = (unsigned char)d;
        split_rank[n] = n_splits;
        n_splits += s;
        n_dups += d;
    }
    counts[0] = n_splits;
    counts[2] = n_dups;
    if (!samples && n_splits > 0) { free(split); free(dup); free(split_rank); return 0; }

    int len[6];
    row_lens(K, len);
    int out = 0, culled = 0;
    /* virtual list: originals, samples j = 0, samples j = 1, dups */
    for (int part = 0; part < 4; part++) {
        for (int n = 0; n < N; n++) {
            const int is_orig = part == 0, is_sample = (part == 1 || part == 2) && split[n],
                      is_dup = part == 3 && dup[n];
            if (!(is_orig || is_sample || is_dup)) continue;
            float sc[3];
            for (int k = 0; k < 3; k++)
                sc[k] = is_sample ? logf(expf(scales[3 * n + k]) / size_fac) : scales[3 * n + k]; /* :373 */
            /* cull, :423-441 */
            int cull = sigmoidf_(opac[n]) < cull_alpha;
            if (is_orig && split[n]) cull = 1; /* splitsMask */
            if (cull_huge) {
                int huge = max3(expf(sc[0]), expf(sc[1]), expf(sc[2])) > cull_scale;
                if (check_screen && is_orig) huge = huge || (m2d[n] > cull_screen); /* new: max2DSize = 0 */
                cull = cull || huge;
            }
            if (cull) { culled++; continue; }
            for (int t = 0; t < 6; t++) {
                for (int c = 0; c < len[t]; c++) {
                    out_params[t][(size_t)out * len[t] + c] = params[t][(size_t)n * len[t] + c];
                    /* survivors keep their moments, new Gaussians start from zero */
                    out_exp_avg[t][(size_t)out * len[t] + c] = is_orig ? exp_avg[t][(size_t)n * len[t] + c] : 0.0f;
                    out_exp_avg_sq[t][(size_t)out * len[t] + c] = is_orig ? exp_avg_sq[t][(size_t)n * len[t] + c] : 0.0f;
                }
            }
            if (is_sample) {
                const int j = part - 1;
                const float *smp = samples + 3 * ((size_t)j * n_splits + split_rank[n]);
                /* scaledSamples, :361 */
                const float v0 = expf(scales[3 * n]) * smp[0], v1 = expf(scales[3 * n + 1]) * smp[1],
                            v2 = expf(scales[3 * n + 2]) * smp[2];
                /* qs = quats / |quats| (:362); quatToRotMat normalises again (tensor_math.cpp:6) */
                float q[4];
                float nrm = 0.0f;
                for (int k = 0; k < 4; k++) nrm += quats[4 * n + k] * quats[4 * n + k];
                nrm = sqrtf(nrm);
                for (int k = 0; k < 4; k++) q[k] = quats[4 * n + k] / nrm;
                float nrm2 = 0.0f;
                for (int k = 0; k < 4; k++) nrm2 += q[k] * q[k];
                nrm2 = fmaxf(sqrtf(nrm2), 1e-12f);
                const float w = q[0] / nrm2, x = q[1] / nrm2, y = q[2] / nrm2, z = q[3] / nrm2;
                const float R[9] = {1.0f - 2.0f * (y * y + z * z), 2.0f * (x * y - w * z), 2.0f * (x * z + w * y),
                                    2.0f * (x * y + w * z), 1.0f - 2.0f * (x * x + z * z), 2.0f * (y * z - w * x),
                                    2.0f * (x * z - w * y), 2.0f * (y * z + w * x), 1.0f - 2.0f * (x * x + y * y)};
                for (int k = 0; k < 3; k++) /* bmm + means, :364-365 */
                    out_params[0][(size_t)out * 3 + k] =
                        (R[3 * k] * v0 + R[3 * k + 1] * v1 + R[3 * k + 2] * v2) + means[3 * n + k];
                for (int k = 0; k < 3; k++) out_params[1][(size_t)out * 3 + k] = sc[k];
            }
            out++;
        }
    }
    counts[1] = out;
    counts[3] = culled;
    free(split); free(dup); free(split_rank);
    return 0;
}

/* model.cpp:464-466: opacities = clamp_max(opacities, logit(resetValue)) */
void orc_reset_opacity(int N, float reset_value, float *logits) {
    const float mx = logf(reset_value / (1.0f - reset_value));
    for (int n = 0; n < N; n++) logits[n] = logits[n] < mx ? logits[n] : mx;
}
