/* image_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the three OpenCV operations Camera::loadImage / Camera::getImage apply to a
 * capture (input_data.cpp:54-61, 66-84, 112):
 *     cv::resize(..., INTER_AREA)          8-bit, 3 channels, down-scaling
 *     cv::getOptimalNewCameraMatrix(K, dist, size, alpha = 0, Size(), &roi)
 *     cv::undistort(src, dst, K, dist, newK)
 * OpenCV is a third-party dependency that is absent from /root/reference and from this image (no
 * network).  PINNED TO THE PUBLISHED ALGORITHM, NOT TO AN OPENCV BUILD: what follows restates OpenCV
 * 4.5.4 (Ubuntu 22.04's libopencv-dev, what the reference's Dockerfile and README install) as its
 * sources read —
 *     modules/imgproc/src/resize.cpp        resize(), ResizeAreaFast_Invoker, ResizeAreaFastVec_SIMD_8u,
 *                                           computeResizeAreaTab, ResizeArea_Invoker
 *     modules/calib3d/src/calibration.cpp   cvGetOptimalNewCameraMatrix, icvGetRectangles
 *     modules/calib3d/src/undistort.dispatch.cpp  cvUndistortPointsInternal (5 iterations),
 *                                           undistort() (row stripes), initUndistortRectifyMap
 *                                           (CV_16SC2 + CV_16UC1 maps, INTER_BITS = 5)
 *     modules/imgproc/src/imgwarp.cpp       remapBilinear (fixed-point weights scaled by 2^15),
 *     modules/core/src/lapack.cpp           invert(): closed-form 3 x 3 inverse
 *     saturate_cast<int / uchar>(float / double) = cvRound = round to nearest, ties to even
 * with the same operation order and the same precisions (float where OpenCV uses float, double where
 * it uses double), so that an implementation that follows the same sources agrees with it BIT FOR BIT.
 * tests/test_image.py holds opensplat_amd/colmap.py to that.  The 2 x 2 case of the area resize takes
 * OpenCV's SIMD path ((a + b + c + d + 2) >> 2), which every x86-64 / aarch64 build has.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int cv_round_d(double v) { return (int)nearbyint(v); }   /* default rounding mode: ties to even */
static int cv_round_f(float v) { return (int)nearbyintf(v); }
static int cv_ceil(double v) { return (int)ceil(v); }
static int cv_floor(double v) { return (int)floor(v); }
static uint8_t sat_u8_i(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
static uint8_t sat_u8_f(float v) { return sat_u8_i(cv_round_f(v)); }

/* ------------------------------------------------------------------------------------------------
 * cv::resize(src, dst, dsize, inv_scale_x, inv_scale_y, INTER_AREA), CV_8UC3, down-scaling.
 * dw = dh = 0: sizes from the scale factors, dsize = cvRound(ssize * inv_scale) (resize.cpp, resize()).
 * Otherwise the scale factors follow from the sizes (inv_scale = dsize / ssize).  *out_w / *out_h
 * receive the destination size; dst may be NULL to query it.  Returns 0, or -1 when the request is
 * not a down-scale in both directions (INTER_AREA then means something else in OpenCV). */
typedef struct { int si, di; float alpha; } AreaTab;

static int area_tab(int ssize, int dsize, int cn, double scale, AreaTab *tab) {
    /* computeResizeAreaTab */
    int k = 0;
    for (int dx = 0; dx < dsize; dx++) {
        double fsx1 = dx * scale;
        double fsx2 = fsx1 + scale;
        double cellWidth = fmin(scale, ssize - fsx1);
        int sx1 = cv_ceil(fsx1), sx2 = cv_floor(fsx2);
        sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
        sx1 = sx1 < sx2 ? sx1 : sx2;
        if (sx1 - fsx1 > 1e-3) {
            tab[k].di = dx * cn;
            tab[k].si = (sx1 - 1) * cn;
            tab[k++].alpha = (float)((sx1 - fsx1) / cellWidth);
        }
        for (int sx = sx1; sx < sx2; sx++) {
            tab[k].di = dx * cn;
            tab[k].si = sx * cn;
            tab[k++].alpha = (float)(1.0 / cellWidth);
        }
        if (fsx2 - sx2 > 1e-3) {
            tab[k].di = dx * cn;
            tab[k].si = sx2 * cn;
            tab[k++].alpha = (float)(fmin(fmin(fsx2 - sx2, 1.), cellWidth) / cellWidth);
        }
    }
    return k;
}

int orc_resize_area_u8c3(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh,
                         double inv_scale_x, double inv_scale_y, int *out_w, int *out_h) {
    const int cn = 3;
    if (dw <= 0 || dh <= 0) {
        dw = cv_round_d(sw * inv_scale_x);
        dh = cv_round_d(sh * inv_scale_y);
    } else {
        inv_scale_x = (double)dw / sw;
        inv_scale_y = (double)dh / sh;
    }
    if (out_w) *out_w = dw;
    if (out_h) *out_h = dh;
    if (dw <= 0 || dh <= 0) return -1;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    if (!(scale_x >= 1 && scale_y >= 1)) return -1;
    if (!dst) return 0;
    const int iscale_x = cv_round_d(scale_x), iscale_y = cv_round_d(scale_y);
    const int is_area_fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
    if (is_area_fast) {
        /* ResizeAreaFast_Invoker<uchar, int, ResizeAreaFastVec<uchar, ResizeAreaFastVec_SIMD_8u>> */
        const int area = iscale_x * iscale_y;
        const float scale = 1.f / area;
        const int dwidth1 = (sw / iscale_x) * cn;
        const int dsw = dw * cn, ssw = sw * cn;
        const int fast2 = iscale_x == 2 && iscale_y == 2;
        for (int dy = 0; dy < dh; dy++) {
            uint8_t *D = dst + (size_t)dy * dsw;
            const int sy0 = dy * iscale_y;
            const int w = sy0 + iscale_y <= sh ? dwidth1 : 0;
            if (sy0 >= sh) {
                memset(D, 0, (size_t)dsw);
                continue;
            }
            int dx = 0;
            for (; dx < w; dx++) {
                const uint8_t *S = src + (size_t)sy0 * ssw + (dx / cn) * iscale_x * cn + dx % cn;
                int sum = 0;
                for (int sy = 0; sy < iscale_y; sy++)
                    for (int sx = 0; sx < iscale_x; sx++) sum += S[(size_t)sy * ssw + sx * cn];
                /* the SIMD path of the 2 x 2 case rounds half up in integers; the generic path
                 * multiplies by 1 / area in float and rounds to nearest even */
                D[dx] = fast2 ? (uint8_t)((sum + 2) >> 2) : sat_u8_f((float)sum * scale);
            }
            for (; dx < dsw; dx++) {
                int sum = 0, count = 0;
                const int sx0 = (dx / cn) * iscale_x * cn + dx % cn;
                if (sx0 >= ssw) D[dx] = 0;
                for (int sy = 0; sy < iscale_y; sy++) {
                    if (sy0 + sy >= sh) break;
                    const uint8_t *S = src + (size_t)(sy0 + sy) * ssw + sx0;
                    for (int sx = 0; sx < iscale_x * cn; sx += cn) {
                        if (sx0 + sx >= ssw) break;
                        sum += S[sx];
                        count++;
                    }
                }
                D[dx] = count ? sat_u8_f((float)sum / count) : 0;
            }
        }
        return 0;
    }
    /* ResizeArea_Invoker<uchar, float> */
    AreaTab *xtab = (AreaTab *)malloc(sizeof(AreaTab) * (size_t)(sw * 2 + 2));
    AreaTab *ytab = (AreaTab *)malloc(sizeof(AreaTab) * (size_t)(sh * 2 + 2));
    const int xn = area_tab(sw, dw, cn, scale_x, xtab);
    const int yn = area_tab(sh, dh, 1, scale_y, ytab);
    const int dsw = dw * cn;
    float *buf = (float *)malloc(sizeof(float) * (size_t)dsw);
    float *sum = (float *)malloc(sizeof(float) * (size_t)dsw);
    int prev_dy = ytab[0].di;
    for (int dx = 0; dx < dsw; dx++) sum[dx] = 0.f;
    for (int j = 0; j < yn; j++) {
        const int dy = ytab[j].di, sy = ytab[j].si;
        const float beta = ytab[j].alpha;
        const uint8_t *S = src + (size_t)sy * sw * cn;
        for (int dx = 0; dx < dsw; dx++) buf[dx] = 0.f;
        for (int k = 0; k < xn; k++) {
            const int dxn = xtab[k].di;
            const float alpha = xtab[k].alpha;
            for (int c = 0; c < cn; c++) buf[dxn + c] += S[xtab[k].si + c] * alpha;
        }
        if (dy != prev_dy) {
            uint8_t *D = dst + (size_t)prev_dy * dsw;
            for (int dx = 0; dx < dsw; dx++) {
                D[dx] = sat_u8_f(sum[dx]);
                sum[dx] = beta * buf[dx];
            }
            prev_dy = dy;
        } else {
            for (int dx = 0; dx < dsw; dx++) sum[dx] += beta * buf[dx];
        }
    }
    {
        uint8_t *D = dst + (size_t)prev_dy * dsw;
        for (int dx = 0; dx < dsw; dx++) D[dx] = sat_u8_f(sum[dx]);
    }
    free(xtab); free(ytab); free(buf); free(sum);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * cvUndistortPointsInternal with TermCriteria(COUNT, 5, 0.01), R = identity; P = newK (row-major
 * 3 x 3 doubles) or NULL (normalised coordinates).  dist = (k1, k2, p1, p2, k3, k4, k5, k6). */
static void undistort_point(double u, double v, const double *A, const double *k, const double *P,
                            double *ox, double *oy) {
    const double fx = A[0], fy = A[4], ifx = 1. / fx, ify = 1. / fy, cx = A[2], cy = A[5];
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) /
                              (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) {
            x = (u - cx) * ifx;
            y = (v - cy) * ify;
            break;
        }
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    if (P) {
        const double xx = P[0] * x + P[1] * y + P[2], yy = P[3] * x + P[4] * y + P[5];
        const double ww = 1. / (P[6] * x + P[7] * y + P[8]);
        x = xx * ww;
        y = yy * ww;
    }
    *ox = x;
    *oy = y;
}

/* icvGetRectangles: 9 x 9 grid over [0, W-1] x [0, H-1]; rect = (x, y, w, h) */
static void get_rectangles(const double *A, const double *k, const double *P, int W, int H,
                           double inner[4], double outer[4]) {
    const int N = 9;
    double iX0 = -FLT_MAX, iX1 = FLT_MAX, iY0 = -FLT_MAX, iY1 = FLT_MAX;
    double oX0 = FLT_MAX, oX1 = -FLT_MAX, oY0 = FLT_MAX, oY1 = -FLT_MAX;
    for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++) {
            double px, py;
            undistort_point((double)x * (W - 1) / (N - 1), (double)y * (H - 1) / (N - 1), A, k, P, &px, &py);
            oX0 = fmin(oX0, px); oX1 = fmax(oX1, px);
            oY0 = fmin(oY0, py); oY1 = fmax(oY1, py);
            if (x == 0) iX0 = fmax(iX0, px);
            if (x == N - 1) iX1 = fmin(iX1, px);
            if (y == 0) iY0 = fmax(iY0, py);
            if (y == N - 1) iY1 = fmin(iY1, py);
        }
    inner[0] = iX0; inner[1] = iY0; inner[2] = iX1 - iX0; inner[3] = iY1 - iY0;
    outer[0] = oX0; outer[1] = oY0; outer[2] = oX1 - oX0; outer[3] = oY1 - oY0;
}

/* cv::getOptimalNewCameraMatrix(K (CV_32F), dist (8 floats), Size(W, H), alpha, Size(), &roi):
 * newK comes back in K's type (float); the ROI is computed from the double-precision matrix. */
int orc_optimal_new_camera_matrix(const float *K9, const float *dist8, int W, int H, double alpha,
                                  float *newK9, int roi[4]) {
    double A[9], k[8], M[9];
    for (int i = 0; i < 9; i++) A[i] = K9[i];
    for (int i = 0; i < 8; i++) k[i] = dist8[i];
    memcpy(M, A, sizeof(M));
    double inner[4], outer[4];
    get_rectangles(A, k, NULL, W, H, inner, outer);
    const double fx0 = (W - 1) / inner[2], fy0 = (H - 1) / inner[3];
    const double cx0 = -fx0 * inner[0], cy0 = -fy0 * inner[1];
    const double fx1 = (W - 1) / outer[2], fy1 = (H - 1) / outer[3];
    const double cx1 = -fx1 * outer[0], cy1 = -fy1 * outer[1];
    M[0] = fx0 * (1 - alpha) + fx1 * alpha;
    M[4] = fy0 * (1 - alpha) + fy1 * alpha;
    M[2] = cx0 * (1 - alpha) + cx1 * alpha;
    M[5] = cy0 * (1 - alpha) + cy1 * alpha;
    get_rectangles(A, k, M, W, H, inner, outer);
    /* cv::Rect r = inner (Rect_<double> -> Rect_<int>: saturate_cast = cvRound); r &= image */
    int x = cv_round_d(inner[0]), y = cv_round_d(inner[1]);
    int w = cv_round_d(inner[2]), h = cv_round_d(inner[3]);
    int x0 = x > 0 ? x : 0, y0 = y > 0 ? y : 0;
    int x1 = x + w < W ? x + w : W, y1 = y + h < H ? y + h : H;
    if (x1 <= x0 || y1 <= y0) { x0 = y0 = x1 = y1 = 0; }
    roi[0] = x0; roi[1] = y0; roi[2] = x1 - x0; roi[3] = y1 - y0;
    for (int i = 0; i < 9; i++) newK9[i] = (float)M[i];
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * cv::undistort(src, dst, K, dist, newK), CV_8UC3: row stripes of initUndistortRectifyMap (fixed-point
 * maps, 5 fractional bits) + remap(INTER_LINEAR, BORDER_CONSTANT 0). */
static void invert3(const double *S, double *t) {   /* cv::invert, n == 3 */
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) +
               S[2] * (S[3] * S[7] - S[4] * S[6]);
    d = 1. / d;
    t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
    t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
    t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
    t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
    t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
    t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
    t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
    t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
    t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
}

int orc_undistort_u8c3(const uint8_t *src, int W, int H, const float *K9, const float *dist8,
                       const float *newK9, uint8_t *dst) {
    const int cn = 3;
    double A[9], Ar[9], k[8];
    for (int i = 0; i < 9; i++) { A[i] = K9[i]; Ar[i] = newK9[i]; }
    for (int i = 0; i < 8; i++) k[i] = dist8[i];
    const double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3], k3 = k[4], k4 = k[5], k5 = k[6], k6 = k[7];
    const double fx = A[0], fy = A[4], u0 = A[2], v0a = A[5];
    int stripe0 = (1 << 12) / (W > 1 ? W : 1);
    stripe0 = stripe0 > 1 ? stripe0 : 1;
    stripe0 = stripe0 < H ? stripe0 : H;
    const double v0 = Ar[5];
    for (int y = 0; y < H; y += stripe0) {
        const int stripe = stripe0 < H - y ? stripe0 : H - y;
        Ar[5] = v0 - y;
        double ir[9];
        invert3(Ar, ir);   /* (Ar * I).inv(DECOMP_LU) */
        for (int i = 0; i < stripe; i++) {
            uint8_t *D = dst + (size_t)(y + i) * W * cn;
            double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
            for (int j = 0; j < W; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
                const double w = 1. / _w, x = _x * w, yy = _y * w;
                const double x2 = x * x, y2 = yy * yy;
                const double r2 = x2 + y2, _2xy = 2 * x * yy;
                const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
                const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2));
                const double yd = (yy * kr + p1 * (r2 + 2 * y2) + p2 * _2xy);
                const double u = fx * xd + u0, v = fy * yd + v0a;
                const int iu = cv_round_d(u * 32), iv = cv_round_d(v * 32);
                const int sx = (short)(iu >> 5), sy = (short)(iv >> 5);
                const int fxq = iu & 31, fyq = iv & 31;
                /* remapBilinear: weights (1 - a)(1 - b) .. a b scaled by 2^15 (exact multiples of 32),
                 * result = (sum + 2^14) >> 15; taps outside the image read the border value 0 */
                const int w00 = (32 - fxq) * (32 - fyq) * 32, w01 = fxq * (32 - fyq) * 32;
                const int w10 = (32 - fxq) * fyq * 32, w11 = fxq * fyq * 32;
                for (int c = 0; c < cn; c++) {
                    int v00 = 0, v01 = 0, v10 = 0, v11 = 0;
                    if (sy >= 0 && sy < H) {
                        if (sx >= 0 && sx < W) v00 = src[((size_t)sy * W + sx) * cn + c];
                        if (sx + 1 >= 0 && sx + 1 < W) v01 = src[((size_t)sy * W + sx + 1) * cn + c];
                    }
                    if (sy + 1 >= 0 && sy + 1 < H) {
                        if (sx >= 0 && sx < W) v10 = src[((size_t)(sy + 1) * W + sx) * cn + c];
                        if (sx + 1 >= 0 && sx + 1 < W) v11 = src[((size_t)(sy + 1) * W + sx + 1) * cn + c];
                    }
                    D[j * cn + c] = sat_u8_i((v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15);
                }
            }
        }
    }
    return 0;
}
