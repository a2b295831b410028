/* gsplat_image.h — host-side image ingest of the training path, OpenCV-free.
 *
 * Replaces, for a COLMAP capture, what Camera::loadImage (input_data.cpp:40-96) gets from OpenCV:
 *   imreadRGB (cv_utils.cpp:3-14: cv::imread + BGR->RGB)      -> gs_jpeg_info / gs_jpeg_decode_rgb
 * for 8-bit Huffman-coded JPEG files — baseline / extended sequential (one scan or one per component) and
 * progressive (SOF2: spectral selection + successive approximation) — what cameras, phones and COLMAP
 * pipelines write.
 * The decoder follows the algorithms of the IJG library that cv::imread delegates to (libjpeg /
 * libjpeg-turbo defaults: "islow" integer IDCT jidctint.c, "fancy" triangle chroma upsampling
 * jdsample.c, fixed-point YCbCr->RGB jdcolor.c), so that the pixels are the ones OpenSplat trains on:
 * bit-exact against libjpeg's own output on every fixture (tests/test_image.py, pinned through
 * Pillow, which wraps the same library).
 * Not supported (GS_ERR_UNSUPPORTED): arithmetic-coded / lossless / 12-bit / CMYK files.
 * EXIF orientation is NOT applied (cv::imread applies it; COLMAP's own reader does not).
 *
 * Plain host C (libgsplat_image.so, gcc): no HIP, no torch.  Status codes as in gsplat_hip.h.
 */
#ifndef GSPLAT_IMAGE_H
#define GSPLAT_IMAGE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_IMG_OK 0
#define GS_IMG_ERR_INVALID_ARGUMENT (-1)
#define GS_IMG_ERR_UNSUPPORTED (-2)
#define GS_IMG_ERR_CORRUPT (-6)

/* Parses the headers: image size and number of components (1 = greyscale, 3 = YCbCr / RGB). */
int gs_jpeg_info(const uint8_t *data, size_t size, int *width, int *height, int *components);

/* Decodes into interleaved 8-bit RGB, row-major [height, width, 3] (greyscale is replicated to the
 * three channels, as cv::imread's default flag does). */
int gs_jpeg_decode_rgb(const uint8_t *data, size_t size, uint8_t *out_rgb, size_t out_bytes);

const char *gs_image_strerror(int status);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_IMAGE_H */
