#!/usr/bin/env python3
"""tests/golden/jpeg_fixtures.npz: small JPEG files and the pixels LIBJPEG decodes them to.

The reference reads its training images with cv::imread (cv_utils.cpp:3-14), i.e. with the IJG
library's default decoding path (islow IDCT, fancy upsampling).  OpenCV is not installed here, but
Pillow wraps the same library with the same defaults, so its output pins opensplat_amd's own decoder
(gs_image.c).  Each fixture stores the file's bytes and libjpeg's RGB output.

Usage: python tests/golden/make_golden_jpeg.py        (needs Pillow; the tests do not)"""
import io
import os

import numpy as np
from PIL import Image, ImageFile

ImageFile.MAXBLOCK = 1 << 24
HERE = os.path.dirname(os.path.abspath(__file__))


def picture(W, H, seed):
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    a = np.stack([128 + 100 * np.sin(xx / 7.0 + yy / 13.0), 128 + 90 * np.cos(xx / 5.0) * np.sin(yy / 9.0),
                  (xx * 3 + yy * 5) % 256], -1).astype(np.float64)
    a += rs.normal(0, 12, a.shape)
    a[H // 3:H // 3 + 3] = 255
    a[:, W // 2:W // 2 + 2] = 0
    return np.clip(a, 0, 255).astype(np.uint8)


CASES = [  # name, W, H, save options
    ("q75_420", 83, 61, dict(quality=75, subsampling=2)),
    ("q90_422", 83, 61, dict(quality=90, subsampling=1)),
    ("q95_444_opt", 83, 61, dict(quality=95, subsampling=0, optimize=True)),
    ("q30_420", 40, 24, dict(quality=30, subsampling=2)),
    ("q100_420_odd", 17, 9, dict(quality=100, subsampling=2)),
    ("q85_420_rst", 83, 61, dict(quality=85, subsampling=2, restart_marker_blocks=2)),
    ("q85_444_rst", 50, 33, dict(quality=85, subsampling=0, restart_marker_rows=1)),
    ("tiny_1x1", 1, 1, dict(quality=90, subsampling=2)),
    ("tiny_3x2", 3, 2, dict(quality=90, subsampling=2)),
    ("grey", 45, 37, dict(quality=80)),
]


PROGRESSIVE = [  # SOF2: DC first / refinement, AC bands first / refinement (libjpeg's default script)
    ("prog_q85_420", 83, 61, dict(quality=85, subsampling=2)),
    ("prog_q92_422_opt", 83, 61, dict(quality=92, subsampling=1, optimize=True)),
    ("prog_q97_444", 50, 33, dict(quality=97, subsampling=0)),
    ("prog_q25_420", 64, 40, dict(quality=25, subsampling=2)),
    ("prog_q85_420_rst", 83, 61, dict(quality=85, subsampling=2, restart_marker_blocks=3)),
    ("prog_tiny_3x2", 3, 2, dict(quality=90, subsampling=2)),
    ("prog_q80_grey", 45, 37, dict(quality=80)),
]


def main():
    out = {}
    for i, (name, W, H, opts) in enumerate(CASES):
        img = picture(W, H, i)
        if name == "grey":
            img = img[..., 0]
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", **opts)
        blob = b.getvalue()
        out[name + "_file"] = np.frombuffer(blob, np.uint8)
        out[name + "_rgb"] = np.asarray(Image.open(io.BytesIO(blob)).convert("RGB"))
    for i, (name, W, H, opts) in enumerate(PROGRESSIVE):
        img = picture(W, H, 50 + i)
        if name.endswith("grey"):
            img = img[..., 0]
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", progressive=True, **opts)
        blob = b.getvalue()
        assert b"\xff\xc2" in blob
        out[name + "_file"] = np.frombuffer(blob, np.uint8)
        out[name + "_rgb"] = np.asarray(Image.open(io.BytesIO(blob)).convert("RGB"))
    np.savez_compressed(os.path.join(HERE, "jpeg_fixtures.npz"), **out)
    print("wrote %d fixtures" % (len(CASES) + len(PROGRESSIVE)))


if __name__ == "__main__":
    main()
