"""CPU: the OpenCV-free image path of Camera::loadImage.

* baseline JPEG decoder (gs_image.c) — bit-exact against libjpeg's output (the library cv::imread
  uses): stored fixtures (tests/golden/make_golden_jpeg.py) and, where Pillow is importable, a live
  sweep over sizes / qualities / subsamplings;
* INTER_AREA down-scaling, lens undistortion, optimal new camera matrix: written from OpenCV's
  documented behaviour, PARITY UNPINNED (no OpenCV here) — checked against exact arithmetic and
  against an analytically distorted picture instead."""
import io
import os

import numpy as np
import pytest

from opensplat_amd import colmap

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_fixtures.npz")


def test_jpeg_fixtures_decode_to_libjpegs_pixels():
    g = np.load(GOLD)
    names = sorted(k[:-5] for k in g.files if k.endswith("_file") and k != "progressive_file")
    assert len(names) >= 10
    for n in names:
        got = colmap.decode_jpeg(g[n + "_file"].tobytes())
        assert got.shape == g[n + "_rgb"].shape and np.array_equal(got, g[n + "_rgb"]), n


def test_jpeg_unsupported_and_corrupt_files_are_refused():
    g = np.load(GOLD)
    with pytest.raises(ValueError, match="unsupported"):
        colmap.decode_jpeg(g["progressive_file"].tobytes())
    blob = g["q75_420_file"].tobytes()
    with pytest.raises(ValueError):
        colmap.decode_jpeg(blob[:40])
    with pytest.raises(ValueError):
        colmap.decode_jpeg(b"not a jpeg at all")
    # a file cut in the middle of the scan still decodes (the missing part is grey), like libjpeg
    half = colmap.decode_jpeg(blob[: len(blob) * 2 // 3])
    assert half.shape == g["q75_420_rgb"].shape
    assert np.array_equal(half[:8], g["q75_420_rgb"][:8])


def test_jpeg_live_sweep_against_pillow():
    PIL = pytest.importorskip("PIL")
    from PIL import Image, ImageFile

    ImageFile.MAXBLOCK = 1 << 24
    rs = np.random.RandomState(1)
    n = 0
    for (W, H) in [(64, 48), (203, 117), (8, 8), (33, 65), (2, 5)]:
        img = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
        img[H // 2:] = (np.linspace(0, 255, W)[None, :, None] * np.ones((H - H // 2, 1, 3))).astype(np.uint8)
        for q in (20, 75, 98):
            for sub in (0, 1, 2):
                b = io.BytesIO()
                Image.fromarray(img).save(b, "JPEG", quality=q, subsampling=sub, optimize=bool(q & 1))
                ref = np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGB"))
                assert np.array_equal(colmap.decode_jpeg(b.getvalue()), ref), (W, H, q, sub)
                n += 1
    assert n == 45


def test_resize_area_integer_and_fractional_scales():
    rs = np.random.RandomState(2)
    img = rs.randint(0, 256, (48, 64, 3)).astype(np.uint8)
    half = colmap.downscale_area(img, 2)
    box = img.astype(np.uint32).reshape(24, 2, 32, 2, 3).sum((1, 3))
    assert np.array_equal(half, ((box + 2) >> 2).astype(np.uint8))
    quarter = colmap.downscale_area(img, 4)
    exact = img.astype(np.float64).reshape(12, 4, 16, 4, 3).mean((1, 3))
    assert np.abs(quarter.astype(np.float64) - exact).max() <= 0.5
    # size not a multiple of the factor: 63 x 47 -> 32 x 24 (cv::resize rounds the size), the last
    # row / column average what is left
    odd = colmap.downscale_area(img[:47, :63], 2)
    assert odd.shape == (24, 32, 3)
    assert np.array_equal(odd[:23, :31], half[:23, :31])
    assert np.abs(odd[23, 5].astype(int) - img[46:47, 10:12].reshape(-1, 3).mean(0)).max() <= 0.5
    # fractional scale (Camera::getImage with a size that does not divide): area-weighted mean;
    # constant images stay constant, the mean is preserved, and a brute-force integration agrees
    flat = np.full((30, 45, 3), 77, np.uint8)
    assert (colmap.resize_area(flat, 22, 15) == 77).all()
    small = colmap.resize_area(img[:31, :47], 23, 15)
    assert small.shape == (15, 23, 3)
    sx, sy = 47 / 23.0, 31 / 15.0
    fine = np.kron(img[:31, :47].astype(np.float64), np.ones((15, 23, 1)))     # 15x / 23x super-sampling
    ref = fine.reshape(15, 31, 23, 47, 3).mean((1, 3))
    assert np.abs(small.astype(np.float64) - ref).max() <= 1.0 and abs(sx - sy) < 0.1


def _analytic(u, v):
    return 0.5 + 0.25 * np.sin(u / 11.0) * np.cos(v / 17.0) + 0.2 * np.sin((u + 2.0 * v) / 29.0)


def test_undistortion_recovers_the_ideal_pinhole_view():
    """A smooth picture is rendered through a Brown-Conrady lens (k1, k2, p1, p2, k3); undistorting it
    must give the ideal camera's view of the same picture inside the valid ROI."""
    W, H = 240, 180
    K = np.array([[200.0, 0, 118.0], [0, 198.0, 91.0], [0, 0, 1]], np.float32)
    dist = (-0.18, 0.05, 0.002, -0.001, 0.01)
    newK, roi = colmap.optimal_new_camera_matrix(K, dist, W, H)
    x, y, w, h = roi
    assert 0 <= x < W // 4 and 0 <= y < H // 4 and w > W // 2 and h > H // 2 and x + w <= W and y + h <= H
    # the distorted camera image: pixel (u, v) sees the scene point whose ideal normalised coordinates
    # undistort_points gives; the "scene" is _analytic over ideal pixel coordinates of camera K
    uu, vv = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    xn, yn = colmap.undistort_points(uu, vv, K.astype(np.float64), dist, None, iters=30)
    scene = lambda xn_, yn_: _analytic(xn_ * 200.0 + 118.0, yn_ * 198.0 + 91.0)
    distorted = np.clip(np.rint(scene(xn, yn) * 255.0), 0, 255).astype(np.uint8)
    img = np.repeat(distorted[..., None], 3, 2)
    und = colmap.undistort_image(img, K, dist, newK)
    xi = (uu - newK[0, 2]) / newK[0, 0]
    yi = (vv - newK[1, 2]) / newK[1, 1]
    ideal = scene(xi, yi) * 255.0
    err = np.abs(und[..., 0].astype(np.float64) - ideal)[y:y + h, x:x + w]
    # (the ROI comes from a 9 x 9 sampling of the border, as in OpenCV: between the samples the border
    # pixels of the ROI may blend in a little of the zero border — judged on the interior)
    assert err[2:-2, 2:-2].max() < 2.5 and err.mean() < 0.6, (err[2:-2, 2:-2].max(), err.mean())
    # every ROI pixel draws from inside the source image (no black border inside the ROI)
    xd, yd = colmap._distort(xi, yi, dist)
    su, sv = K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]
    inside = (su >= 0) & (su <= W - 1) & (sv >= 0) & (sv <= H - 1)
    assert inside[y + 2:y + h - 2, x + 2:x + w - 2].all()
    # no distortion: identity matrix up to the (W - 1) / W convention, full-frame ROI, image unchanged
    K0, roi0 = colmap.optimal_new_camera_matrix(K, (0, 0, 0, 0, 0), W, H)
    assert roi0[2] >= W - 1 and roi0[3] >= H - 1
    assert abs(K0[0, 0] / K[0, 0] - (W - 1) / W) < 1e-3 and abs(K0[0, 2] - K[0, 2] * (W - 1) / W) < 0.51
    same = colmap.undistort_image(img, K, (0, 0, 0, 0, 0), K)
    assert np.array_equal(same, img)


def test_load_image_runs_the_reference_sequence_on_a_distorted_jpeg(tmp_path):
    """Camera::loadImage end to end: JPEG file, intrinsics given for a larger image (rescale),
    down-scale by 2, undistort, crop to the ROI, intrinsics of the new pinhole camera."""
    g = np.load(GOLD)
    path = str(tmp_path / "frame.jpg")
    open(path, "wb").write(g["q90_422_file"].tobytes())          # 83 x 61
    cam = colmap.Camera(width=166, height=122, fx=150.0, fy=150.0, cx=83.0, cy=61.0, k1=-0.1, k2=0.02,
                        file_path=path)
    colmap.load_image(cam, downscale=2.0)
    assert not cam.has_distortion()
    assert cam.image.dtype == np.float32 and cam.image.shape == (cam.height, cam.width, 3)
    assert 30 <= cam.width <= 42 and 22 <= cam.height <= 31          # 42 x 31 minus the invalid border
    assert 0.0 <= cam.image.min() and cam.image.max() <= 1.0
    assert 30.0 < cam.fx < 45.0 and abs(cam.cx - cam.width / 2) < 3.0
    plain = colmap.Camera(width=83, height=61, fx=75.0, fy=75.0, cx=41.5, cy=30.5, file_path=path)
    colmap.load_image(plain)
    assert np.array_equal(np.rint(plain.image * 255).astype(np.uint8), g["q90_422_rgb"])
    assert (plain.fx, plain.cx, plain.width, plain.height) == (75.0, 41.5, 83, 61)
