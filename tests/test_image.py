"""CPU: the OpenCV-free image path of Camera::loadImage.

* baseline JPEG decoder (gs_image.c) — bit-exact against libjpeg's output (the library cv::imread
  uses): stored fixtures (tests/golden/make_golden_jpeg.py) and, where Pillow is importable, a live
  sweep over sizes / qualities / subsamplings;
* INTER_AREA down-scaling, lens undistortion, optimal new camera matrix: PINNED TO THE PUBLISHED
  ALGORITHM — oracle/image_oracle.c restates OpenCV 4.5.4's sources (resize.cpp, calibration.cpp,
  undistort.dispatch.cpp, imgwarp.cpp) in plain C and opensplat_amd/colmap.py must agree with it bit for
  bit on random and on lens-distorted synthetic captures; not pinned to an OpenCV build (none offline).
  Exact arithmetic and an analytically distorted picture check the meaning of the results;
* EXIF orientation (cv::imread applies it) and the hand-over of what the decoder refuses (CMYK, ...) to Pillow."""
import io
import os

import numpy as np
import pytest

from opensplat_amd import colmap

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_fixtures.npz")


def test_jpeg_fixtures_decode_to_libjpegs_pixels():
    g = np.load(GOLD)
    names = sorted(k[:-5] for k in g.files if k.endswith("_file") and not k.startswith("prog"))
    assert len(names) >= 10
    for n in names:
        got = colmap.decode_jpeg(g[n + "_file"].tobytes())
        assert got.shape == g[n + "_rgb"].shape and np.array_equal(got, g[n + "_rgb"]), n


def test_progressive_jpeg_fixtures_decode_to_libjpegs_pixels():
    """SOF2 files (spectral selection + successive approximation, T.81 annex G): 4:2:0 / 4:2:2 / 4:4:4,
    greyscale, optimised tables, restart markers — libjpeg's final pixels, bit for bit."""
    g = np.load(GOLD)
    names = sorted(k[:-5] for k in g.files if k.startswith("prog_") and k.endswith("_file"))
    assert len(names) >= 6
    for n in names:
        got = colmap.decode_jpeg(g[n + "_file"].tobytes())
        assert got.shape == g[n + "_rgb"].shape and np.array_equal(got, g[n + "_rgb"]), n


def _split_scans_444(blob):
    """A baseline 4:4:4 file re-written as a sequential file with ONE SCAN PER COMPONENT (T.81 allows
    it; some encoders write it): with 1x1 sampling every component's blocks come in the same raster order
    in both layouts, so each block's bits are copied as they are."""
    def segs(b):
        i, out = 2, []
        while b[i + 1] != 0xDA:
            n = (b[i + 2] << 8) | b[i + 3]
            out.append(b[i:i + 2 + n]); i += 2 + n
        n = (b[i + 2] << 8) | b[i + 3]
        return out, b[i:i + 2 + n], b[i + 2 + n:]
    hdrs, sos, data = segs(blob)
    huff = {}
    for h in hdrs:
        if h[1] == 0xC4:
            q = 4
            while q < len(h):
                tc_th, bits = h[q], list(h[q + 1:q + 17]); vals = h[q + 17:q + 17 + sum(bits)]
                code, k, table = 0, 0, {}
                for ln in range(1, 17):
                    for _ in range(bits[ln - 1]):
                        table[(ln, code)] = vals[k]; code += 1; k += 1
                    code <<= 1
                huff[tc_th] = table; q += 17 + sum(bits)
        if h[1] == 0xC0:
            H, W, nc = (h[5] << 8) | h[6], (h[7] << 8) | h[8], h[9]
            assert nc == 3 and all(h[11 + 3 * i] == 0x11 for i in range(3))
    comps = [(sos[5 + 2 * i], sos[6 + 2 * i]) for i in range(3)]
    raw = bytearray(); i = 0
    while i < len(data):                          # un-stuff, stop at EOI
        if data[i] == 0xFF:
            if data[i + 1] == 0: raw.append(0xFF); i += 2; continue
            break
        raw.append(data[i]); i += 1
    bits = "".join(format(x, "08b") for x in raw)
    pos, streams = 0, ["", "", ""]

    def sym(tab):
        nonlocal pos
        code = 0
        for ln in range(1, 17):
            code = (code << 1) | int(bits[pos]); pos += 1
            if (ln, code) in tab:
                return tab[(ln, code)]
        raise AssertionError("bad code")
    for _ in range(((W + 7) // 8) * ((H + 7) // 8)):
        for c, (cid, t) in enumerate(comps):
            start = pos
            cat = sym(huff[t >> 4])                      # DC: category, then that many bits
            pos += cat
            k = 1
            while k < 64:
                rs = sym(huff[0x10 | (t & 15)])
                if rs == 0: break
                if rs == 0xF0: k += 16; continue
                k += (rs >> 4) + 1; pos += rs & 15
            streams[c] += bits[start:pos]
    out = bytearray(b"\xff\xd8") + b"".join(hdrs)
    for c, (cid, t) in enumerate(comps):
        out += bytes([0xFF, 0xDA, 0, 8, 1, cid, t, 0, 63, 0])
        st = streams[c] + "1" * (-len(streams[c]) % 8)
        for q in range(0, len(st), 8):
            v = int(st[q:q + 8], 2); out.append(v)
            if v == 0xFF: out.append(0)
    return bytes(out + b"\xff\xd9")


def test_sequential_file_with_one_scan_per_component():
    g = np.load(GOLD)
    blob = _split_scans_444(g["q95_444_opt_file"].tobytes())
    assert blob.count(b"\xff\xda") == 3
    assert np.array_equal(colmap.decode_jpeg(blob), g["q95_444_opt_rgb"])


def test_jpeg_unsupported_and_corrupt_files_are_refused():
    g = np.load(GOLD)
    # arithmetic coding (SOF9) is refused by name; so is 12-bit precision
    blob = bytearray(g["q75_420_file"].tobytes())
    sof = blob.index(b"\xff\xc0")
    ari = bytes(blob[:sof + 1]) + b"\xc9" + bytes(blob[sof + 2:])
    with pytest.raises(ValueError, match="unsupported"):
        colmap.decode_jpeg(ari)
    deep = bytearray(blob); deep[sof + 4] = 12
    with pytest.raises(ValueError, match="unsupported"):
        colmap.decode_jpeg(bytes(deep))
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
                for prog in (False, True):
                    b = io.BytesIO()
                    Image.fromarray(img).save(b, "JPEG", quality=q, subsampling=sub, optimize=bool(q & 1),
                                              progressive=prog)
                    ref = np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGB"))
                    assert np.array_equal(colmap.decode_jpeg(b.getvalue()), ref), (W, H, q, sub, prog)
                    n += 1
    assert n == 90


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
    odd = colmap.resize_area(img[:47, :63], inv_scale=0.5)
    assert odd.shape == (24, 32, 3)
    assert np.array_equal(odd[:23, :31], half[:23, :31])
    assert np.abs(odd[23, 5].astype(int) - img[46:47, 10:12].reshape(-1, 3).mean(0)).max() <= 0.5
    # Camera::getImage's pyramid (input_data.cpp:112) truncates the size instead: 63 x 47 -> 31 x 23
    assert colmap.downscale_area(img[:47, :63], 2).shape == (23, 31, 3)
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
    # no distortion: the same matrix (the 9 x 9 grid spans [0, W-1] x [0, H-1] in OpenCV 4.5), full-frame
    # ROI, image unchanged
    K0, roi0 = colmap.optimal_new_camera_matrix(K, (0, 0, 0, 0, 0), W, H)
    assert roi0[2] >= W - 1 and roi0[3] >= H - 1
    assert np.abs(K0 - K).max() < 1e-3
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


# ---- pinned to the published algorithm: colmap.py against oracle/image_oracle.c ---------------------

@pytest.mark.parametrize("sw,sh,kw", [(64, 48, dict(inv_scale=0.5)), (63, 47, dict(inv_scale=0.5)),
                                      (65, 49, dict(inv_scale=0.25)), (97, 61, dict(inv_scale=float(np.float32(1.0) / np.float32(3.0)))),
                                      (120, 90, dict(inv_scale=float(np.float32(1.0) / np.float32(1.5)))),
                                      (101, 77, dict(dst_w=50, dst_h=38)), (96, 64, dict(dst_w=32, dst_h=16)),
                                      (96, 64, dict(dst_w=24, dst_h=16)), (57, 43, dict(dst_w=19, dst_h=14)),
                                      (640, 480, dict(inv_scale=0.125)), (31, 47, dict(dst_w=23, dst_h=15))])
def test_resize_area_is_bit_exact_with_the_restated_opencv_algorithm(sw, sh, kw, restated):
    rs = np.random.RandomState(sw * 1000 + sh)
    img = rs.randint(0, 256, (sh, sw, 3)).astype(np.uint8)
    img[: sh // 3] = (np.linspace(0, 255, sw)[None, :, None] * np.ones((sh // 3, 1, 3))).astype(np.uint8)
    got = colmap.resize_area(img, **kw)
    want = restated.resize_area(img, kw.get("dst_w", 0), kw.get("dst_h", 0), kw.get("inv_scale", 0.0))
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.array_equal(got, want), int((got != want).sum())


def _lens_capture(W, H, K, dist, seed):
    """A textured scene seen through the lens: what a COLMAP OPENCV-model camera delivers."""
    rs = np.random.RandomState(seed)
    uu, vv = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    xn, yn = colmap.undistort_points(uu, vv, np.asarray(K, np.float64), dist, None, iters=30)
    chans = []
    for c in range(3):
        a, b, ph = rs.uniform(20, 90), rs.uniform(20, 90), rs.uniform(0, 6)
        chans.append(0.5 + 0.3 * np.sin(xn * a + ph) * np.cos(yn * b) + 0.15 * np.sin((xn + 2 * yn) * 37.0))
    img = np.clip(np.rint(np.stack(chans, -1) * 255.0), 0, 255).astype(np.uint8)
    return img ^ (rs.randint(0, 4, img.shape).astype(np.uint8))      # + sensor noise in the low bits


LENSES = [
    (240, 180, [[200.0, 0, 118.0], [0, 198.0, 91.0], [0, 0, 1]], (-0.18, 0.05, 0.002, -0.001, 0.01)),
    (203, 117, [[150.5, 0, 101.25], [0, 149.75, 58.5], [0, 0, 1]], (0.12, -0.03, -0.0015, 0.0025, 0.0)),   # pincushion
    (1300, 40, [[900.0, 0, 650.0], [0, 900.0, 20.0], [0, 0, 1]], (-0.05, 0.0, 0.0, 0.0, 0.0)),             # stripes of 3 rows
    (64, 300, [[80.0, 0, 31.5], [0, 80.0, 150.0], [0, 0, 1]], (-0.25, 0.08, 0.0, 0.0, -0.01)),             # one 64-row stripe ... several
]


@pytest.mark.parametrize("W,H,K,dist", LENSES)
def test_undistortion_is_bit_exact_with_the_restated_opencv_algorithm(W, H, K, dist, restated):
    K = np.asarray(K, np.float32)
    newK, roi = colmap.optimal_new_camera_matrix(K, dist, W, H)
    newK_o, roi_o = restated.optimal_new_camera_matrix(K, dist, W, H)
    assert newK.dtype == np.float32 and np.array_equal(newK, newK_o), (newK, newK_o)
    assert tuple(roi) == tuple(roi_o), (roi, roi_o)
    assert roi[2] > W // 2 and roi[3] > H // 2
    img = _lens_capture(W, H, K, dist, seed=W + H)
    got = colmap.undistort_image(img, K, dist, newK)
    want = restated.undistort(img, K, dist, newK)
    assert np.array_equal(got, want), int((got != want).sum())
    # with alpha = 1 (all source pixels kept) parts of the frame lie outside the source: zero border
    K1, _ = colmap.optimal_new_camera_matrix(K, dist, W, H, alpha=1.0)
    K1o, _ = restated.optimal_new_camera_matrix(K, dist, W, H, alpha=1.0)
    assert np.array_equal(K1, K1o)
    g1, w1 = colmap.undistort_image(img, K, dist, K1), restated.undistort(img, K, dist, K1)
    assert np.array_equal(g1, w1)


def test_load_image_equals_the_restated_opencv_sequence(tmp_path, restated):
    """Camera::loadImage (input_data.cpp:40-105) on a distorted capture stored as PNG: colmap.load_image
    against the same three steps through the C restatement."""
    W, H = 322, 242
    K = np.array([[260.0, 0, 160.0], [0, 258.0, 120.5], [0, 0, 1]], np.float32)
    dist = (-0.2, 0.06, 0.001, -0.0005, 0.0)
    img = _lens_capture(W, H, K, dist, seed=5)
    path = str(tmp_path / "cap.npy")
    np.save(path, img)
    cam = colmap.Camera(width=W, height=H, fx=260.0, fy=258.0, cx=160.0, cy=120.5, k1=dist[0], k2=dist[1],
                        p1=dist[2], p2=dist[3], k3=dist[4], file_path=path)
    colmap.load_image(cam, downscale=2.0)
    f32 = np.float32
    sf = f32(1.0) / f32(2.0)
    small = restated.resize_area(img, inv_scale=float(sf))
    Ks = np.array([[f32(260.0) * sf, 0, f32(160.0) * sf], [0, f32(258.0) * sf, f32(120.5) * sf], [0, 0, 1]], np.float32)
    newK, roi = restated.optimal_new_camera_matrix(Ks, dist, small.shape[1], small.shape[0])
    und = restated.undistort(small, Ks, dist, newK)
    x, y, w, h = roi
    want = und[y:y + h, x:x + w].astype(np.float32) / np.float32(255.0)
    assert cam.image.shape == want.shape and np.array_equal(cam.image, want)
    assert (cam.fx, cam.fy, cam.cx, cam.cy) == tuple(float(v) for v in (newK[0, 0], newK[1, 1], newK[0, 2], newK[1, 2]))


# ---- EXIF orientation, non-baseline JPEGs ----------------------------------------------------------------

def _with_exif(jpeg: bytes, orientation: int, big_endian: bool) -> bytes:
    import struct

    e = ">" if big_endian else "<"
    tiff = (b"MM" if big_endian else b"II") + struct.pack(e + "HI", 42, 8) + struct.pack(e + "H", 2)
    tiff += struct.pack(e + "HHIHH", 0x010F, 3, 1, 7, 0)                 # some other tag first
    tiff += struct.pack(e + "HHIHH", 0x0112, 3, 1, orientation, 0) + struct.pack(e + "I", 0)
    seg = b"Exif\x00\x00" + tiff
    return jpeg[:2] + b"\xff\xe1" + struct.pack(">H", len(seg) + 2) + seg + jpeg[2:]


@pytest.mark.parametrize("big_endian", [False, True])
def test_exif_orientation_is_applied_like_cv_imread(tmp_path, big_endian):
    g = np.load(GOLD)
    blob, rgb = g["q90_422_file"].tobytes(), g["q90_422_rgb"]
    assert colmap.exif_orientation(blob) == 1
    T = lambda a: a.transpose(1, 0, 2)
    want = {1: rgb, 2: rgb[:, ::-1], 3: rgb[::-1, ::-1], 4: rgb[::-1], 5: T(rgb), 6: T(rgb)[:, ::-1],
            7: T(rgb)[::-1, ::-1], 8: T(rgb)[::-1]}
    for o, ref in want.items():
        b = _with_exif(blob, o, big_endian)
        assert colmap.exif_orientation(b) == o
        path = str(tmp_path / ("o%d.jpg" % o))
        open(path, "wb").write(b)
        assert np.array_equal(colmap.read_image_u8(path), ref), o
    # orientation 6 = the camera was held rotated: a 90-degree clockwise turn brings the picture upright
    assert np.array_equal(want[6], np.rot90(rgb, -1))
    assert colmap.exif_orientation(_with_exif(blob, 9, big_endian)) == 1     # out of range: ignored
    assert colmap.exif_orientation(blob[:2] + b"\xff\xe1\x00\x08Exif\x00\x00" + blob[2:]) == 1


def test_progressive_jpeg_is_read_natively(tmp_path):
    """read_image_u8 on a progressive file: the own decoder (no Pillow needed since round 3)."""
    g = np.load(GOLD)
    path = str(tmp_path / "prog.jpg")
    open(path, "wb").write(g["prog_q85_420_file"].tobytes())
    assert np.array_equal(colmap.read_image_u8(path), g["prog_q85_420_rgb"])


def test_cmyk_jpeg_goes_to_pillow_or_says_what_to_do(tmp_path):
    """What the own decoder refuses by name (here: a four-component CMYK file) is handed to Pillow when that
    is installed — the same libjpeg cv::imread uses — and refused with instructions otherwise."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image

    rs = np.random.RandomState(3)
    path = str(tmp_path / "cmyk.jpg")
    Image.fromarray(rs.randint(0, 256, (24, 32, 4)).astype(np.uint8), "CMYK").save(path, "JPEG", quality=90)
    with pytest.raises(ValueError, match="unsupported"):
        colmap.decode_jpeg(open(path, "rb").read())
    ref = np.asarray(Image.open(path).convert("RGB"))
    assert np.array_equal(colmap.read_image_u8(path), ref)
