"""COLMAP ingest and model initialisation (host code like the reference's).

  read_colmap           cm::inputDataFromColmap (colmap.cpp:10-155): cameras.bin / images.bin /
                        points3D.bin (also under sparse/0), world-to-camera quaternion + translation
                        inverted to camera-to-world, OpenCV -> OpenGL camera axes, poses centred and
                        scaled by autoScaleAndCenterPoses (tensor_math.cpp:30-45), points moved with them
  render_camera         the camera block of Model::forward (model.cpp:85-113): down-scaled intrinsics,
                        y / z flip, world-to-camera view matrix, fov, projection matrix
  init_from_points      Model's constructor (model.hpp:33-53): means = points, log-scales from the mean
                        distance to the three nearest neighbours (kdtree_tensor.cpp:4-23), random
                        quaternions from torch's CPU generator seeded 42 (model.cpp:23-33) — the same
                        stream, hence the same quaternions, as the reference — featuresDc = rgb2sh,
                        opacity logit(0.1)
  load_image            Camera::loadImage without OpenCV: .npy, binary PPM and 8-bit RGB(A) PNG are
                        decoded here; intrinsics rescaled to the image as input_data.cpp:44-52 does;
                        integer down-scaling = box average.  Lens undistortion (cv::undistort,
                        input_data.cpp:66-80) is NOT implemented: cameras with distortion parameters
                        are rejected unless `ignore_distortion=True`.
  write_colmap          the inverse of read_colmap (test fixtures, synthetic captures)

Parity: quatToRotMat / autoScaleAndCenterPoses / the pose and view-matrix statements are checked
against the reference's own tensor_math.cpp compiled in place (tests/test_colmap.py,
oracle/ref_train_shim.cpp); the binary layout against COLMAP's published format through a round trip
and hand-packed records.  The OpenCV-dependent image path has no oracle here ("parity unpinned").
"""
from __future__ import annotations

import math
import os
import struct
import zlib
from dataclasses import dataclass, field

import numpy as np

# CameraModel ids, colmap.hpp
SIMPLE_PINHOLE, PINHOLE, SIMPLE_RADIAL, RADIAL, OPENCV = 0, 1, 2, 3, 4
_NUM_PARAMS = {SIMPLE_PINHOLE: 3, PINHOLE: 4, SIMPLE_RADIAL: 4, OPENCV: 8}


@dataclass
class Camera:
    id: int = -1
    width: int = 0
    height: int = 0
    fx: float = 0.0
    fy: float = 0.0
    cx: float = 0.0
    cy: float = 0.0
    k1: float = 0.0
    k2: float = 0.0
    k3: float = 0.0
    p1: float = 0.0
    p2: float = 0.0
    cam_to_world: np.ndarray | None = None   # [4, 4] float32, OpenGL axes, normalised
    file_path: str = ""
    image: np.ndarray | None = None          # [H, W, 3] float32 in [0, 1] once loaded

    def has_distortion(self) -> bool:        # input_data.cpp:117-119
        return any(v != 0.0 for v in (self.k1, self.k2, self.k3, self.p1, self.p2))


@dataclass
class InputData:
    cameras: list = field(default_factory=list)
    scale: float = 1.0
    translation: np.ndarray | None = None    # [3] float32
    points_xyz: np.ndarray | None = None     # [P, 3] float32 (normalised)
    points_rgb: np.ndarray | None = None     # [P, 3] uint8


def quat_to_rotmat(q) -> np.ndarray:
    """tensor_math.cpp:5-28 in float32 (normalises with eps 1e-12 like F::normalize)."""
    q = np.asarray(q, np.float32)
    n = np.float32(max(float(np.sqrt((q * q).sum(dtype=np.float32))), 1e-12))
    w, x, y, z = (q / n).astype(np.float32)
    f = np.float32
    return np.array([[f(1) - f(2) * (y * y + z * z), f(2) * (x * y - w * z), f(2) * (x * z + w * y)],
                     [f(2) * (x * y + w * z), f(1) - f(2) * (x * x + z * z), f(2) * (y * z - w * x)],
                     [f(2) * (x * z - w * y), f(2) * (y * z + w * x), f(1) - f(2) * (x * x + y * y)]],
                    dtype=np.float32)


def colmap_pose(qvec, tvec) -> np.ndarray:
    """One image record -> un-normalised camera-to-world pose, colmap.cpp:88-118."""
    R = quat_to_rotmat(qvec)
    T = np.asarray(tvec, np.float32).reshape(3, 1)
    Rinv = R.T
    Tinv = (-Rinv) @ T
    pose = np.zeros((4, 4), np.float32)
    pose[:3, :3] = Rinv
    pose[:3, 3:4] = Tinv
    pose[3, 3] = 1.0
    pose[:3, 1:3] *= np.float32(-1.0)   # OpenCV -> OpenGL camera axes
    return pose


def auto_scale_and_center_poses(poses: np.ndarray):
    """tensor_math.cpp:30-45 -> (poses, centre [3], scale)."""
    poses = np.array(poses, np.float32, copy=True)
    origins = poses[:, :3, 3]
    center = origins.mean(axis=0, dtype=np.float32)
    origins = origins - center
    f = np.float32(1.0) / np.float32(np.abs(origins).max())
    poses[:, :3, 3] = origins * f
    return poses, center.astype(np.float32), float(f)


def _read(fmt, f):
    size = struct.calcsize(fmt)
    b = f.read(size)
    if len(b) != size:
        raise ValueError("unexpected end of COLMAP file")
    return struct.unpack(fmt, b)


def read_colmap(project_root: str, image_source: str | None = None) -> InputData:
    root = project_root
    if not os.path.exists(os.path.join(root, "cameras.bin")) and \
            os.path.exists(os.path.join(root, "sparse", "0", "cameras.bin")):
        root = os.path.join(root, "sparse", "0")
    paths = {n: os.path.join(root, n) for n in ("cameras.bin", "images.bin", "points3D.bin")}
    for p in paths.values():
        if not os.path.exists(p):
            raise FileNotFoundError(p + " does not exist")
    cams = {}
    with open(paths["cameras.bin"], "rb") as f:
        (n_cams,) = _read("<Q", f)
        for _ in range(n_cams):
            cid, model, w, h = _read("<IiQQ", f)
            if model not in _NUM_PARAMS:
                raise ValueError("Unsupported camera model: %d" % model)
            p = _read("<%dd" % _NUM_PARAMS[model], f)
            c = Camera(id=cid, width=int(w), height=int(h))
            if model == SIMPLE_PINHOLE:
                c.fx = c.fy = p[0]; c.cx, c.cy = p[1], p[2]
            elif model == PINHOLE:
                c.fx, c.fy, c.cx, c.cy = p
            elif model == SIMPLE_RADIAL:
                c.fx = c.fy = p[0]; c.cx, c.cy, c.k1 = p[1], p[2], p[3]
            else:
                c.fx, c.fy, c.cx, c.cy, c.k1, c.k2, c.p1, c.p2 = p
            for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2"):
                setattr(c, k, float(np.float32(getattr(c, k))))    # Camera holds floats
            cams[cid] = c
    out = InputData()
    poses = []
    with open(paths["images.bin"], "rb") as f:
        (n_img,) = _read("<Q", f)
        for _ in range(n_img):
            _read("<I", f)                                  # image id
            q = _read("<4d", f)
            t = _read("<3d", f)
            (cam_id,) = _read("<I", f)
            name = bytearray()
            while True:
                ch = f.read(1)
                if ch in (b"\0", b""):
                    break
                name += ch
            (n2d,) = _read("<Q", f)
            f.seek(n2d * 24, os.SEEK_CUR)                   # x, y, point3D id
            cam = Camera(**{k: v for k, v in cams[cam_id].__dict__.items()})
            base = image_source if image_source else os.path.join(project_root, "images")
            cam.file_path = os.path.join(base, name.decode())
            poses.append(colmap_pose(q, t))
            out.cameras.append(cam)
    norm, center, scale = auto_scale_and_center_poses(np.stack(poses)) if poses else \
        (np.zeros((0, 4, 4), np.float32), np.zeros(3, np.float32), 1.0)
    for c, pose in zip(out.cameras, norm):
        c.cam_to_world = pose
    out.translation, out.scale = center, scale
    xyz, rgb = [], []
    with open(paths["points3D.bin"], "rb") as f:            # point_io.cpp:361-392
        (n_pts,) = _read("<Q", f)
        for _ in range(n_pts):
            _pid, x, y, z, r, g, b, _err, track = _read("<Q3d3BdQ", f)
            f.seek(track * 8, os.SEEK_CUR)
            xyz.append((x, y, z)); rgb.append((r, g, b))
    pts = np.asarray(xyz, np.float32).reshape(-1, 3)
    out.points_xyz = ((pts - center) * np.float32(scale)).astype(np.float32)   # colmap.cpp:148
    out.points_rgb = np.asarray(rgb, np.uint8).reshape(-1, 3)
    return out


def write_colmap(root: str, cameras: list, world_to_cam: list, points_xyz, points_rgb, names=None):
    """cameras: list of Camera (PINHOLE, or OPENCV when distorted); world_to_cam: list of
    (qvec [w,x,y,z], tvec) in COLMAP's convention; writes <root>/sparse/0/*.bin."""
    d = os.path.join(root, "sparse", "0")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(cameras)))
        for c in cameras:
            if c.has_distortion():
                f.write(struct.pack("<IiQQ8d", c.id, OPENCV, c.width, c.height, c.fx, c.fy, c.cx, c.cy,
                                    c.k1, c.k2, c.p1, c.p2))
            else:
                f.write(struct.pack("<IiQQ4d", c.id, PINHOLE, c.width, c.height, c.fx, c.fy, c.cx, c.cy))
    with open(os.path.join(d, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(world_to_cam)))
        for i, (q, t) in enumerate(world_to_cam):
            f.write(struct.pack("<I4d3dI", i + 1, *[float(v) for v in q], *[float(v) for v in t],
                                cameras[i % len(cameras)].id if len(cameras) != len(world_to_cam)
                                else cameras[i].id))
            f.write((names[i] if names else "%05d.npy" % i).encode() + b"\0")
            f.write(struct.pack("<Q", 0))
    with open(os.path.join(d, "points3D.bin"), "wb") as f:
        P = len(points_xyz)
        f.write(struct.pack("<Q", P))
        for i in range(P):
            x, y, z = [float(v) for v in points_xyz[i]]
            r, g, b = [int(v) for v in points_rgb[i]]
            f.write(struct.pack("<Q3d3BdQ", i + 1, x, y, z, r, g, b, 0.0, 0))


def projection_matrix(znear, zfar, fovx, fovy) -> np.ndarray:
    """model.cpp:35-47 (float arithmetic)."""
    f = np.float32
    t = f(znear) * f(math.tan(0.5 * fovy)); b = -t
    r = f(znear) * f(math.tan(0.5 * fovx)); l = -r
    return np.array([[f(2) * f(znear) / (r - l), 0, (r + l) / (r - l), 0],
                     [0, f(2) * f(znear) / (t - b), (t + b) / (t - b), 0],
                     [0, 0, (f(zfar) + f(znear)) / (f(zfar) - f(znear)),
                      f(-1.0) * f(zfar) * f(znear) / (f(zfar) - f(znear))],
                     [0, 0, 1, 0]], dtype=np.float32)


def render_camera(cam: Camera, downscale: float = 1.0) -> dict:
    """Model::forward's camera set-up (model.cpp:85-113) -> the dict Trainer.render takes
    (projmat = proj @ view, model.cpp:152)."""
    s = np.float32(downscale)
    fx, fy = np.float32(cam.fx) / s, np.float32(cam.fy) / s
    cx, cy = np.float32(cam.cx) / s, np.float32(cam.cy) / s
    H, W = int(np.float32(cam.height) / s), int(np.float32(cam.width) / s)
    R = cam.cam_to_world[:3, :3] @ np.diag(np.array([1.0, -1.0, -1.0], np.float32))
    T = cam.cam_to_world[:3, 3:4]
    Rinv = R.T
    view = np.eye(4, dtype=np.float32)
    view[:3, :3] = Rinv
    view[:3, 3:4] = (-Rinv) @ T
    fovx = 2.0 * math.atan(W / (2.0 * float(fx)))
    fovy = 2.0 * math.atan(H / (2.0 * float(fy)))
    proj = projection_matrix(0.001, 1000.0, fovx, fovy)
    return dict(viewmat=view, projmat=(proj @ view).astype(np.float32), fx=float(fx), fy=float(fy),
                cx=float(cx), cy=float(cy), W=W, H=H)


C0 = 0.28209479177387814


def init_from_points(xyz, rgb, sh_degree: int = 3):
    """Model's constructor (model.hpp:33-53) -> the six raw parameter arrays
    [means, log_scales, quats, opacity_logits, features_dc, features_rest]."""
    import torch
    from scipy.spatial import cKDTree

    xyz = np.ascontiguousarray(xyz, np.float32)
    n = xyz.shape[0]
    K = (sh_degree + 1) ** 2
    # PointsTensor::scales (kdtree_tensor.cpp:4-23): mean distance to the three nearest neighbours
    # (nanoflann returns squared float distances; the query point itself comes first)
    d2 = cKDTree(xyz).query(xyz, k=min(4, n))[0].astype(np.float32) ** 2
    dist = np.sqrt(d2[:, 1:]).astype(np.float32)
    scale = (dist.sum(axis=1, dtype=np.float32) / np.float32(3.0)).reshape(n, 1)
    log_scales = np.log(np.repeat(scale, 3, axis=1)).astype(np.float32)
    torch.manual_seed(42)                                   # model.hpp:37
    u, v, w = torch.rand(n), torch.rand(n), torch.rand(n)   # randomQuatTensor, model.cpp:23-33
    PI = 3.14159265358979323846
    quats = torch.stack([torch.sqrt(1 - u) * torch.sin(2 * PI * v), torch.sqrt(1 - u) * torch.cos(2 * PI * v),
                         torch.sqrt(u) * torch.sin(2 * PI * w), torch.sqrt(u) * torch.cos(2 * PI * w)], -1)
    dc = ((np.asarray(rgb, np.float64) / 255.0 - 0.5) / C0).astype(np.float32)   # rgb2sh in fp64
    rest = np.zeros((n, K - 1, 3), np.float32)
    logits = np.full((n, 1), np.float32(math.log(0.1 / 0.9)), np.float32)       # torch::logit(0.1)
    return [xyz, log_scales, quats.numpy().astype(np.float32), logits, dc, rest]


# ---- images (OpenCV-free) ------------------------------------------------------------------------

def _decode_png(blob: bytes) -> np.ndarray:
    if blob[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(blob):
        (ln,), typ = struct.unpack(">I", blob[pos:pos + 4]), blob[pos + 4:pos + 8]
        data = blob[pos + 8:pos + 8 + ln]
        pos += 12 + ln
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        elif typ == b"IDAT":
            idat.append(data)
        elif typ == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or ctype not in (2, 6) or interlace:
        raise ValueError("only 8-bit non-interlaced RGB / RGBA PNGs are supported")
    bpp = 3 if ctype == 2 else 4
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, 1 + w * bpp)
    out = np.zeros((h, w * bpp), np.uint8)
    prev = np.zeros(w * bpp, np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        cur = np.zeros(w * bpp, np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:   # Sub, Average, Paeth need the left neighbour: per byte
            for i in range(w * bpp):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ft == 1:
                    p = a
                elif ft == 3:
                    p = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + p) & 255
        out[y] = cur
        prev = cur
    return out.reshape(h, w, bpp)[:, :, :3]


_image_lib = None


def image_lib():
    """libgsplat_image.so (include/gsplat_image.h): the baseline JPEG decoder, plain host C."""
    global _image_lib
    if _image_lib is None:
        import ctypes as C

        from . import _build

        if not os.path.exists(_build.IMAGE_LIB):
            raise ImportError("libgsplat_image.so is not built: run `python -m opensplat_amd._build`")
        l = C.CDLL(_build.IMAGE_LIB)
        l.gs_image_strerror.restype = C.c_char_p
        l.gs_jpeg_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int)]
        l.gs_jpeg_decode_rgb.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
        _image_lib = l
    return _image_lib


def decode_jpeg(blob: bytes) -> np.ndarray:
    """[H, W, 3] uint8 RGB of a baseline JPEG: the pixels libjpeg (hence cv::imread, cv_utils.cpp:3-14)
    produces, bit for bit (tests/test_image.py)."""
    import ctypes as C

    l = image_lib()
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    rc = l.gs_jpeg_info(blob, len(blob), C.byref(w), C.byref(h), C.byref(c))
    if rc != 0:
        raise ValueError("JPEG: " + l.gs_image_strerror(rc).decode())
    out = np.empty((h.value, w.value, 3), np.uint8)
    rc = l.gs_jpeg_decode_rgb(blob, len(blob), out.ctypes.data_as(C.c_void_p), out.size)
    if rc != 0:
        raise ValueError("JPEG: " + l.gs_image_strerror(rc).decode())
    return out


def read_image_u8(path: str) -> np.ndarray:
    """[H, W, 3] uint8 RGB from baseline JPEG, PNG, binary PPM (P6) or .npy (uint8 or float in
    [0, 1]) — imreadRGB (cv_utils.cpp:3-14) without OpenCV."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        a = np.load(path)
        return a if a.dtype == np.uint8 else np.clip(np.rint(a * 255.0), 0, 255).astype(np.uint8)
    blob = open(path, "rb").read()
    if ext in (".jpg", ".jpeg", ".jpe") or blob[:2] == b"\xff\xd8":
        return decode_jpeg(blob)
    if ext == ".png":
        return _decode_png(blob)
    if ext in (".ppm", ".pnm") and blob[:2] == b"P6":
        parts, pos = [], 2
        while len(parts) < 3:
            while blob[pos:pos + 1].isspace():
                pos += 1
            if blob[pos:pos + 1] == b"#":
                pos = blob.index(b"\n", pos) + 1
                continue
            end = pos
            while not blob[end:end + 1].isspace():
                end += 1
            parts.append(int(blob[pos:end])); pos = end
        w, h, mx = parts
        if mx != 255:
            raise ValueError("only 8-bit PPM files are supported")
        return np.frombuffer(blob[pos + 1:pos + 1 + w * h * 3], np.uint8).reshape(h, w, 3)
    raise ValueError("unsupported image format (no OpenCV here): " + path)


def downscale_area(img_u8: np.ndarray, factor: int) -> np.ndarray:
    """Integer-factor box average (what cv::INTER_AREA computes for integer factors), uint8 out."""
    if factor <= 1:
        return img_u8
    return resize_area(img_u8, int(round(img_u8.shape[1] / factor)), int(round(img_u8.shape[0] / factor)),
                       scale=float(factor))


def _area_table(ssize: int, dsize: int, scale: float):
    """1-D INTER_AREA decomposition: for every destination index the source indices it covers and
    their weights (cell [d * scale, (d + 1) * scale), partial pixels weighted by their overlap)."""
    idx, wts = [], []
    for d in range(dsize):
        f1 = d * scale
        f2 = min(f1 + scale, float(ssize))
        cell = min(scale, ssize - f1)
        s1, s2 = int(np.ceil(f1)), int(np.floor(f2))
        s2 = min(s2, ssize - 1) if s2 >= ssize else s2
        s1 = min(s1, s2)
        ii, ww = [], []
        if s1 - f1 > 1e-3:
            ii.append(s1 - 1); ww.append((s1 - f1) / cell)
        for sx in range(s1, s2):
            ii.append(sx); ww.append(1.0 / cell)
        if f2 - s2 > 1e-3 and s2 < ssize:
            ii.append(s2); ww.append(min(min(f2 - s2, 1.0), cell) / cell)
        idx.append(ii); wts.append(ww)
    return idx, wts


def resize_area(img_u8: np.ndarray, dst_w: int, dst_h: int, scale: float | None = None) -> np.ndarray:
    """cv::resize(..., INTER_AREA) for down-scaling ([H, W, C] uint8).  scale = source pixels per
    destination pixel (both axes); default src / dst per axis, as cv::resize derives it from dsize.

    Integer scales: the mean of the scale x scale box — (sum + 2) >> 2 for 2 x 2 boxes, round-to-nearest-
    even of sum / area otherwise (the two code paths of OpenCV's ResizeAreaFast for 8-bit images);
    other scales: area-weighted mean in float32.  PARITY UNPINNED: written from OpenCV's documented
    behaviour, no OpenCV in this environment to generate fixtures from."""
    sh, sw = img_u8.shape[:2]
    sx = scale if scale is not None else sw / float(dst_w)
    sy = scale if scale is not None else sh / float(dst_h)
    ix, iy = int(round(sx)), int(round(sy))
    src = img_u8.astype(np.float32)
    if abs(sx - ix) < 1e-9 and abs(sy - iy) < 1e-9 and ix >= 1 and iy >= 1:
        fw, fh = min(dst_w, sw // ix), min(dst_h, sh // iy)
        out = np.zeros((dst_h, dst_w, img_u8.shape[2]), np.uint8)
        box = img_u8[:fh * iy, :fw * ix].astype(np.uint32).reshape(fh, iy, fw, ix, -1).sum((1, 3))
        if ix == 2 and iy == 2:
            out[:fh, :fw] = ((box + 2) >> 2).astype(np.uint8)
        else:
            out[:fh, :fw] = np.rint(box.astype(np.float32) * np.float32(1.0 / (ix * iy))).astype(np.uint8)
        # a last partial row / column (source size not a multiple of the factor): mean of what is left
        for dy in range(dst_h):
            for dx in range(dst_w):
                if dy < fh and dx < fw:
                    continue
                blk = src[dy * iy:min((dy + 1) * iy, sh), dx * ix:min((dx + 1) * ix, sw)]
                if blk.size:
                    out[dy, dx] = np.rint(blk.reshape(-1, blk.shape[-1]).sum(0) / np.float32(blk.shape[0] * blk.shape[1]))
        return out
    xi, xw = _area_table(sw, dst_w, sx)
    yi, yw = _area_table(sh, dst_h, sy)
    rows = np.zeros((sh, dst_w, img_u8.shape[2]), np.float32)
    for d in range(dst_w):
        for i, w in zip(xi[d], xw[d]):
            rows[:, d] += src[:, i] * np.float32(w)
    out = np.zeros((dst_h, dst_w, img_u8.shape[2]), np.float32)
    for d in range(dst_h):
        for i, w in zip(yi[d], yw[d]):
            out[d] += rows[i] * np.float32(w)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


# ---- lens undistortion (cv::getOptimalNewCameraMatrix + cv::undistort, input_data.cpp:66-80) ----------
def _distort(x, y, dist):
    """Brown-Conrady model OpenCV uses: normalised pinhole (x, y) -> distorted normalised coordinates.
    dist = (k1, k2, p1, p2, k3)."""
    k1, k2, p1, p2, k3 = dist
    r2 = x * x + y * y
    kr = 1.0 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * kr + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
    yd = y * kr + p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
    return xd, yd


def undistort_points(u, v, K, dist, newK=None, iters: int = 5):
    """cv::undistortPoints: distorted pixel coordinates -> ideal pixel coordinates of camera newK
    (normalised coordinates when newK is None); fixed-point iteration, 5 rounds like OpenCV."""
    k1, k2, p1, p2, k3 = dist
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x0 = (np.asarray(u, np.float64) - cx) / fx
    y0 = (np.asarray(v, np.float64) - cy) / fy
    x, y = x0.copy(), y0.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
        dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
        x = (x0 - dx) * icdist
        y = (y0 - dy) * icdist
    if newK is None:
        return x, y
    return x * newK[0, 0] + newK[0, 2], y * newK[1, 1] + newK[1, 2]


def _inner_outer_rect(K, dist, newK, W, H, n: int = 9):
    """icvGetRectangles: the 9 x 9 grid of image points undistorted; inner = largest rectangle inside
    the undistorted border, outer = its bounding box.  (x, y, w, h) each."""
    gx, gy = np.meshgrid(np.arange(n) * (W / (n - 1.0)), np.arange(n) * (H / (n - 1.0)))
    px, py = undistort_points(gx, gy, K, dist, newK)
    ix0, ix1 = px[:, 0].max(), px[:, -1].min()
    iy0, iy1 = py[0, :].max(), py[-1, :].min()
    ox0, ox1, oy0, oy1 = px.min(), px.max(), py.min(), py.max()
    return (ix0, iy0, ix1 - ix0, iy1 - iy0), (ox0, oy0, ox1 - ox0, oy1 - oy0)


def optimal_new_camera_matrix(K, dist, W: int, H: int, alpha: float = 0.0):
    """cv::getOptimalNewCameraMatrix(K, dist, (W, H), alpha, (W, H), &roi): the camera matrix whose
    image shows (alpha = 0) only valid pixels of the undistorted image, and the valid-pixel ROI
    (x, y, w, h).  PARITY UNPINNED (see resize_area)."""
    K = np.asarray(K, np.float64)
    inner, outer = _inner_outer_rect(K, dist, None, W, H)
    fx0, fy0 = (W - 1) / inner[2], (H - 1) / inner[3]
    cx0, cy0 = -fx0 * inner[0], -fy0 * inner[1]
    fx1, fy1 = (W - 1) / outer[2], (H - 1) / outer[3]
    cx1, cy1 = -fx1 * outer[0], -fy1 * outer[1]
    newK = np.eye(3)
    newK[0, 0] = fx0 * (1 - alpha) + fx1 * alpha
    newK[1, 1] = fy0 * (1 - alpha) + fy1 * alpha
    newK[0, 2] = cx0 * (1 - alpha) + cx1 * alpha
    newK[1, 2] = cy0 * (1 - alpha) + cy1 * alpha
    inner2, _ = _inner_outer_rect(K, dist, newK, W, H)
    x, y = int(np.ceil(inner2[0])), int(np.ceil(inner2[1]))
    w, h = int(np.floor(inner2[2])), int(np.floor(inner2[3]))
    x0, y0 = max(x, 0), max(y, 0)
    x1, y1 = min(x + w, W), min(y + h, H)
    roi = (x0, y0, max(x1 - x0, 0), max(y1 - y0, 0))
    return newK.astype(np.float32), roi


def undistort_image(img_u8: np.ndarray, K, dist, newK) -> np.ndarray:
    """cv::undistort(src, dst, K, dist, newK): for every pixel of the ideal camera newK the source
    position under the distortion model, sampled bilinearly with the source coordinates quantised
    to 1/32 pixel (OpenCV's fixed-point remap) and a zero border.  PARITY UNPINNED."""
    H, W = img_u8.shape[:2]
    K = np.asarray(K, np.float64)
    newK = np.asarray(newK, np.float64)
    uu, vv = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    x = (uu - newK[0, 2]) / newK[0, 0]
    y = (vv - newK[1, 2]) / newK[1, 1]
    xd, yd = _distort(x, y, dist)
    su = K[0, 0] * xd + K[0, 2]
    sv = K[1, 1] * yd + K[1, 2]
    iu = np.rint(np.clip(su, -4.0, W + 4.0) * 32.0).astype(np.int64)     # INTER_TAB_SIZE = 32
    iv = np.rint(np.clip(sv, -4.0, H + 4.0) * 32.0).astype(np.int64)
    x0, y0, fxq, fyq = iu >> 5, iv >> 5, iu & 31, iv & 31
    src = np.zeros((H + 2, W + 2, img_u8.shape[2]), np.int64)            # zero border
    src[1:-1, 1:-1] = img_u8

    def tap(yy, xx):
        ok = (yy >= -1) & (yy <= H) & (xx >= -1) & (xx <= W)
        return src[np.clip(yy + 1, 0, H + 1), np.clip(xx + 1, 0, W + 1)] * ok[..., None]
    w00 = ((32 - fxq) * (32 - fyq))[..., None]
    w01 = (fxq * (32 - fyq))[..., None]
    w10 = ((32 - fxq) * fyq)[..., None]
    w11 = (fxq * fyq)[..., None]
    acc = tap(y0, x0) * w00 + tap(y0, x0 + 1) * w01 + tap(y0 + 1, x0) * w10 + tap(y0 + 1, x0 + 1) * w11
    return ((acc + 512) >> 10).astype(np.uint8)


def load_image(cam: Camera, downscale: float = 1.0, ignore_distortion: bool = False) -> None:
    """Camera::loadImage (input_data.cpp:40-105): reads the file, rescales the intrinsics to the image
    actually found, down-scales (INTER_AREA), undistorts when the camera has distortion parameters
    (optimal new camera matrix at alpha = 0, crop to the valid ROI) and fills cam.image
    ([H, W, 3] float32 / 255) and the final intrinsics."""
    img = read_image_u8(cam.file_path)
    f32 = np.float32
    rescale = f32(1.0)
    if img.shape[0] != cam.height or img.shape[1] != cam.width:
        rescale = f32(img.shape[0]) / f32(cam.height)
    fx, fy, cx, cy = (f32(getattr(cam, k)) * rescale for k in ("fx", "fy", "cx", "cy"))
    if downscale > 1.0:
        sf = f32(1.0) / f32(downscale)
        dw, dh = int(np.rint(img.shape[1] * float(sf))), int(np.rint(img.shape[0] * float(sf)))
        img = resize_area(img, dw, dh, scale=float(downscale))
        fx, fy, cx, cy = fx * sf, fy * sf, cx * sf, cy * sf
    if cam.has_distortion() and not ignore_distortion:
        K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float32)
        dist = (cam.k1, cam.k2, cam.p1, cam.p2, cam.k3)
        newK, roi = optimal_new_camera_matrix(K, dist, img.shape[1], img.shape[0])
        img = undistort_image(img, K, dist, newK)
        x, y, w, h = roi
        img = img[y:y + h, x:x + w]
        fx, fy, cx, cy = newK[0, 0], newK[1, 1], newK[0, 2], newK[1, 2]
        cam.k1 = cam.k2 = cam.k3 = cam.p1 = cam.p2 = 0.0     # the loaded image is an ideal pinhole view
    cam.fx, cam.fy, cam.cx, cam.cy = float(fx), float(fy), float(cx), float(cy)
    cam.height, cam.width = int(img.shape[0]), int(img.shape[1])
    cam.image = (img.astype(np.float32) / np.float32(255.0))
