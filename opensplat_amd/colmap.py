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


def exif_orientation(blob: bytes) -> int:
    """EXIF Orientation (tag 0x0112 of IFD0 in a JPEG's APP1 "Exif" segment), 1 when absent or
    malformed — what cv::imread reads before rotating the decoded picture (grfmt_jpeg / exif.cpp)."""
    import struct

    pos, n = 2, len(blob)
    while pos + 4 <= n and blob[pos] == 0xFF:
        marker = blob[pos + 1]
        if marker in (0xD8, 0x01) or 0xD0 <= marker <= 0xD7:
            pos += 2
            continue
        if marker == 0xDA or marker == 0xD9:            # start of scan / end: no EXIF further on
            break
        seglen = struct.unpack(">H", blob[pos + 2:pos + 4])[0]
        seg = blob[pos + 4:pos + 2 + seglen]
        pos += 2 + seglen
        if marker != 0xE1 or seg[:6] != b"Exif\x00\x00":
            continue
        t = seg[6:]
        if len(t) < 8 or t[:2] not in (b"II", b"MM"):
            return 1
        e = "<" if t[:2] == b"II" else ">"
        if struct.unpack(e + "H", t[2:4])[0] != 42:
            return 1
        ifd = struct.unpack(e + "I", t[4:8])[0]
        if ifd + 2 > len(t):
            return 1
        cnt = struct.unpack(e + "H", t[ifd:ifd + 2])[0]
        for k in range(cnt):
            ent = t[ifd + 2 + 12 * k: ifd + 14 + 12 * k]
            if len(ent) < 12:
                return 1
            tag, typ, num = struct.unpack(e + "HHI", ent[:8])
            if tag == 0x0112 and typ == 3 and num == 1:
                v = struct.unpack(e + "H", ent[8:10])[0]
                return v if 1 <= v <= 8 else 1
        return 1
    return 1


def apply_exif_orientation(img: np.ndarray, orientation: int) -> np.ndarray:
    """OpenCV's ExifTransform (imgcodecs/src/loadsave.cpp): transpose and / or flips per orientation."""
    if orientation == 2:
        img = img[:, ::-1]
    elif orientation == 3:
        img = img[::-1, ::-1]
    elif orientation == 4:
        img = img[::-1]
    elif orientation == 5:
        img = img.transpose(1, 0, 2)
    elif orientation == 6:
        img = img.transpose(1, 0, 2)[:, ::-1]
    elif orientation == 7:
        img = img.transpose(1, 0, 2)[::-1, ::-1]
    elif orientation == 8:
        img = img.transpose(1, 0, 2)[::-1]
    return np.ascontiguousarray(img)


def _decode_jpeg_any(blob: bytes) -> np.ndarray:
    """Sequential and progressive Huffman JPEGs go through the in-tree decoder (bit-exact against libjpeg).
    Arithmetic-coded, lossless, 12-bit or CMYK files — which it refuses — are handed to Pillow when that is
    installed (it wraps the same libjpeg cv::imread uses); without it the error says what to do."""
    try:
        return decode_jpeg(blob)
    except ValueError as e:
        if "unsupported" not in str(e):
            raise                                      # corrupt / truncated: nothing to hand over
        try:
            import io

            from PIL import Image
        except ImportError:
            raise ValueError("%s — this file needs a full JPEG decoder: install Pillow, or re-encode the "
                             "capture as baseline JPEG / PNG" % e) from None
        try:
            im = Image.open(io.BytesIO(blob))
            return np.asarray(im.convert("RGB"), dtype=np.uint8)
        except Exception as pe:                        # Pillow's own exception types -> ours
            raise ValueError("JPEG: %s" % pe) from None


def read_image_u8(path: str) -> np.ndarray:
    """[H, W, 3] uint8 RGB from JPEG, PNG, binary PPM (P6) or .npy (uint8 or float in [0, 1]) —
    imreadRGB (cv_utils.cpp:3-14) without OpenCV, including the EXIF orientation cv::imread applies to
    JPEGs."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        a = np.load(path)
        return a if a.dtype == np.uint8 else np.clip(np.rint(a * 255.0), 0, 255).astype(np.uint8)
    blob = open(path, "rb").read()
    if ext in (".jpg", ".jpeg", ".jpe") or blob[:2] == b"\xff\xd8":
        return apply_exif_orientation(_decode_jpeg_any(blob), exif_orientation(blob))
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
    """Camera::getImage's pyramid level (input_data.cpp:112): cv::resize to (cols / factor, rows / factor)
    with INTER_AREA."""
    if factor <= 1:
        return img_u8
    return resize_area(img_u8, img_u8.shape[1] // factor, img_u8.shape[0] // factor)


def _area_table(ssize: int, dsize: int, scale: float):
    """computeResizeAreaTab (OpenCV resize.cpp): (dst index, src index, weight) triples in OpenCV's
    order; the weights are float32 like OpenCV's."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = int(np.ceil(fsx1)), int(np.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def _round_half_even_u8(a):
    return np.clip(np.rint(a), 0, 255).astype(np.uint8)          # saturate_cast<uchar>(float) = cvRound


def resize_area(img_u8: np.ndarray, dst_w: int | None = None, dst_h: int | None = None,
                inv_scale: float | None = None) -> np.ndarray:
    """cv::resize(src, dst, dsize, inv_scale, inv_scale, INTER_AREA) for down-scaling, [H, W, 3] uint8.

    Either the destination size is given (Camera::getImage, input_data.cpp:112) or the scale factor
    (Camera::loadImage: cv::Size(), 1 / downscaleFactor as float, input_data.cpp:54-61; the size is then
    cvRound(src * inv_scale)).  Follows OpenCV 4.5.4's resize.cpp step for step: integer scale ratios
    (|scale - round(scale)| < DBL_EPSILON) take ResizeAreaFast — (a + b + c + d + 2) >> 2 for 2 x 2
    boxes (the SIMD path), float32 sum * (1 / area) rounded half-to-even otherwise, partial boxes at the
    right / bottom edge averaged over the pixels they hold; every other ratio takes ResizeArea with
    computeResizeAreaTab's float32 weights, accumulated in float32 in OpenCV's order.
    PINNED TO THE PUBLISHED ALGORITHM (oracle/image_oracle.c restates the same sources in C and
    tests/test_image.py requires bit equality); no OpenCV build was available to pin against."""
    sh, sw, cn = img_u8.shape
    if dst_w is None or dst_h is None or dst_w <= 0 or dst_h <= 0:
        inv_x = inv_y = float(inv_scale)
        dst_w, dst_h = int(np.rint(sw * inv_x)), int(np.rint(sh * inv_y))
    else:
        inv_x, inv_y = dst_w / float(sw), dst_h / float(sh)
    sx, sy = 1.0 / inv_x, 1.0 / inv_y
    if not (sx >= 1 and sy >= 1):
        raise ValueError("resize_area only down-scales (INTER_AREA up-scaling is bilinear in OpenCV)")
    ix, iy = int(np.rint(sx)), int(np.rint(sy))
    eps = np.finfo(np.float64).eps
    out = np.zeros((dst_h, dst_w, cn), np.uint8)
    if abs(sx - ix) < eps and abs(sy - iy) < eps:
        area = ix * iy
        scale = np.float32(1.0) / np.float32(area)
        fw = min(sw // ix, dst_w)                     # columns whose box is complete
        fh = min(sh // iy, dst_h)                     # rows whose box is complete (w = dwidth1 there)
        if fw and fh:
            box = img_u8[:fh * iy, :fw * ix].astype(np.int32).reshape(fh, iy, fw, ix, cn).sum((1, 3))
            out[:fh, :fw] = ((box + 2) >> 2).astype(np.uint8) if (ix == 2 and iy == 2) else \
                _round_half_even_u8(box.astype(np.float32) * scale)
        src = img_u8.astype(np.int32)

        def partial(dy0, dy1, dx0, dx1):              # boxes clipped by the image edge: mean of what is there
            for dy in range(dy0, dy1):
                y0 = dy * iy
                if y0 >= sh:
                    continue                          # (row stays zero)
                ys = src[y0:min(y0 + iy, sh)]
                for dx in range(dx0, dx1):
                    x0 = dx * ix
                    if x0 >= sw:
                        continue
                    blk = ys[:, x0:min(x0 + ix, sw)]
                    cnt = blk.shape[0] * blk.shape[1]
                    out[dy, dx] = _round_half_even_u8(blk.sum((0, 1)).astype(np.float32) / np.float32(cnt))
        partial(0, fh, fw, dst_w)                     # right edge of the complete rows
        partial(fh, dst_h, 0, dst_w)                  # rows below the last complete one: every box partial
        return out
    xtab = _area_table(sw, dst_w, sx)
    ytab = _area_table(sh, dst_h, sy)
    src = img_u8.astype(np.float32)
    buf = np.zeros((sh, dst_w, cn), np.float32)       # the x pass of every source row
    for dx, si, alpha in xtab:
        buf[:, dx] += src[:, si] * alpha
    acc = np.zeros((dst_h, dst_w, cn), np.float32)
    first = np.ones(dst_h, bool)
    for dy, si, beta in ytab:
        if first[dy]:
            acc[dy] = beta * buf[si]
            first[dy] = False
        else:
            acc[dy] += beta * buf[si]
    return _round_half_even_u8(acc)


# ---- lens undistortion (cv::getOptimalNewCameraMatrix + cv::undistort, input_data.cpp:66-80) ----------
def _dist8(dist):
    """(k1, k2, p1, p2, k3[, k4, k5, k6]) -> 8 float64 coefficients that went through float32, like
    Camera::undistortionParameters' std::vector<float> (input_data.cpp:123-126)."""
    d = np.zeros(8, np.float64)
    d[:len(dist)] = np.asarray(dist, np.float32).astype(np.float64)
    return d


def _distort(x, y, dist):
    """Brown-Conrady model as initUndistortRectifyMap evaluates it: normalised pinhole (x, y) ->
    distorted normalised coordinates."""
    k1, k2, p1, p2, k3, k4, k5, k6 = _dist8(dist)
    x2, y2 = x * x, y * y
    r2, _2xy = x2 + y2, 2 * x * y
    kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
    xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)
    yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy
    return xd, yd


def undistort_points(u, v, K, dist, newK=None, iters: int = 5):
    """cvUndistortPointsInternal with TermCriteria(COUNT, 5): distorted pixel coordinates -> ideal pixel
    coordinates of camera newK (normalised coordinates when newK is None).  float64 throughout."""
    k = _dist8(dist)
    K = np.asarray(K, np.float64)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    ifx, ify = 1.0 / fx, 1.0 / fy
    u, v = np.asarray(u, np.float64), np.asarray(v, np.float64)
    x0, y0 = (u - cx) * ifx, (v - cy) * ify
    x, y = x0.copy(), y0.copy()
    done = np.zeros(x.shape, bool)                   # points that hit icdist < 0 keep their start value
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
        bad = (icdist < 0) & ~done
        dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
        dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
        xn, yn = (x0 - dx) * icdist, (y0 - dy) * icdist
        x = np.where(done, x, np.where(bad, x0, xn))
        y = np.where(done, y, np.where(bad, y0, yn))
        done |= bad
    if newK is None:
        return x, y
    P = np.asarray(newK, np.float64)
    xx, yy = P[0, 0] * x + P[0, 1] * y + P[0, 2], P[1, 0] * x + P[1, 1] * y + P[1, 2]
    ww = 1.0 / (P[2, 0] * x + P[2, 1] * y + P[2, 2])
    return xx * ww, yy * ww


def _inner_outer_rect(K, dist, newK, W, H, n: int = 9):
    """icvGetRectangles (OpenCV 4.5.4 calibration.cpp): the 9 x 9 grid over [0, W-1] x [0, H-1]
    undistorted; inner = largest rectangle inside the undistorted border, outer = its bounding box.
    (x, y, w, h) each, float64."""
    gx, gy = np.meshgrid(np.arange(n, dtype=np.float64) * (W - 1) / (n - 1),
                         np.arange(n, dtype=np.float64) * (H - 1) / (n - 1))
    px, py = undistort_points(gx, gy, K, dist, newK)
    ix0, ix1 = px[:, 0].max(), px[:, -1].min()
    iy0, iy1 = py[0, :].max(), py[-1, :].min()
    ox0, ox1, oy0, oy1 = px.min(), px.max(), py.min(), py.max()
    return (ix0, iy0, ix1 - ix0, iy1 - iy0), (ox0, oy0, ox1 - ox0, oy1 - oy0)


def optimal_new_camera_matrix(K, dist, W: int, H: int, alpha: float = 0.0):
    """cv::getOptimalNewCameraMatrix(K, dist, (W, H), alpha, Size(), &roi) as OpenCV 4.5.4 computes it:
    the camera matrix whose image shows (alpha = 0) only valid pixels of the undistorted image —
    returned in K's type, float32 — and the valid-pixel ROI (x, y, w, h): the inner rectangle under the
    float64 matrix, each component rounded to nearest (cv::Rect r = inner), clipped to the image.
    PINNED TO THE PUBLISHED ALGORITHM (oracle/image_oracle.c)."""
    K = np.asarray(K, np.float32).astype(np.float64)
    inner, outer = _inner_outer_rect(K, dist, None, W, H)
    fx0, fy0 = (W - 1) / inner[2], (H - 1) / inner[3]
    cx0, cy0 = -fx0 * inner[0], -fy0 * inner[1]
    fx1, fy1 = (W - 1) / outer[2], (H - 1) / outer[3]
    cx1, cy1 = -fx1 * outer[0], -fy1 * outer[1]
    M = K.copy()
    M[0, 0] = fx0 * (1 - alpha) + fx1 * alpha
    M[1, 1] = fy0 * (1 - alpha) + fy1 * alpha
    M[0, 2] = cx0 * (1 - alpha) + cx1 * alpha
    M[1, 2] = cy0 * (1 - alpha) + cy1 * alpha
    inner2, _ = _inner_outer_rect(K, dist, M, W, H)
    x, y, w, h = (int(np.rint(c)) for c in inner2)
    x0, y0 = max(x, 0), max(y, 0)
    x1, y1 = min(x + w, W), min(y + h, H)
    roi = (x0, y0, x1 - x0, y1 - y0) if (x1 > x0 and y1 > y0) else (0, 0, 0, 0)
    return M.astype(np.float32), roi


def _invert3(S):
    """cv::invert for a 3 x 3 double matrix (closed form, lapack.cpp)."""
    S = S.ravel()
    d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6])
    d = 1.0 / d
    return np.array([(S[4] * S[8] - S[5] * S[7]) * d, (S[2] * S[7] - S[1] * S[8]) * d, (S[1] * S[5] - S[2] * S[4]) * d,
                     (S[5] * S[6] - S[3] * S[8]) * d, (S[0] * S[8] - S[2] * S[6]) * d, (S[2] * S[3] - S[0] * S[5]) * d,
                     (S[3] * S[7] - S[4] * S[6]) * d, (S[1] * S[6] - S[0] * S[7]) * d, (S[0] * S[4] - S[1] * S[3]) * d])


def undistort_image(img_u8: np.ndarray, K, dist, newK) -> np.ndarray:
    """cv::undistort(src, dst, K, dist, newK) as OpenCV 4.5.4 computes it: row stripes of
    (1 << 12) / W rows, per stripe initUndistortRectifyMap with the new matrix shifted to the stripe
    (its inverse in closed form, the ray of pixel j accumulated from pixel 0 in float64), source
    coordinates rounded to 1/32 pixel and split into an int16 pixel and a 5 + 5 bit fraction, then
    remap(INTER_LINEAR, BORDER_CONSTANT 0) with the fixed-point weights (2^15 scale).
    PINNED TO THE PUBLISHED ALGORITHM (oracle/image_oracle.c)."""
    H, W, cn = img_u8.shape
    A = np.asarray(K, np.float32).astype(np.float64)
    Ar = np.asarray(newK, np.float32).astype(np.float64).copy()
    fx, fy, u0, v0a = A[0, 0], A[1, 1], A[0, 2], A[1, 2]
    stripe0 = min(max(1, (1 << 12) // max(W, 1)), H)
    v0 = Ar[1, 2]
    src = np.zeros((H + 2, W + 2, cn), np.int64)                         # zero border
    src[1:-1, 1:-1] = img_u8
    out = np.empty_like(img_u8)
    jj = np.arange(W)
    for y in range(0, H, stripe0):
        n = min(stripe0, H - y)
        Ar[1, 2] = v0 - y
        ir = _invert3(Ar)
        i = np.arange(n, dtype=np.float64)[:, None]

        def ray(a, b, step):                                              # _x = i * a + b, then += step per pixel
            first = i * a + b
            seq = np.concatenate([first, np.full((n, W - 1), step)], axis=1)
            return np.cumsum(seq, axis=1)
        _x, _y, _w = ray(ir[1], ir[2], ir[0]), ray(ir[4], ir[5], ir[3]), ray(ir[7], ir[8], ir[6])
        w = 1.0 / _w
        x, yy = _x * w, _y * w
        xd, yd = _distort(x, yy, dist)
        u, v = fx * xd + u0, fy * yd + v0a
        iu, iv = np.rint(u * 32).astype(np.int64), np.rint(v * 32).astype(np.int64)
        wrap = lambda a: ((a + 32768) & 0xFFFF) - 32768                  # (short)
        sx, sy = wrap(iu >> 5), wrap(iv >> 5)
        fxq, fyq = iu & 31, iv & 31

        def tap(ty, tx):
            ok = (ty >= 0) & (ty < H) & (tx >= 0) & (tx < W)
            return src[np.clip(ty + 1, 0, H + 1), np.clip(tx + 1, 0, W + 1)] * ok[..., None]
        w00 = ((32 - fxq) * (32 - fyq) * 32)[..., None]
        w01 = (fxq * (32 - fyq) * 32)[..., None]
        w10 = ((32 - fxq) * fyq * 32)[..., None]
        w11 = (fxq * fyq * 32)[..., None]
        acc = tap(sy, sx) * w00 + tap(sy, sx + 1) * w01 + tap(sy + 1, sx) * w10 + tap(sy + 1, sx + 1) * w11
        out[y:y + n] = np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
    return out


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
        img = resize_area(img, inv_scale=float(sf))        # cv::resize(img, img, Size(), sf, sf, INTER_AREA)
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
