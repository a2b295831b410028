"""Seeded synthetic scenes for the parity tests and bench.py.

Pure numpy/torch-CPU generators: the same bits go to the CPU oracle and (after .to(device)) to the
HIP kernels.  Nothing here touches the oracle or the native libraries.

  simple_trainer_scene  C1: OpenSplat's own synthetic set-up (simple_trainer.cpp:79-146)
  camera_scene          C2/C3-style: camera at the origin looking down +z, Gaussians with a
                        controlled pixel-space footprint, SH colours
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

BACKGROUND = (0.6130, 0.0101, 0.3984)  # model.hpp:54


@dataclass
class Scene:
    name: str
    W: int
    H: int
    means: np.ndarray          # [N,3]
    scales: np.ndarray         # [N,3]  (already exp'd, as the operators receive them)
    quats: np.ndarray          # [N,4]  (w,x,y,z)
    opacities: np.ndarray      # [N,1]  (already sigmoid'ed)
    viewmat: np.ndarray        # [4,4]
    projmat: np.ndarray        # [4,4]  full projection handed to ProjectGaussians
    fx: float
    fy: float
    cx: float
    cy: float
    background: np.ndarray     # [3]
    colors: np.ndarray | None = None    # [N,3]  when there is no SH node (C1)
    sh_coeffs: np.ndarray | None = None  # [N,K,3]
    dirs: np.ndarray | None = None       # [N,3] unit view directions
    degrees_to_use: int = 0
    v_out: np.ndarray | None = None      # [H,W,3] cotangent
    extra: dict = field(default_factory=dict)

    @property
    def N(self) -> int:
        return int(self.means.shape[0])

    @property
    def K(self) -> int:
        return 0 if self.sh_coeffs is None else int(self.sh_coeffs.shape[1])


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> np.ndarray:
    """OpenGL-style perspective matrix as OpenSplat builds it (model.cpp:35-47)."""
    t = znear * math.tan(0.5 * fovy)
    b = -t
    r = znear * math.tan(0.5 * fovx)
    l = -r
    return np.array([[2.0 * znear / (r - l), 0.0, (r + l) / (r - l), 0.0],
                     [0.0, 2.0 * znear / (t - b), (t + b) / (t - b), 0.0],
                     [0.0, 0.0, (zfar + znear) / (zfar - znear), -1.0 * zfar * znear / (zfar - znear)],
                     [0.0, 0.0, 1.0, 0.0]], dtype=np.float32)


def random_quats(u, v, w) -> np.ndarray:
    """simple_trainer.cpp:121-126 / model.cpp:24-32."""
    return np.stack([np.sqrt(1.0 - u) * np.sin(2.0 * np.pi * v), np.sqrt(1.0 - u) * np.cos(2.0 * np.pi * v),
                     np.sqrt(u) * np.sin(2.0 * np.pi * w), np.sqrt(u) * np.cos(2.0 * np.pi * w)],
                    axis=-1).astype(np.float32)


def simple_trainer_scene(N: int = 10_000, W: int = 256, H: int = 256, seed: int = 0) -> Scene:
    """BASELINE config 1: exactly simple_trainer.cpp's tensors (torch CPU RNG, same draw order)."""
    import torch

    torch.manual_seed(seed)
    means = 2.0 * (torch.rand(N, 3) - 0.5)
    scales = torch.rand(N, 3)
    rgbs = torch.rand(N, 3)
    u, v, w = torch.rand(N, 1), torch.rand(N, 1), torch.rand(N, 1)
    PI = math.pi
    quats = torch.cat([torch.sqrt(1.0 - u) * torch.sin(2.0 * PI * v),
                       torch.sqrt(1.0 - u) * torch.cos(2.0 * PI * v),
                       torch.sqrt(u) * torch.sin(2.0 * PI * w),
                       torch.sqrt(u) * torch.cos(2.0 * PI * w)], -1)
    opacities = torch.ones(N, 1)
    viewmat = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 8], [0, 0, 0, 1]], dtype=np.float32)
    focal = 0.5 * W / math.tan(0.5 * (PI / 2.0))
    gt = np.ones((H, W, 3), dtype=np.float32)
    gt[: H // 2, : W // 2, :] = np.array([1.0, 0.0, 0.0], dtype=np.float32)
    gt[H // 2:, W // 2:, :] = np.array([0.0, 0.0, 1.0], dtype=np.float32)
    return Scene(name="C1_simple_trainer", W=W, H=H, means=means.numpy(), scales=scales.numpy(),
                 quats=quats.numpy(), opacities=torch.sigmoid(opacities).numpy(),
                 viewmat=viewmat, projmat=viewmat.copy(), fx=float(focal), fy=float(focal),
                 cx=float(W // 2), cy=float(H // 2), background=np.zeros(3, dtype=np.float32),
                 colors=torch.sigmoid(rgbs).numpy(), extra=dict(gt_image=gt, raw_rgbs=rgbs.numpy()))


def yaw_camera(W: int, H: int, yaw_deg: float, znear: float = 0.001, zfar: float = 1000.0):
    """(viewmat, projmat) of camera_scene's camera — at the origin, fovX = 90 deg, rotated about +y
    by yaw_deg (C4: one such camera per rank, C4_YAWS)."""
    fx = fy = 0.5 * W
    fovx = 2.0 * math.atan(W / (2.0 * fx))
    fovy = 2.0 * math.atan(H / (2.0 * fy))
    viewmat = np.eye(4, dtype=np.float32)
    if yaw_deg != 0.0:
        a = math.radians(yaw_deg)
        R = np.array([[math.cos(a), 0, -math.sin(a)], [0, 1, 0], [math.sin(a), 0, math.cos(a)]],
                     dtype=np.float32)
        viewmat[:3, :3] = R
    projmat = projection_matrix(znear, zfar, fovx, fovy) @ viewmat
    return viewmat, projmat.astype(np.float32)


C4_YAWS = [-14.0, -10.0, -6.0, -2.0, 2.0, 6.0, 10.0, 14.0]


def camera_scene(N: int, W: int, H: int, K: int = 16, seed: int = 1, sigma_px=(0.5, 4.0),
                 z_range=(2.0, 10.0), znear: float = 0.001, zfar: float = 1000.0,
                 degrees_to_use: int | None = None, yaw_deg: float = 0.0, name: str | None = None,
                 with_cotangent: bool = True, hot=(0.0, 0)) -> Scene:
    """C2/C3-style scene: fovX = 90 deg, Gaussians spread over the image footprint
    at depth z, pixel-space sigma log-uniform in `sigma_px` with per-axis anisotropy U(0.3, 1).

    Depths lie on a jittered, shuffled grid so that no two Gaussians share a depth (the reference's
    CPU sort is unstable, gsplat_cpu.cpp:155-159): spacing (z1-z0)/N.

    hot = (fraction, box_px): that fraction of the Gaussians is concentrated in a box_px x box_px
    window at the image centre — a few tiles with very long lists, as real captures have.
    """
    rng = np.random.RandomState(seed)
    fx = fy = 0.5 * W
    cx, cy = W / 2.0, H / 2.0
    fovx = 2.0 * math.atan(W / (2.0 * fx))
    fovy = 2.0 * math.atan(H / (2.0 * fy))
    z0, z1 = z_range
    slot = rng.permutation(N).astype(np.float64)
    z = z0 + (z1 - z0) * (slot + 0.25 + 0.5 * rng.rand(N)) / N
    margin = 0.02
    px = (margin + (1 - 2 * margin) * rng.rand(N)) * W
    py = (margin + (1 - 2 * margin) * rng.rand(N)) * H
    if hot[0] > 0:
        nh = int(N * hot[0])
        px[:nh] = cx + (rng.rand(nh) - 0.5) * hot[1]
        py[:nh] = cy + (rng.rand(nh) - 0.5) * hot[1]
    x = (px - cx) * z / fx
    y = (py - cy) * z / fy
    means = np.stack([x, y, z], -1).astype(np.float32)
    s_px = np.exp(rng.uniform(math.log(sigma_px[0]), math.log(sigma_px[1]), size=N))
    aniso = rng.uniform(0.3, 1.0, size=(N, 3))
    scales = (s_px[:, None] * aniso * z[:, None] / fx).astype(np.float32)
    quats = random_quats(rng.rand(N), rng.rand(N), rng.rand(N))
    opac = 1.0 / (1.0 + np.exp(-rng.normal(0.0, 2.0, size=(N, 1))))
    viewmat, projmat = yaw_camera(W, H, yaw_deg, znear, zfar)
    sh = None
    dirs = None
    colors = None
    if K > 0:
        sh = np.empty((N, K, 3), dtype=np.float32)
        sh[:, 0, :] = rng.uniform(-1.5, 1.5, size=(N, 3))
        if K > 1:
            sh[:, 1:, :] = rng.normal(0.0, 0.1, size=(N, K - 1, 3))
        d = means.astype(np.float64)  # camera at the origin (model.cpp:176-177)
        dirs = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    else:
        colors = rng.rand(N, 3).astype(np.float32)
    deg = {0: 0, 1: 0, 4: 1, 9: 2, 16: 3, 25: 4}[K] if degrees_to_use is None else degrees_to_use
    v_out = rng.uniform(-1.0, 1.0, size=(H, W, 3)).astype(np.float32) if with_cotangent else None
    return Scene(name=name or f"camera_N{N}_{W}x{H}_K{K}_s{seed}", W=W, H=H, means=means,
                 scales=scales, quats=quats, opacities=opac.astype(np.float32), viewmat=viewmat,
                 projmat=projmat.astype(np.float32), fx=float(fx), fy=float(fy), cx=float(cx),
                 cy=float(cy), background=np.array(BACKGROUND, dtype=np.float32), colors=colors,
                 sh_coeffs=sh, dirs=dirs, degrees_to_use=deg, v_out=v_out)


# The named BASELINE.json configurations
def config_c1() -> Scene:
    return simple_trainer_scene(10_000, 256, 256, seed=0)


def config_c2(N: int = 1_000_000) -> Scene:
    return camera_scene(N, 1920, 1080, K=16, seed=1, sigma_px=(0.5, 4.0), name="C2_1M_1080p_sh3")


def config_c3(N: int = 5_000_000) -> Scene:
    return camera_scene(N, 3840, 2160, K=16, seed=2, sigma_px=(1.0, 8.0), name="C3_5M_4k_sh3")


def config_c4(rank: int, N: int = 1_000_000) -> Scene:
    """C4: the C2 Gaussians seen by camera `rank` of 8 (yaw offsets)."""
    s = camera_scene(N, 1920, 1080, K=16, seed=3, sigma_px=(0.5, 4.0), yaw_deg=C4_YAWS[rank % 8],
                     name=f"C4_cam{rank}")
    return s


# ---------------------------------------------------------------------------------------------
# inputs of the loss / optimiser tests and measurements

def loss_images(W: int, H: int, seed: int = 0, noise: float = 0.1):
    """(rendered, gt) as float32 [H, W, 3]: a smooth random "photo" in [0, 1] (what
    Camera::getImage hands to Model::mainLoss) and a rendering of it that is off by low-frequency
    error + noise, clamped to <= 1 like Model::forward's output (model.cpp:222), with a patch of
    exactly equal pixels (sign(0) in the L1 gradient) and a saturated patch."""
    rng = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32),
                         indexing="ij")
    gt = np.zeros((H, W, 3), np.float32)
    for _ in range(6):
        fx, fy = rng.uniform(0.005, 0.12, 2)
        ph = rng.uniform(0, 2 * np.pi, 3)
        amp = rng.uniform(0.05, 0.25, 3)
        for c in range(3):
            gt[..., c] += amp[c] * np.sin(fx * xx + fy * yy + ph[c]).astype(np.float32)
    gt = np.clip(gt + 0.5, 0.0, 1.0).astype(np.float32)
    err = 0.15 * np.sin(0.03 * xx[..., None] + rng.uniform(0, 6, 3)).astype(np.float32)
    rendered = gt + err + noise * rng.standard_normal((H, W, 3)).astype(np.float32)
    rendered = np.minimum(np.maximum(rendered, 0.0), 1.0).astype(np.float32)
    h4, w4 = max(H // 4, 1), max(W // 4, 1)
    rendered[:h4, :w4] = gt[:h4, :w4]          # exactly equal
    rendered[-h4:, -w4:] = 1.0                 # saturated
    return np.ascontiguousarray(rendered), np.ascontiguousarray(gt)


def adam_problem(n: int, steps: int, seed: int = 0):
    """(param0, [grad_1 .. grad_steps]) float32: gradients spanning 1e-12 .. 1e3 in magnitude,
    with exact zeros (Gaussians outside the frustum get zero gradients every step)."""
    rng = np.random.RandomState(seed)
    p = rng.standard_normal(n).astype(np.float32)
    mag = (10.0 ** rng.uniform(-12, 3, n)).astype(np.float32)
    grads = []
    for _ in range(steps):
        g = (mag * rng.standard_normal(n)).astype(np.float32)
        g[rng.rand(n) < 0.1] = 0.0
        grads.append(g)
    return p, grads


def raw_parameters(s: Scene):
    """The optimiser's view of scene `s` (model.hpp): (means, log-scales, raw quats, opacity logits
    [N,1], featuresDc [N,3], featuresRest [N,K-1,3]) such that Model::forward's glue (exp,
    normalise, sigmoid, cat; model.cpp:114-215) reproduces the scene's tensors."""
    o = np.clip(s.opacities.reshape(-1, 1), 1e-6, 1 - 1e-6)
    logits = np.log(o / (1 - o)).astype(np.float32)
    return (s.means, np.log(s.scales).astype(np.float32), s.quats.astype(np.float32), logits,
            np.ascontiguousarray(s.sh_coeffs[:, 0, :]), np.ascontiguousarray(s.sh_coeffs[:, 1:, :]))


def densify_problem(N: int, K: int = 4, seed: int = 0, width: int = 640, height: int = 480):
    """Inputs of one refinement step: the six raw parameter tensors, Adam
    moments, the accumulated statistics, with every branch of Model::afterTrain populated: high /
    low gradients, small / large / huge Gaussians, large screen footprints (so that some Gaussians
    are BOTH split and duplicated, as the reference allows), faint opacities.  Quantities compared
    with thresholds are kept >= 1e-3 (relative) away from them, so that exp / sigmoid rounding
    cannot flip a decision."""
    rs = np.random.RandomState(seed)
    f = np.float32

    def away(x, thr):  # push values out of the +-0.1 % band around a threshold
        x = x.copy()
        near = np.abs(x / thr - 1.0) < 1e-3
        x[near] = thr * 1.01
        return x

    size = away(10.0 ** rs.uniform(-3.2, 0.2, N), 0.01)              # densifySizeThresh, cullScaleThresh
    size = away(size, 0.5)
    size = away(size, 0.5 * 1.6)                                     # split samples vs cullScaleThresh
    aniso = rs.uniform(0.2, 1.0, (N, 3))
    aniso[np.arange(N), rs.randint(0, 3, N)] = 1.0
    log_scales = np.log(size[:, None] * aniso).astype(f)
    means = rs.uniform(-1, 1, (N, 3)).astype(f)
    quats = (rs.standard_normal((N, 4)) * rs.uniform(0.3, 3.0, (N, 1))).astype(f)
    alpha = away(rs.uniform(0.005, 0.995, N), 0.1)
    logits = np.log(alpha / (1 - alpha)).astype(f).reshape(N, 1)
    dc = rs.uniform(-1.5, 1.5, (N, 3)).astype(f)
    rest = (0.1 * rs.standard_normal((N, K - 1, 3))).astype(f)
    params = [means, log_scales, quats, logits, dc, rest]
    exp_avg = [(1e-3 * rs.standard_normal(p.shape)).astype(f) for p in params]
    exp_avg_sq = [(1e-6 * rs.uniform(0, 1, p.shape)).astype(f) for p in params]
    vis = rs.randint(1, 100, N).astype(f)
    side = float(max(width, height))
    avg = away(10.0 ** rs.uniform(-5.5, -2.0, N), 0.0002)            # densifyGradThresh
    gnorm = (avg * vis / (0.5 * side)).astype(f)
    m2d = away(rs.uniform(0.0, 0.3, N) * (rs.rand(N) < 0.5), 0.05)   # splitScreenSize, cullScreenSize
    m2d = away(m2d, 0.15).astype(f)
    return dict(params=params, exp_avg=exp_avg, exp_avg_sq=exp_avg_sq, xys_grad_norm=gnorm,
                vis_counts=vis, max_2d_size=m2d, width=width, height=height, K=K, N=N)
