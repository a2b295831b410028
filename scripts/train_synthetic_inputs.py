#!/usr/bin/env python3
"""Inputs of the synthetic end-to-end run (scripts/train_synthetic.py), pure numpy: the camera ring,
the ground-truth Gaussian set and the SfM-like initial point set.  Also imported by
`bench.py --train-cpu-baselines`, which times the reference's CPU code on the same initial set."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from opensplat_amd import scenes  # noqa: E402

C0 = 0.28209479177387814


def look_at(pos, target=(0.0, 0.0, 0.0)):
    """World -> camera (x right, y down, z forward: the convention of scenes.py / model.cpp:93-104)."""
    pos, target = np.asarray(pos, np.float64), np.asarray(target, np.float64)
    f = target - pos
    f /= np.linalg.norm(f)
    r = np.cross(f, np.array([0.0, -1.0, 0.0]))
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    R = np.stack([r, d, f])
    vm = np.eye(4, dtype=np.float32)
    vm[:3, :3] = R
    vm[:3, 3] = -R @ pos
    return vm


def make_camera(pos, W, H, fov_deg=50.0):
    fx = fy = 0.5 * W / math.tan(0.5 * math.radians(fov_deg))
    fovx, fovy = 2.0 * math.atan(W / (2.0 * fx)), 2.0 * math.atan(H / (2.0 * fy))
    vm = look_at(pos)
    pm = (scenes.projection_matrix(0.001, 1000.0, fovx, fovy) @ vm).astype(np.float32)
    return dict(viewmat=vm, projmat=pm, fx=fx, fy=fy, cx=W / 2.0, cy=H / 2.0, W=W, H=H)


def ground_truth(n, K, rs):
    """A few blobs and a shell of small anisotropic Gaussians with position-dependent colour."""
    centres = rs.uniform(-0.6, 0.6, (6, 3))
    which = rs.randint(0, 6, n)
    means = centres[which] + 0.22 * rs.standard_normal((n, 3))
    shell = rs.rand(n) < 0.3
    d = rs.standard_normal((n, 3))
    means[shell] = 0.95 * d[shell] / np.linalg.norm(d[shell], axis=1, keepdims=True)
    log_scales = np.log(rs.uniform(0.012, 0.05, (n, 1)) * rs.uniform(0.3, 1.0, (n, 3)))
    quats = scenes.random_quats(rs.rand(n), rs.rand(n), rs.rand(n))
    logits = rs.normal(1.5, 1.0, (n, 1))
    rgb = 0.5 + 0.45 * np.sin(3.0 * means + np.array([0.0, 2.0, 4.0]))
    dc = (rgb - 0.5) / C0
    rest = 0.03 * rs.standard_normal((n, K - 1, 3))
    f = np.float32
    return [means.astype(f), log_scales.astype(f), quats.astype(f), logits.astype(f), dc.astype(f),
            rest.astype(f)]


def sfm_like_init(gt, n, K, rs):
    """Model's initialisation from points (model.hpp:36-60): means = points, scales = log of the mean
    distance to the 3 nearest neighbours, random quats, opacity logit(0.1), featuresDc = rgb2sh."""
    from scipy.spatial import cKDTree

    idx = rs.choice(gt[0].shape[0], n, replace=False)
    pts = gt[0][idx] + 0.01 * rs.standard_normal((n, 3)).astype(np.float32)
    rgb = np.clip(gt[4][idx] * C0 + 0.5 + 0.05 * rs.standard_normal((n, 3)), 0.0, 1.0)
    dist, _ = cKDTree(pts).query(pts, k=4)
    scale = np.log(np.maximum(dist[:, 1:].mean(1), 1e-4))[:, None].repeat(3, 1)
    f = np.float32
    return [pts.astype(f), scale.astype(f),
            scenes.random_quats(rs.rand(n), rs.rand(n), rs.rand(n)),
            np.full((n, 1), math.log(0.1 / 0.9), f), ((rgb - 0.5) / C0).astype(f),
            np.zeros((n, K - 1, 3), f)]


