#!/usr/bin/env python3
"""OpenSplat's Model::forward + backward (model.cpp:83-225) at config C2,
  (a) as the reference runs it — torch element-wise glue (exp, normalise, cat, view dirs, +0.5,
      clamp_min, sigmoid, clamp_max) around the three operators (this repo's implementations), and
  (b) through the fused SplatRender operator,
both through libtorch autograd on one MI355X.  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from opensplat_amd import ops, scenes  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
s = scenes.config_c2()
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
o = np.clip(s.opacities.reshape(-1, 1), 1e-6, 1 - 1e-6)
R, tr = s.viewmat[:3, :3], s.viewmat[:3, 3]
params = [t(s.means), t(np.log(s.scales).astype(np.float32)), t(s.quats),
          t(np.log(o / (1 - o)).astype(np.float32)), t(s.sh_coeffs[:, 0, :]), t(s.sh_coeffs[:, 1:, :])]
for p in params:
    p.requires_grad_(True)
cam_pos = t((-R.T @ tr).astype(np.float32))
vm, pm, bg, v_img = t(s.viewmat), t(s.projmat), t(s.background), t(s.v_out)
xys_grad = torch.zeros((s.N, 2), device=dev)


def unfused():
    means, ls, q, lo, dc, rest = params
    p = ops.project_gaussians(means, torch.exp(ls), 1.0, q / q.norm(2, -1, True), vm, pm, s.fx, s.fy,
                              s.cx, s.cy, s.H, s.W)
    xys = p[0]
    xys.retain_grad()
    colors = torch.cat([dc[:, None, :], rest], 1)
    dirs = means.detach() - cam_pos
    dirs = dirs / dirs.norm(2, -1, True)
    rgbs = torch.clamp_min(ops.spherical_harmonics(s.degrees_to_use, dirs, colors) + 0.5, 0.0)
    img = ops.rasterize_gaussians(xys, p[1], p[2], p[3], p[4], rgbs, torch.sigmoid(lo), s.H, s.W, bg,
                                  p[6])
    torch.clamp_max(img, 1.0).backward(v_img)


def fused():
    out = ops.splat_render(*params, vm, pm, cam_pos, s.fx, s.fy, s.cx, s.cy, s.H, s.W,
                           s.degrees_to_use, bg, xys_grad)
    out[0].backward(v_img)


def timeit(fn):
    for _ in range(3):
        for p in params:
            p.grad = None
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for p in params:
            p.grad = None
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


a, b = timeit(unfused), timeit(fused)
print(json.dumps({"workload": "C2 Model::forward+backward through libtorch autograd, %d steps" % steps,
                  "unfused_ms": a, "fused_ms": b, "speedup": a / b,
                  "unfused_per_s": 1e3 / a, "fused_per_s": 1e3 / b}))
