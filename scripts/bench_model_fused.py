"""Iterations/s of OpenSplat's own `Model` on the fused operators (oracle/_ref/model_fused_shim: the
reference's model.cpp patched by `integration/apply_hip_native.py --fused`, training loop at the
opensplat.cpp:151-170 level in C++/libtorch) next to the Python `Trainer` on the same scene.

    python scripts/bench_model_fused.py [--n 6000] [--iters 400] [--out profiles/model_fused_r03.json]

Needs the GPU box and the shim (built by __graft_entry__.build() where /root/reference exists).
"""
import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=6000)
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--width", type=int, default=384)
    ap.add_argument("--height", type=int, default=288)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()

    import torch

    import tests.test_gpu_model_fused as t
    from opensplat_amd import scenes
    from tests.util import to_dev

    W, H, K = a.width, a.height, 16
    s = scenes.camera_scene(a.n, W, H, K=K, seed=4, znear=1.0, zfar=100.0)
    # camera_scene is laid out for the camera at the origin looking down +z: what cam_to_world(0) gives
    params = [s.means, np.log(s.scales), s.quats, np.log(s.opacities / (1 - s.opacities)).reshape(-1, 1),
              s.sh_coeffs[:, 0, :].copy(), s.sh_coeffs[:, 1:, :].copy()]
    params = [np.ascontiguousarray(p, np.float32) for p in params]
    yaws = [0.0, 1.0, -1.0, 2.0]
    # no refinement inside the timed loop (warm-up beyond it): the steady-state iteration
    case, c, gts = t.make_case(params, s.fx, s.fy, W, H, yaws, a.iters, shDegree=3, shDegreeInterval=1,
                               warmupLength=10 ** 6)
    tmp = Path(tempfile.mkdtemp())
    out = {"scene": {"gaussians": a.n, "width": W, "height": H, "K": K, "cameras": len(yaws)}, "iterations": a.iters}
    # a first run pays the page-in of the libraries; time the second
    t.run_shim(tmp, dict(case, cfg=np.concatenate([case["cfg"][:10], [8], case["cfg"][11:]]).astype(np.float32)),
               "gpu", "warm")
    r, _ = t.run_shim(tmp, case, "gpu", "bench")
    out["model_fused_cpp"] = {"seconds": float(r["seconds"][0]), "iterations_per_s": a.iters / float(r["seconds"][0]),
                              "final_loss": float(r["losses"][-1])}
    t0 = time.perf_counter()
    py = t.run_trainer(params, s.fx, s.fy, W, H, yaws, 8, gts, c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    py = t.run_trainer(params, s.fx, s.fy, W, H, yaws, a.iters, gts, c)
    torch.cuda.synchronize()
    sec = time.perf_counter() - t0
    out["python_trainer"] = {"seconds": sec, "iterations_per_s": a.iters / sec, "final_loss": float(py["losses"][-1]),
                             "note": "includes Trainer construction and one float(loss) host sync per iteration"}
    print(json.dumps(out))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
