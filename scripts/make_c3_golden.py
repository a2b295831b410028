#!/usr/bin/env python3
"""Whole-frame golden record of BASELINE config 3 (5 M Gaussians, 3840x2160, SH degree 3) — VERDICT r03 "next" 5.

Runs ONCE, in the build container (minutes of CPU, ~10 GB of RAM), the REFERENCE's own compositing
(`oracle/_ref`: rasterizer/gsplat-cpu/gsplat_cpu.cpp:137-376 compiled in place, forward and backward) on the
whole C3 frame and stores, in tests/golden/c3_whole_frame.json, what a GPU run must reproduce:

  inputs   sha256 of the 2-D inputs (xys, conics, cov2d, depths, colours, radii).  They come from the plain-C
           restatement of the projection / SH stages (oracle/gsplat_oracle.c — deterministic scalar code; the
           reference's own projection is a chain of libtorch CPU ops whose bits may depend on the host's vector
           ISA), so that the GPU box can rebuild them bit for bit on its host and feed the SAME numbers to the
           HIP compositing kernels;
  forward  sha256 of the image and of final_Ts (all 8 294 400 pixels), sha256 of the per-pixel contributor
           counts, the total number of (pixel, Gaussian) contributions;
  backward for each of the four 2-D gradient tensors the compositing backward returns and each of the six
           parameter gradients behind them (oracle projection / SH backward fed the reference's 2-D gradients):
           max |g|, the fp64 L2 norm, and fp64 dot products with three seeded standard-normal vectors.

tests/test_gpu_c3_golden.py asserts the hashes bit for bit and the projections within 2e-5 of the norm.

    python scripts/make_c3_golden.py            # needs oracle/_ref (python -c "import __graft_entry__ as g; g.build()")
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from opensplat_amd import scenes  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "c3_whole_frame.json")
V_OUT_SEED = 3
DOT_SEEDS = (101, 202, 303)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def grad_record(g):
    """max |g|, L2 norm and three seeded projections of a gradient tensor, in fp64."""
    g64 = np.asarray(g, dtype=np.float64).ravel()
    rec = {"max_abs": float(np.abs(g64).max()), "l2": float(np.sqrt((g64 * g64).sum())), "dots": []}
    for seed in DOT_SEEDS:
        r = np.random.RandomState(seed).standard_normal(g64.size)
        rec["dots"].append(float(g64 @ r))
    return rec


def inputs_2d(O, s):
    """The frame's 2-D inputs from the deterministic C restatement (shared with the GPU test)."""
    o = O.project_forward(s.means, s.scales, s.quats, s.viewmat, s.projmat, s.fx, s.fy, s.cx, s.cy, s.H, s.W)
    shc = O.sh_forward(s.degrees_to_use, s.dirs, s.sh_coeffs)
    colors = np.maximum(shc + np.float32(0.5), 0.0).astype(np.float32)   # model.cpp:192
    return o, shc, colors


def cotangent(s):
    return np.random.RandomState(V_OUT_SEED).uniform(-1, 1, (s.H, s.W, 3)).astype(np.float32)


def main():
    if not oracle.have_reference():
        sys.exit("oracle/_ref is not built (needs /root/reference): python -c 'import __graft_entry__ as g; g.build()'")
    O, R = oracle.restated(), oracle.reference()
    t0 = time.time()
    s = scenes.config_c3()
    o, shc, colors = inputs_2d(O, s)
    v_out = cotangent(s)
    print("scene + 2-D inputs: %.1f s" % (time.time() - t0), flush=True)
    rec = {"config": "C3: %d Gaussians, %dx%d, K=%d, seed 2 (scenes.config_c3)" % (s.N, s.W, s.H, s.K),
           "made_by": "scripts/make_c3_golden.py: compositing by oracle/_ref (gsplat_cpu.cpp compiled in place), "
                      "2-D inputs and per-Gaussian backward stages by oracle/gsplat_oracle.c",
           "v_out_seed": V_OUT_SEED, "dot_seeds": list(DOT_SEEDS),
           "inputs_sha256": {k: sha(o[k]) for k in ("xys", "conics", "cov2d", "depths", "radii")}}
    rec["inputs_sha256"]["colors"] = sha(colors)
    t0 = time.time()
    f = R.rasterize_forward(s.W, s.H, o["xys"], o["conics"], colors, s.opacities, s.background, o["cov2d"],
                            o["depths"], want_contributors=False)
    rec["forward_s"] = time.time() - t0
    print("reference forward: %.1f s" % rec["forward_s"], flush=True)
    rec["forward"] = {"img_sha256": sha(f["img"]), "final_Ts_sha256": sha(f["final_Ts"]),
                      "px_counts_sha256": sha(f["px_counts"].astype(np.int32)),
                      "contributions": int(f["px_counts"].astype(np.int64).sum()),
                      "img_sum_fp64": float(f["img"].astype(np.float64).sum()),
                      "saturated_pixels": int((f["final_Ts"] <= 1e-4 * 1.0001).sum())}
    t0 = time.time()
    g = R.rasterize_backward(s.W, s.H, o["xys"], o["conics"], colors, s.opacities, s.background, o["cov2d"],
                             o["depths"], f["final_Ts"], f["state"], v_out)
    rec["backward_s"] = time.time() - t0
    print("reference backward: %.1f s" % rec["backward_s"], flush=True)
    v_rgb = (g["v_colors"] * (shc + np.float32(0.5) > 0)).astype(np.float32)
    v_coeffs = O.sh_backward(s.degrees_to_use, s.dirs, s.sh_coeffs, v_rgb)
    pb = O.project_backward(s.means, s.scales, s.quats, s.viewmat, s.projmat, s.fx, s.fy, s.cx, s.cy, s.H, s.W,
                            g["v_xy"], g["v_conic"])
    rec["backward"] = {k: grad_record(g[k]) for k in ("v_xy", "v_conic", "v_colors", "v_opacity")}
    rec["backward"].update({"v_means": grad_record(pb["v_means"]), "v_scales": grad_record(pb["v_scales"]),
                            "v_quats": grad_record(pb["v_quats"]), "v_coeffs": grad_record(v_coeffs)})
    with open(OUT, "w") as fh:
        json.dump(rec, fh, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
