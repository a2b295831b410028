"""-m gpu: ALL 8 294 400 pixels of BASELINE config 3 against the reference (VERDICT r03 "next" 5).

tests/golden/c3_whole_frame.json was made once by scripts/make_c3_golden.py: the REFERENCE's own compositing
(oracle/_ref = gsplat_cpu.cpp:137-376 compiled in place) on the whole 5 M-Gaussian 3840x2160 frame — sha256 of
the image / final_Ts, and for ten gradient tensors max |g|, the L2 norm and three seeded fp64 projections.
Here the GPU box rebuilds the frame's 2-D inputs on its host with the same deterministic C restatement
(checked against the recorded input hashes), feeds them to the HIP binning + compositing kernels through the
C ABI and must reproduce the forward BIT FOR BIT and every projection within 2e-5 of the tensor's norm.
(The three 256x160 windows of test_gpu_baseline_parity.py stay as the element-wise check; this one pins the
other 99.5 % of the frame.)"""
import hashlib
import json
import os

import numpy as np
import pytest

from opensplat_amd import scenes
from tests.util import np_, to_dev

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c3_whole_frame.json")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.skipif(not os.path.exists(GOLD), reason="tests/golden/c3_whole_frame.json not made")
def test_c3_whole_frame_equals_the_reference(restated):
    import torch

    from opensplat_amd import cabi

    gold = json.load(open(GOLD))
    s = scenes.config_c3()
    O = restated
    o = O.project_forward(s.means, s.scales, s.quats, s.viewmat, s.projmat, s.fx, s.fy, s.cx, s.cy, s.H, s.W)
    shc = O.sh_forward(s.degrees_to_use, s.dirs, s.sh_coeffs)
    colors = np.maximum(shc + np.float32(0.5), 0.0).astype(np.float32)
    mine = {k: sha(o[k]) for k in ("xys", "conics", "cov2d", "depths", "radii")}
    mine["colors"] = sha(colors)
    if mine != gold["inputs_sha256"]:
        pytest.skip("this host's libm / CPU rebuilds other 2-D inputs than the golden run's: %s" %
                    [k for k in mine if mine[k] != gold["inputs_sha256"][k]])
    v_out = np.random.RandomState(gold["v_out_seed"]).uniform(-1, 1, (s.H, s.W, 3)).astype(np.float32)

    N = s.N
    cov2d3 = np.ascontiguousarray(o["cov2d"].reshape(N, 4)[:, [0, 1, 3]])
    xys, conics, col = to_dev(o["xys"]), to_dev(o["conics"]), to_dev(colors)
    opac, c2, depths = to_dev(s.opacities.reshape(-1)), to_dev(cov2d3), to_dev(o["depths"])
    radii = to_dev(o["radii"].astype(np.int32))
    b = cabi.bin_and_sort(s.W, s.H, xys, depths, radii, conics, col, opac, c2)
    f = cabi.rasterize_forward(s.W, s.H, b, s.background)
    torch.cuda.synchronize()
    img, fT = np_(f["img"]), np_(f["final_Ts"])
    assert sha(img) == gold["forward"]["img_sha256"], "C3 image differs from the reference somewhere in the frame"
    assert sha(fT) == gold["forward"]["final_Ts_sha256"]
    assert float(img.astype(np.float64).sum()) == gold["forward"]["img_sum_fp64"]

    g = cabi.rasterize_backward(s.W, s.H, N, b, s.background, f["final_Ts"], f["final_idx"], to_dev(v_out))
    cam = cabi.make_camera(s.viewmat, s.projmat, s.fx, s.fy, s.cx, s.cy, s.W, s.H)
    means, scales, quats = to_dev(s.means), to_dev(s.scales), to_dev(s.quats)
    v_rgb = (g["v_colors"] * (to_dev(shc) + 0.5 > 0).float()).contiguous()
    v_coeffs = cabi.sh_backward(s.degrees_to_use, s.K, to_dev(s.dirs), v_rgb)
    pb = cabi.project_backward(cam, means, scales, quats, radii, g["v_xy"], g["v_conic"])
    torch.cuda.synchronize()
    got = {"v_xy": g["v_xy"], "v_conic": g["v_conic"], "v_colors": g["v_colors"], "v_opacity": g["v_opacity"],
           "v_means": pb["v_means"], "v_scales": pb["v_scales"], "v_quats": pb["v_quats"], "v_coeffs": v_coeffs}
    report = {}
    for name, t in got.items():
        ref = gold["backward"][name]
        a = np_(t).astype(np.float64).ravel()
        worst = abs(np.abs(a).max() - ref["max_abs"]) / ref["max_abs"]
        for seed, want in zip(gold["dot_seeds"], ref["dots"]):
            r = np.random.RandomState(seed).standard_normal(a.size)
            worst = max(worst, abs(float(a @ r) - want) / ref["l2"])
        report[name] = worst
        assert worst < 2e-5, (name, worst)
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        json.dump({"forward": "bit-exact (sha256 of image and final_Ts, 8 294 400 pixels)",
                   "worst_relative_projection_error": report}, open(os.path.join(d, "parity_c3_whole_r04.json"), "w"),
                  indent=1)
    except OSError:
        pass
