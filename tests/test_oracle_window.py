"""CPU: the pixel-window variant of the restated oracle (used by the BASELINE-size GPU parity tests)
is the full-frame evaluation restricted to the window, bit for bit."""
import numpy as np

from opensplat_amd import scenes


def test_window_equals_full_frame_restricted(restated):
    s = scenes.camera_scene(3000, 160, 112, K=4, seed=17, sigma_px=(0.8, 6.0), znear=1.0, zfar=100.0)
    O = restated
    o = O.project_forward(s.means, s.scales, s.quats, s.viewmat, s.projmat, s.fx, s.fy, s.cx, s.cy,
                          s.H, s.W)
    c = np.maximum(O.sh_forward(s.degrees_to_use, s.dirs, s.sh_coeffs) + 0.5, 0).astype(np.float32)
    args = (s.W, s.H, o["xys"], o["conics"], c, s.opacities, s.background, o["cov2d"], o["depths"])
    full = O.rasterize_forward(*args, want_contributors=False)
    gfull_state = full["state"]
    win = (37, 20, 121, 77)
    x0, y0, x1, y1 = win
    part = O.rasterize_forward(*args, want_contributors=False, window=win)
    assert np.array_equal(part["img"][y0:y1, x0:x1], full["img"][y0:y1, x0:x1])
    assert np.array_equal(part["final_Ts"][y0:y1, x0:x1], full["final_Ts"][y0:y1, x0:x1])
    outside = np.ones((s.H, s.W), bool)
    outside[y0:y1, x0:x1] = False
    assert (part["final_Ts"][outside] == 1.0).all() and (part["px_counts"][outside] == 0).all()
    assert np.array_equal(part["px_counts"][y0:y1, x0:x1], full["px_counts"][y0:y1, x0:x1])
    # backward: the windowed gradient equals the full-frame gradient of a cotangent zeroed outside
    v = s.v_out.copy()
    v[outside] = 0.0
    gw = O.rasterize_backward(*args, part["final_Ts"], part["state"], s.v_out, window=win)
    gf = O.rasterize_backward(*args, full["final_Ts"], gfull_state, v)
    for k in gw:
        assert np.array_equal(gw[k], gf[k]), k
