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


def test_tile_contract_restricts_each_gaussian_to_its_tiles(restated):
    """orc_rasterize_forward_tiles (the launcher-level contract, forward.cu:86-94 + :256-283): with every tile
    allowed it IS the plain oracle; with Gaussians confined to a tile range, pixels outside it never see them
    and pixels inside composite exactly the confined set."""
    from opensplat_amd import scenes

    O = restated
    s = scenes.camera_scene(2500, 160, 96, K=0, seed=5, znear=1.0, zfar=100.0)
    o = O.project_forward(s.means, s.scales, s.quats, s.viewmat, s.projmat, s.fx, s.fy, s.cx, s.cy, s.H, s.W)
    args = (s.W, s.H, o["xys"], o["conics"], s.colors, s.opacities, s.background, o["cov2d"], o["depths"])
    full = O.rasterize_forward(*args)
    tr = np.tile(np.array([0, 10, 0, 6], np.int32), (s.N, 1))
    a = O.rasterize_forward(*args, tile_rect=tr)
    assert np.array_equal(a["img"], full["img"]) and np.array_equal(a["contributors"], full["contributors"])
    # odd Gaussians confined to the left half, even ones to the top half
    tr[1::2, 1] = 5
    tr[0::2, 3] = 3
    b = O.rasterize_forward(*args, tile_rect=tr)
    # reference: render the two sub-populations' allowed regions by zeroing opacities
    op_odd = s.opacities.copy(); op_odd[0::2] = 0
    op_even = s.opacities.copy(); op_even[1::2] = 0
    only_odd = O.rasterize_forward(s.W, s.H, o["xys"], o["conics"], s.colors, op_odd, s.background, o["cov2d"], o["depths"])
    only_even = O.rasterize_forward(s.W, s.H, o["xys"], o["conics"], s.colors, op_even, s.background, o["cov2d"], o["depths"])
    assert np.array_equal(b["img"][:48, :80], full["img"][:48, :80])          # both populations allowed
    assert np.array_equal(b["img"][48:, :80], only_odd["img"][48:, :80])      # bottom left: odd only
    assert np.array_equal(b["img"][:48, 80:], only_even["img"][:48, 80:])     # top right: even only
    assert np.allclose(b["img"][48:, 80:], s.background)                      # bottom right: nobody
    for st in (full, a, b, only_odd, only_even):
        O.rasterize_free(st["state"])
