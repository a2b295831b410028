"""-m gpu: GS_FLAG_DETERMINISTIC (SURVEY.md §5 "race detection"): the compositing backward with
order-independent sums (per-wave partials added as 64-bit fixed point with integer atomics).  Two
runs must be BIT-IDENTICAL — so any run-to-run difference under this flag is a race, not summation
order — and agree with the default fp32-atomic path and with the oracle within the summation-order
tolerance."""
import numpy as np
import pytest

from opensplat_amd import scenes
from tests.util import hip_pipeline, np_, oracle_raster, rel_err, to_dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene", ["camera", "deep", "c1"])
def test_deterministic_backward_is_bit_reproducible_and_close(scene, restated):
    import torch

    from opensplat_amd import cabi

    if scene == "camera":
        s = scenes.camera_scene(20000, 400, 240, K=16, seed=7, znear=1.0, zfar=100.0)
    elif scene == "deep":      # long lists, heavy atomic contention: few big tiles' worth of pixels
        s = scenes.camera_scene(30000, 96, 64, K=4, seed=8, sigma_px=(1.0, 8.0), znear=1.0, zfar=100.0)
    else:
        s = scenes.simple_trainer_scene(3000, 128, 96, seed=0)
        s.v_out = np.random.RandomState(3).uniform(-1, 1, (s.H, s.W, 3)).astype(np.float32)
    out = hip_pipeline(s, backward=True)
    b, v_out = out["binned"], to_dev(s.v_out)
    runs = []
    for _ in range(3):
        g = cabi.rasterize_backward(s.W, s.H, s.N, b, s.background, out["final_Ts"], out["final_idx"],
                                    v_out, cabi.GS_FLAG_DETERMINISTIC)
        torch.cuda.synchronize()
        runs.append({k: np_(v).copy() for k, v in g.items()})
    for k in runs[0]:
        assert np.array_equal(runs[0][k], runs[1][k]) and np.array_equal(runs[0][k], runs[2][k]), k
        assert rel_err(runs[0][k], np_(out[k])) < 2e-5, k          # vs the fp32-atomic path
    f, g = oracle_raster(restated, s, np_(out["xys"]), np_(out["conics"]), np_(out["colors"]),
                         np_(out["cov2d"]), np_(out["depths"]), s.v_out)
    for k in ("v_xy", "v_conic", "v_colors", "v_opacity"):
        assert rel_err(runs[0][k], g[k].reshape(runs[0][k].shape)) < 2e-5, k


def test_deterministic_mode_needs_its_larger_workspace():
    import torch

    from opensplat_amd import cabi

    l = cabi.lib()
    assert l.gs_rasterize_backward_workspace_bytes_det(1000) == 1000 * 16 * (4 + 8)
    s = scenes.camera_scene(500, 64, 48, K=1, seed=3, znear=1.0, zfar=100.0)
    out = hip_pipeline(s, backward=False)
    small = torch.zeros(l.gs_rasterize_backward_workspace_bytes(s.N), device="cuda", dtype=torch.uint8)
    v = to_dev(np.zeros((s.H, s.W, 3), np.float32))
    with pytest.raises(cabi.GsError):
        # (cabi.rasterize_backward would grow the workspace: call with the too-small one directly)
        import ctypes as C
        b = out["binned"]
        o = {k: torch.empty(n, device="cuda") for k, n in (("a", (s.N, 2)), ("b", (s.N, 3)), ("c", (s.N, 3)), ("d", (s.N,)))}
        cabi._check(l.gs_rasterize_backward(
            C.c_int(s.W), C.c_int(s.H), C.c_int(s.N), cabi._p(b.gaussian_ids_sorted), cabi._p(b.block_masks),
            cabi._p(b.tile_bins),
            cabi._p(b.packed), cabi._vec3(s.background), cabi._p(out["final_Ts"]), cabi._p(out["final_idx"]),
            cabi._p(v), None, None, cabi._p(o["a"]), cabi._p(o["b"]), cabi._p(o["c"]), cabi._p(o["d"]),
            cabi._p(small), C.c_size_t(small.numel()), None, None, C.c_uint32(cabi.GS_FLAG_DETERMINISTIC),
            cabi._stream()), "gs_rasterize_backward")


QGEOM, CLASSIC = 1 << 25, 2 << 25   # flag bits 25..26: sixteen four-lane groups for every frame / the four-group kernels


@pytest.mark.parametrize("px", [1, 2, 4, "q", "q+det"])
@pytest.mark.parametrize("scene", ["camera", "ragged", "deep"])
def test_backward_wave_geometries_match_oracle(px, scene, restated):
    """The compositing backward with 1, 2 and 4 pixels per lane (4x4 / 4x8 / 8x8 blocks per 16-lane
    group; flag bits 21..22) and with sixteen four-lane groups per wave (4x4 blocks, four pixels per lane:
    the full-frame default since round 5, forced here by flag bit 25) — every geometry against the oracle,
    incl. image sizes that are not multiples of the wave's footprint."""
    import torch

    from opensplat_amd import cabi

    if scene == "camera":
        s = scenes.camera_scene(20000, 400, 240, K=16, seed=7, znear=1.0, zfar=100.0)
    elif scene == "ragged":
        s = scenes.camera_scene(5000, 203, 117, K=4, seed=11, sigma_px=(1.0, 6.0), znear=1.0, zfar=100.0)
    else:
        s = scenes.camera_scene(30000, 96, 64, K=4, seed=8, sigma_px=(1.0, 8.0), znear=1.0, zfar=100.0)
    out = hip_pipeline(s, backward=False)
    flag = ({1: 1, 2: 2, 4: 3}[px] << 21 if isinstance(px, int) else
            QGEOM | (cabi.GS_FLAG_DETERMINISTIC if px == "q+det" else 0))
    g = cabi.rasterize_backward(s.W, s.H, s.N, out["binned"], s.background, out["final_Ts"],
                                out["final_idx"], to_dev(s.v_out), flag)
    torch.cuda.synchronize()
    f, ref = oracle_raster(restated, s, np_(out["xys"]), np_(out["conics"]), np_(out["colors"]),
                           np_(out["cov2d"]), np_(out["depths"]), s.v_out)
    for k in ("v_xy", "v_conic", "v_colors", "v_opacity"):
        assert rel_err(np_(g[k]), ref[k].reshape(np_(g[k]).shape)) < 2e-5, (k, px)
    if px == "q+det":   # order-independent sums: a second run has the same bits
        g2 = cabi.rasterize_backward(s.W, s.H, s.N, out["binned"], s.background, out["final_Ts"],
                                     out["final_idx"], to_dev(s.v_out), flag)
        torch.cuda.synchronize()
        for k in ("v_xy", "v_conic", "v_colors", "v_opacity"):
            assert np.array_equal(np_(g[k]), np_(g2[k])), k


@pytest.mark.parametrize("hot", [False, True])
def test_full_frame_default_backward_matches_oracle_and_the_classic_kernels(hot, restated):
    """A frame of 2 665 tiles — beyond the 2 560 up to which tiles get two or four waves — takes the
    full-frame default: one wave per tile with sixteen four-lane groups (k_rasterize_backward_q), and with a
    few lists far beyond the others (hot: 30 % of the Gaussians in a 40-px window) the launch that hands those
    tiles to four one-pixel-per-lane waves (k_rasterize_backward_q_mixed).  Both against the oracle and
    against the four-group kernels of rounds 2 - 4 (flag bit 26), with fp32 and with fixed-point sums."""
    import torch

    from opensplat_amd import cabi

    s = scenes.camera_scene(60000, 1040, 656, K=0, seed=93, znear=1.0, zfar=100.0, sigma_px=(0.7, 5.0),
                            hot=(0.3, 40) if hot else (0.0, 0))
    out = hip_pipeline(s, backward=False)
    lens = np_(out["binned"].tile_bins)
    lens = lens[:, 1] - lens[:, 0]
    assert lens.size == 65 * 41 and 2 * lens.size > 5120
    if hot:   # the statistics the launch decides on: a list beyond 2 x the mean and 512 entries
        assert lens.max() > max(512, 2 * lens.mean())
        assert out["binned"].list_stats[1] == lens.max()
    f, ref = oracle_raster(restated, s, np_(out["xys"]), np_(out["conics"]), np_(out["colors"]),
                           np_(out["cov2d"]), np_(out["depths"]), s.v_out)
    assert np.array_equal(np_(out["img"]), f["img"])
    res = {}
    for name, flag in (("default", 0), ("classic", CLASSIC), ("default+det", cabi.GS_FLAG_DETERMINISTIC)):
        g = cabi.rasterize_backward(s.W, s.H, s.N, out["binned"], s.background, out["final_Ts"],
                                    out["final_idx"], to_dev(s.v_out), flag)
        torch.cuda.synchronize()
        res[name] = {k: np_(v).copy() for k, v in g.items()}
        for k in ("v_xy", "v_conic", "v_colors", "v_opacity"):
            assert rel_err(res[name][k], ref[k].reshape(res[name][k].shape)) < 2e-5, (name, k)
    for k in ("v_xy", "v_conic", "v_colors", "v_opacity"):
        assert rel_err(res["default"][k], res["classic"][k]) < 2e-5, k
