"""-m gpu: the LAUNCHER-LEVEL binding (opensplat_amd/csrc/bindings_hip_native.cpp): the eight
`*_tensor` functions of rasterizer/gsplat/bindings.h, driven exactly as OpenSplat's own operator
files drive them (rasterize_gaussians.cpp:6-37,56-75: cumsum -> map_gaussian_to_intersects_tensor ->
torch::sort -> gather -> get_tile_bin_edges_tensor -> rasterize_forward_tensor; :100-124
rasterize_backward_tensor), compared with the native path of this repo.

At this level tiles are assigned by the GPU reference's radius square (forward.cu:86-94), natively by
the tightened CPU pixel rectangle.  The launcher image and gradients are compared with the ORACLE under the
same contract (oracle/gsplat_oracle.c orc_rasterize_forward_tiles: gsplat-cpu's per-pixel decisions, each
Gaussian confined to the tiles of its radius square — the lists the reference's own glue hands over) on the
WHOLE scene, bit for bit in the forward (VERDICT r03 "next" 9: round 3 compared HIP with HIP after zeroing
the Gaussians on which the two contracts disagree).  The native path is compared too, on the sub-population
where both contracts coincide."""
import numpy as np
import pytest

from opensplat_amd import scenes
from tests.util import np_, rel_err, to_dev

pytestmark = pytest.mark.gpu


def _glue_bin_and_sort(T, xys, depths, radii, num_tiles_hit, tiles_x, tiles_y):
    """rasterize_gaussians.cpp:6-37,62-66 with torch ops, as the reference runs it."""
    import torch

    cum = torch.cumsum(num_tiles_hit, 0, dtype=torch.int32)
    M = int(cum[-1].item())
    ids, gids = T.launcher_map_gaussian_to_intersects(M, xys, depths, radii, cum, tiles_x, tiles_y)
    ids_sorted, order = torch.sort(ids)
    gids_sorted = torch.gather(gids, 0, order)
    bins = T.launcher_get_tile_bin_edges(M, ids_sorted)
    return M, ids_sorted, gids_sorted.contiguous(), bins


def test_operator_level_bin_and_sort_gaussians_keeps_the_references_five_tuple():
    """binAndSortGaussians with the reference's signature and contract (rasterize_gaussians.hpp:11-20):
    (isectIds, gaussianIds, isectIdsSorted, gaussianIdsSorted, tileBins) from caller-side cumulative tile
    counts — checked against numpy's evaluation of the same definition (tile << 32 | depth bits, stable
    order of equal keys not required by the contract)."""
    import torch

    from opensplat_amd import ops  # noqa: F401  (loads libgsplat_torch.so)
    from tests.util import hip_pipeline

    T = torch.ops.opensplat_amd
    s = scenes.camera_scene(5000, 320, 200, K=0, seed=81, znear=1.0, zfar=100.0)
    p = hip_pipeline(s, backward=False)
    W, H = s.W, s.H
    tx, ty = (W + 15) // 16, (H + 15) // 16
    xys, depths, radii = p["xys"], p["depths"], p["radii"]
    # radius-square tile counts, as the reference's projection kernel reports them (helpers.cuh get_tile_bbox)
    x, y, r = np_(xys)[:, 0], np_(xys)[:, 1], np_(radii).astype(np.float32)
    x0 = np.clip(np.floor((x - r) / 16), 0, tx).astype(np.int64); x1 = np.clip(np.floor((x + r + 16) / 16), 0, tx).astype(np.int64)
    y0 = np.clip(np.floor((y - r) / 16), 0, ty).astype(np.int64); y1 = np.clip(np.floor((y + r + 16) / 16), 0, ty).astype(np.int64)
    hit = np.where(r > 0, (x1 - x0) * (y1 - y0), 0)
    cum = torch.from_numpy(np.cumsum(hit).astype(np.int32)).cuda()
    M = int(hit.sum())
    ids, gids, ids_sorted, gids_sorted, bins = T.bin_and_sort_gaussians(len(x), M, xys, depths, radii, cum, tx, ty)
    torch.cuda.synchronize()
    assert ids.dtype == torch.int64 and gids.dtype == torch.int32 and bins.dtype == torch.int32
    assert ids.shape == (M,) and gids.shape == (M,) and ids_sorted.shape == (M,) and gids_sorted.shape == (M,)
    ids_n, gids_n, ids_s, gids_s, bins_n = (np_(t) for t in (ids, gids, ids_sorted, gids_sorted, bins))
    # the unsorted pairs: every (tile, Gaussian) of the radius square once, key = tile << 32 | depth bits
    dbits = np_(depths).view(np.int32).astype(np.int64)
    want = []
    for g in np.nonzero(hit)[0]:
        for yy in range(y0[g], y1[g]):
            for xx in range(x0[g], x1[g]):
                want.append((((yy * tx + xx) << 32) | dbits[g], g))
    want = np.array(sorted(want), dtype=np.int64)
    got = np.stack([ids_n, gids_n.astype(np.int64)], -1)
    assert np.array_equal(got[np.lexsort((got[:, 1], got[:, 0]))], want)
    # sorted by key; the ids follow their keys; the bins delimit the tiles
    assert np.all(np.diff(ids_s) >= 0) and np.array_equal(np.sort(ids_n), ids_s)
    pair = {(int(k), int(g)) for k, g in zip(ids_n, gids_n)}
    assert all((int(k), int(g)) in pair for k, g in zip(ids_s[::97], gids_s[::97]))
    tiles = ids_s >> 32
    for t in np.unique(tiles)[::7]:
        lo, hi = bins_n[t]
        assert np.all(tiles[lo:hi] == t) and (lo == 0 or tiles[lo - 1] < t) and (hi == M or tiles[hi] > t)


@pytest.mark.parametrize("W,H,N", [(320, 200, 6000), (203, 117, 3000)])
def test_reference_operator_glue_on_the_launcher_functions(W, H, N):
    import torch

    from opensplat_amd import cabi, ops  # noqa: F401  (loads libgsplat_torch.so)

    T = torch.ops.opensplat_amd
    s = scenes.camera_scene(N, W, H, K=16, seed=61, znear=1.0, zfar=100.0, yaw_deg=3.0, sigma_px=(0.6, 5.0))
    tiles_x, tiles_y = (W + 15) // 16, (H + 15) // 16
    means, scales, quats = to_dev(s.means), to_dev(s.scales), to_dev(s.quats)
    vm, pm = to_dev(s.viewmat), to_dev(s.projmat)
    cov3d, xys, depths, radii, conics, tiles_hit = T.launcher_project_gaussians_forward(
        means, scales, 1.0, quats, vm, pm, s.fx, s.fy, s.cx, s.cy, H, W, tiles_x, tiles_y, 0.01)
    cam = cabi.make_camera(s.viewmat, s.projmat, s.fx, s.fy, s.cx, s.cy, W, H)
    p = cabi.project_forward(cam, means, scales, quats)
    for a, b in ((xys, p["xys"]), (depths, p["depths"]), (radii, p["radii"]), (conics, p["conics"]),
                 (cov3d, p["cov3d"])):
        assert torch.equal(a, b)
    dirs, coeffs = to_dev(s.dirs), to_dev(s.sh_coeffs)
    sh = T.launcher_compute_sh_forward(3, s.degrees_to_use, dirs, coeffs)
    assert torch.equal(sh, cabi.sh_forward(s.degrees_to_use, dirs, coeffs))
    colors = torch.clamp_min(sh + 0.5, 0.0)
    opac = to_dev(s.opacities.reshape(-1, 1)).clone()

    # ---- launcher level on the WHOLE scene against the oracle under the radius-square contract ----
    from oracle import restated as _restated

    O = _restated()
    xn, cn, rn = np_(xys), np_(conics), np_(radii).astype(np.float32)
    trunc_ = lambda v: np.trunc(v).astype(np.int64)
    # helpers.cuh:17-49 get_tile_bbox: tile_center = xy / 16, tile_radius = radius / 16, min = (int)(c - r),
    # max = (int)(c + r + 1), both clamped to [0, tiles]
    tx0 = np.clip(trunc_(xn[:, 0] / 16 - rn / 16), 0, tiles_x); tx1 = np.clip(trunc_(xn[:, 0] / 16 + rn / 16 + 1), 0, tiles_x)
    ty0 = np.clip(trunc_(xn[:, 1] / 16 - rn / 16), 0, tiles_y); ty1 = np.clip(trunc_(xn[:, 1] / 16 + rn / 16 + 1), 0, tiles_y)
    tile_rect = np.stack([tx0, tx1, ty0, ty1], -1).astype(np.int32)
    tile_rect[rn <= 0] = 0
    assert np.array_equal(np.where(rn > 0, (tx1 - tx0) * (ty1 - ty0), 0), np_(tiles_hit))
    # the launcher's pixel rectangle comes from conic^-1 (no cov2d at this level): xx = C / det, yy = A / det
    A_, B_, C_ = cn[:, 0], cn[:, 1], cn[:, 2]
    det_ = A_ * C_ - B_ * B_
    cov = np.zeros((N, 2, 2), np.float32)
    cov[:, 0, 0] = C_ / det_
    cov[:, 1, 1] = A_ / det_
    Mw, idsw, gidsw, binsw = _glue_bin_and_sort(T, xys, depths, radii, tiles_hit, tiles_x, tiles_y)
    bgw = to_dev(s.background)
    imgw, Tsw, fidxw = T.launcher_rasterize_forward(tiles_x, tiles_y, W, H, gidsw, binsw, xys, conics, colors,
                                                    opac, bgw)
    gw = T.launcher_rasterize_backward(H, W, gidsw, binsw, xys, conics, colors, opac, bgw, Tsw, fidxw,
                                       to_dev(s.v_out), torch.zeros((H, W), device="cuda"))
    torch.cuda.synchronize()
    fo = O.rasterize_forward(W, H, xn, cn, np_(colors), np_(opac).reshape(-1), s.background, cov, np_(depths),
                             want_contributors=False, tile_rect=tile_rect)
    assert np.array_equal(np_(imgw), fo["img"]), "launcher image differs from the oracle under the radius-square contract"
    assert np.array_equal(np_(Tsw), fo["final_Ts"])
    go = O.rasterize_backward(W, H, xn, cn, np_(colors), np_(opac).reshape(-1), s.background, cov, np_(depths),
                              fo["final_Ts"], fo["state"], s.v_out)
    for a, name in zip(gw, ("v_xy", "v_conic", "v_colors", "v_opacity")):
        assert rel_err(np_(a).reshape(go[name].shape), go[name]) < 2e-5, name
    # and the contract matters: without the tile confinement the oracle renders another image
    ff = O.rasterize_forward(W, H, xn, cn, np_(colors), np_(opac).reshape(-1), s.background, cov, np_(depths),
                             want_contributors=False)
    O.rasterize_free(ff["state"])
    contract_matters = not np.array_equal(ff["img"], fo["img"])

    # ---- launcher against the native path: Gaussians whose tightened rectangle reaches a tile outside
    #      their radius square are taken out of the scene, after which both composite the same lists ----
    nat = cabi.bin_and_sort(W, H, xys, depths, radii, conics, colors, opac.reshape(-1).contiguous(), None)
    pk = np_(nat.packed).view(np.uint32)
    rx, ry = pk[:, 7], pk[:, 11]
    x0, x1, y0, y1 = rx & 0xFFFF, rx >> 16, ry & 0xFFFF, ry >> 16
    c, r = np_(xys), np_(radii).astype(np.float32)
    trunc = lambda v: np.trunc(v).astype(np.int64)
    sx0 = np.clip(trunc(c[:, 0] / 16 - r / 16), 0, tiles_x); sx1 = np.clip(trunc(c[:, 0] / 16 + r / 16 + 1), 0, tiles_x)
    sy0 = np.clip(trunc(c[:, 1] / 16 - r / 16), 0, tiles_y); sy1 = np.clip(trunc(c[:, 1] / 16 + r / 16 + 1), 0, tiles_y)
    empty = (x1 <= x0) | (y1 <= y0)
    inside = (x0 // 16 >= sx0) & ((x1 + 15) // 16 <= sx1) & (y0 // 16 >= sy0) & ((y1 + 15) // 16 <= sy1)
    bad = ~(inside | empty)
    assert bad.mean() < 0.2
    assert contract_matters or not bad.any()
    opac[torch.from_numpy(bad).cuda()] = 0.0

    # ---- launcher level, driven like rasterize_gaussians.cpp ----
    M, ids_sorted, gids_sorted, bins = _glue_bin_and_sort(T, xys, depths, radii, tiles_hit, tiles_x, tiles_y)
    assert M == int(tiles_hit.sum()) and M > N
    k = np_(ids_sorted)
    assert (np.diff(k) >= 0).all()
    tile_of = (k >> 32).astype(np.int64)
    b = np_(bins)[: tiles_x * tiles_y]
    for t in np.unique(tile_of)[:40]:          # [start, end) of every tile's run
        assert (tile_of[b[t, 0]:b[t, 1]] == t).all() and (b[t, 1] - b[t, 0]) == (tile_of == t).sum()
    bg = to_dev(s.background)
    img, Ts, fidx = T.launcher_rasterize_forward(tiles_x, tiles_y, W, H, gids_sorted, bins, xys, conics, colors,
                                                 opac, bg)
    v_out = to_dev(s.v_out)
    g = T.launcher_rasterize_backward(H, W, gids_sorted, bins, xys, conics, colors, opac, bg, Ts, fidx, v_out,
                                      torch.zeros((H, W), device="cuda"))

    # ---- native path on the same 2-D inputs (rectangle derived from the conics, like the launcher's) ----
    nat = cabi.bin_and_sort(W, H, xys, depths, radii, conics, colors, opac.reshape(-1).contiguous(), None)
    f = cabi.rasterize_forward(W, H, nat, s.background)
    gn = cabi.rasterize_backward(W, H, N, nat, s.background, f["final_Ts"], f["final_idx"], v_out)
    torch.cuda.synchronize()
    assert torch.equal(img, f["img"]) and torch.equal(Ts, f["final_Ts"])
    for a, name in zip(g, ("v_xy", "v_conic", "v_colors", "v_opacity")):
        assert rel_err(np_(a).reshape(np_(gn[name]).shape), np_(gn[name])) < 2e-5, name
    assert tuple(g[3].shape) == (N, 1)                      # bindings.cu:596: v_opacity [N, 1]

    # ---- the remaining two launchers against the stage kernels ----
    v_coeffs = T.launcher_compute_sh_backward(3, s.degrees_to_use, dirs, g[2].contiguous())
    assert torch.equal(v_coeffs, cabi.sh_backward(s.degrees_to_use, 16, dirs, g[2].contiguous()))
    pb = T.launcher_project_gaussians_backward(means, scales, 1.0, quats, vm, pm, s.fx, s.fy, s.cx, s.cy, H, W,
                                               cov3d, radii, conics, g[0], torch.zeros(N, device="cuda"), g[1])
    ref = cabi.project_backward(cam, means, scales, quats, radii, g[0].contiguous(), g[1].contiguous(),
                                torch.zeros(N, device="cuda"))
    assert torch.equal(pb[2], ref["v_means"]) and torch.equal(pb[3], ref["v_scales"]) and torch.equal(pb[4], ref["v_quats"])
    assert not pb[0].any() and not pb[1].any()              # v_cov2d / v_cov3d: unused scratch outputs
