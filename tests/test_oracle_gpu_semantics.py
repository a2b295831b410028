"""CPU: orc_project_gpu_semantics (oracle/gsplat_oracle.c) — the near-plane cull and the principal
point the product takes from the reference's GPU path — against an independent
numpy restatement of rasterizer/gsplat/helpers.cuh:13-15,112-122,225-233 and forward.cu:49-52."""
import numpy as np

from opensplat_amd import scenes


def test_gpu_semantics_restatement(restated):
    s = scenes.camera_scene(4000, 320, 200, K=1, seed=4, yaw_deg=5.0)
    s.means[0::9, 2] = -0.5          # behind the camera
    s.means[1::9] *= 0.001           # in front of it, inside the clip distance
    cx, cy = s.W / 2.0 - 9.75, s.H / 2.0 + 4.5
    with np.errstate(all="ignore"):
        o = restated.project_forward(s.means, s.scales, s.quats, s.viewmat, s.projmat, s.fx, s.fy,
                                     cx, cy, s.H, s.W)
    g = restated.project_gpu_semantics(o, s.means, s.viewmat, s.projmat, cx, cy, s.H, s.W, clip=0.01)
    f = np.float32
    # clip_near_plane: p_view = transform_4x3(viewmat, p); p_view.z <= thresh -> culled
    m = s.means.astype(f)
    vm, pm = s.viewmat.astype(f), s.projmat.astype(f)
    pz = ((vm[2, 0] * m[:, 0] + vm[2, 1] * m[:, 1]) + vm[2, 2] * m[:, 2]) + vm[2, 3]
    vis = ~(pz <= f(0.01))
    assert np.array_equal(g["visible"], vis) and 0.1 < (~vis).mean() < 0.3
    assert np.all(g["radii"][~vis] == 0) and np.array_equal(g["radii"][vis], o["radii"][vis])
    # project_pix: rw = 1 / (w + 1e-6); ndc2pix(x, W, cx) = 0.5 W x + cx - 0.5
    h = m @ pm[:, :3].T + pm[:, 3]
    rw = f(1.0) / (h[:, 3] + f(1e-6))
    px = f(0.5) * f(s.W) * (h[:, 0] * rw) + f(cx) - f(0.5)
    py = f(0.5) * f(s.H) * (h[:, 1] * rw) + f(cy) - f(0.5)
    want = np.stack([px, py], -1)
    tol = 1e-3 + 1e-6 * np.abs(want[vis])
    assert np.all(np.abs(g["xys_gpu_formula"][vis] - want[vis]) <= tol)    # (summation order)
    # the offset form the product uses: CPU pixel centre + (cx - W/2).  Differences: fp32 round-off
    # and the perspective divide, 1 / max(w, 1e-6) (gsplat_cpu.cpp:121) against 1 / (w + 1e-6): at most
    # 1e-6 / clip = 1e-4 relative for a Gaussian that survives the cull
    tol = 2e-3 + 1.2e-4 * (np.abs(want[vis]) + max(s.W, s.H))
    assert np.all(np.abs(g["xys"][vis] - want[vis]) <= tol)
    inside = vis & (np.abs(want - np.array([s.W / 2, s.H / 2], f)) < np.array([s.W, s.H], f)).all(-1) & (pz > 0.5)
    assert inside.sum() > 2000 and np.abs(g["xys"][inside] - want[inside]).max() < 2e-3
    assert np.array_equal(g["xys"], o["xys"] + np.array([cx - 0.5 * s.W, cy - 0.5 * s.H], f))
    # centred principal point: the CPU path's values, bit for bit
    g0 = restated.project_gpu_semantics(o, s.means, s.viewmat, s.projmat, s.W / 2.0, s.H / 2.0, s.H, s.W)
    assert np.array_equal(g0["xys"][vis], o["xys"][vis])
