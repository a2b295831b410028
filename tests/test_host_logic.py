"""CPU: host-side logic — scene generators, gradient buffer layout, byte accounting, helpers."""
import os

import numpy as np
import torch

from opensplat_amd import scenes
from opensplat_amd.dist import GradBuffer


def test_simple_trainer_scene_is_deterministic_and_matches_reference_setup():
    a = scenes.simple_trainer_scene(500, 64, 64, seed=0)
    b = scenes.simple_trainer_scene(500, 64, 64, seed=0)
    assert np.array_equal(a.means, b.means) and np.array_equal(a.quats, b.quats)
    assert a.means.min() >= -1 and a.means.max() <= 1            # simple_trainer.cpp:102
    assert np.allclose(np.linalg.norm(a.quats, axis=-1), 1.0, atol=1e-5)
    assert np.allclose(a.opacities, 1 / (1 + np.exp(-1.0)))     # sigmoid(ones), :128,:166
    assert a.viewmat[2, 3] == 8.0 and np.array_equal(a.viewmat, a.projmat)  # :131-136,:153
    assert abs(a.fx - 32.0) < 1e-9 and a.cx == 32.0                       # focal = 0.5 W / tan(45 deg)
    gt = a.extra["gt_image"]
    assert np.array_equal(gt[0, 0], [1, 0, 0]) and np.array_equal(gt[-1, -1], [0, 0, 1])
    assert np.array_equal(gt[0, -1], [1, 1, 1])


def test_camera_scene_properties():
    s = scenes.camera_scene(4000, 320, 200, K=16, seed=4)
    assert s.sh_coeffs.shape == (4000, 16, 3) and s.dirs.shape == (4000, 3)
    assert np.allclose(np.linalg.norm(s.dirs, axis=-1), 1.0, atol=1e-5)
    assert len(np.unique(s.means[:, 2])) == s.N, "depths must be tie-free"
    assert (s.means[:, 2] > 2.0 - 1e-6).all() and (s.means[:, 2] < 10.0 + 1e-6).all()
    # every Gaussian projects inside the image (2 % margin)
    u = s.means[:, 0] / s.means[:, 2] * s.fx + s.cx
    v = s.means[:, 1] / s.means[:, 2] * s.fy + s.cy
    assert (u > 0).all() and (u < s.W).all() and (v > 0).all() and (v < s.H).all()
    assert s.opacities.min() > 0 and s.opacities.max() < 1
    # OpenGL projection as model.cpp:35-47: last row (0,0,1,0)
    assert np.array_equal(s.projmat[3], [0, 0, 1, 0])
    assert s.degrees_to_use == 3


def test_c4_cameras_share_gaussians():
    a, b = scenes.config_c4(0, 2000), scenes.config_c4(5, 2000)
    assert np.array_equal(a.means, b.means) and np.array_equal(a.sh_coeffs, b.sh_coeffs)
    assert not np.array_equal(a.viewmat, b.viewmat)
    r = a.viewmat[:3, :3]
    assert np.allclose(r @ r.T, np.eye(3), atol=1e-6)


def test_grad_buffer_layout():
    N, K = 7, 16
    g = GradBuffer(N, K, torch.device("cpu"))
    assert g.flat.numel() == N * (3 * K + 3 + 3 + 4 + 1) == N * 59
    assert g.nbytes == 4 * N * 59                                  # 236 B/Gaussian
    g.v_rest.fill_(1); g.v_dc.fill_(6); g.v_means.fill_(2); g.v_scales.fill_(3); g.v_quats.fill_(4)
    g.v_opacity.fill_(5)
    f = g.flat
    o = 0
    for val, n in [(1, N * (K - 1) * 3), (6, N * 3), (2, N * 3), (3, N * 3), (4, N * 4), (5, N)]:
        assert (f[o:o + n] == val).all()
        o += n
    assert g.sh_block().numel() + g.rest_block().numel() == f.numel()
    assert g.sh_block().data_ptr() == f.data_ptr()
    assert g.v_rest.is_contiguous() and g.v_dc.is_contiguous() and g.v_quats.is_contiguous()
    assert abs(g.sh_block().numel() / f.numel() - 48 / 59) < 1e-9  # 81 % of the bytes


def test_algorithmic_bytes_match_survey():
    import bench

    total, per = bench.algorithmic_bytes(1_000_000, 16, 3_200_000, 1920 * 1080)
    assert total == 1_000_000 * (244 + 24 * 16) + 100 * 3_200_000 + 40 * 2_073_600
    assert 1.02e9 < total < 1.04e9                                 # ~1.03 GB
    assert per["rasterize_bwd"] == 40 * 3_200_000 + 20 * 2_073_600 + 36 * 1_000_000


def test_sh_helpers():
    from opensplat_amd import ops

    assert [ops.deg_from_sh(k) for k in (1, 4, 9, 16, 25)] == [0, 1, 2, 3, 4]
    rgb = torch.tensor([0.0, 0.25, 1.0])
    assert torch.allclose(ops.sh2rgb(ops.rgb2sh(rgb)), rgb, atol=1e-6)


def test_trainer_schedules_and_morton_permutation_on_cpu():
    """Host logic of opensplat_amd.train that needs no GPU: the reference's SH-degree and resolution
    schedules (model.cpp:178, 249-251) and the Z-order permutation used at refinement."""
    import types

    import torch

    from opensplat_amd import train

    T = types.SimpleNamespace(K=16, sh_degree_interval=1000, num_downscales=2, resolution_schedule=3000)
    assert [train.Trainer.degrees_to_use(T, s) for s in (1, 999, 1000, 2500, 3000, 30000)] == [0, 0, 1, 2, 3, 3]
    assert [train.Trainer.downscale_factor(T, s) for s in (1, 2999, 3000, 5999, 6000, 30000)] == [4, 4, 2, 2, 1, 1]
    T.K = 4
    assert train.Trainer.degrees_to_use(T, 30000) == 1
    g = torch.Generator().manual_seed(0)
    means = torch.rand((4096, 3), generator=g)
    perm = train.morton_permutation(means)
    assert sorted(perm.tolist()) == list(range(4096))
    m = means[perm]
    assert (m[1:] - m[:-1]).norm(dim=1).mean() < 0.3 * (means[1:] - means[:-1]).norm(dim=1).mean()
    # points of one octant stay together (the curve quantises the data's own bounding box, so a few
    # points next to the 0.5 planes may sit on the other side)
    octant = ((m > 0.5).long() * torch.tensor([1, 2, 4])).sum(1)
    assert (octant[1:] != octant[:-1]).sum() <= 40


def test_host_model_of_the_backward_geometries_reproduces_the_measured_counters():
    """scripts/sim_bwd_geometry.py — the host model that picked the sixteen-group backward over the entry-parallel
    forms — rebuilds lists, coverage masks and last contributors of a window of BASELINE
    config 2's tiles on the CPU.  Its model of the round-4 kernel must stay where the instrumented kernel's
    counters are (profiles/work_stats_r04_c2.json), and its prediction for sixteen groups where the round-5
    kernel's are (profiles/work_stats_r05_c2.json): otherwise the numbers quoted from it mean nothing."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "sim_bwd_geometry.py"), "--config", "c2",
                        "--tiles", "12x8"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    sim = json.loads(r.stdout)
    r4 = json.load(open(os.path.join(root, "profiles", "work_stats_r04_c2.json")))["backward"]
    r5 = json.load(open(os.path.join(root, "profiles", "work_stats_r05_c2.json")))["backward"]
    M4 = json.load(open(os.path.join(root, "profiles", "work_stats_r04_c2.json")))["M"]
    # four 16-lane groups on 8 x 8 blocks: steps and (block, entry) pairs per list entry, passes per step, live lanes
    assert abs(sim["cur"]["steps_per_entry"] / r4["steps_per_list_entry"] - 1.0) < 0.08
    assert abs(sim["cur"]["pairs_per_entry"] / (r4["block_entries"] / M4) - 1.0) < 0.08
    assert abs(sim["cur"]["passes_per_step"] / (r4["steps_with_a_needing_lane"] / r4["steps"]) - 1.0) < 0.05
    assert abs(sim["cur"]["live_lanes_per_pass"] / r4["live_lanes_per_needing_step"] - 1.0) < 0.10
    # sixteen four-lane groups on 4 x 4 blocks, 64-entry chunks: predicted before the kernel existed
    assert abs(sim["q64"]["steps_per_entry"] / r5["steps_per_list_entry"] - 1.0) < 0.08
    assert abs(sim["q64"]["pairs_per_entry"] / (r5["block_entries"] / M4) - 1.0) < 0.08
    assert 0.6 < sim["q64"]["vs_cur"] < 0.75          # the instruction-count ratio the decision rested on
