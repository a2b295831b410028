"""CPU: SURVEY.md §8 row f4 pinned to the reference's OWN compiled `Model` (oracle/ref_model_shim.cpp:
model.cpp compiled in place against declaration-only stand-ins for the three third-party headers it
reaches) — Model::afterTrain's densification, statistics and alpha reset, Model::savePly and saveSplat.
Round 2's pin was those statements restated as free functions (oracle/ref_train_shim.cpp); here the
plain-C oracle the GPU tests use (oracle/densify_oracle.c), that restatement and the on-disk formats of
opensplat_amd/io.py are all held against the compiled class.  Skipped where oracle/_ref is absent."""
import numpy as np
import pytest

from opensplat_amd import io, scenes

N = 600
# (K, step < stopScreenSizeAt, step > refineEvery * resetAlphaEvery, seed)
CASES = [(4, True, True, 1), (1, False, True, 2), (16, True, False, 3), (4, False, False, 4)]


def torch_normal_samples(seed):
    """The samples Model::afterTrain draws itself (torch::randn after torch::manual_seed(seed))."""
    import torch

    def fn(n):
        torch.manual_seed(seed)
        return torch.randn(2 * n, 3).numpy()
    return fn


def model_for(reference, prob, K, check_screen, cull_huge):
    """A compiled Model holding `prob`'s tensors and Adam state, configured so that afterTrain(step)
    refines with the wanted branches: refineEvery 100, warm-up 500, reset interval 30 x 100."""
    step = 3500 if cull_huge else 1000                  # > / <= refineEvery * resetAlphaEvery = 3000
    cfg = dict(shDegree={1: 0, 4: 1, 9: 2, 16: 3}[K], stopScreenSizeAt=(4000 if check_screen else 500),
               densifyGradThresh=0.0002, densifySizeThresh=0.01, splitScreenSize=0.05)
    m = reference.model(prob["params"][0], np.zeros((prob["N"], 3), np.uint8), **cfg)
    m.set_state(prob["params"], prob["exp_avg"], prob["exp_avg_sq"], adam_step=7)
    return m, step


@pytest.mark.parametrize("case", CASES, ids=[f"K{c[0]}_screen{int(c[1])}_huge{int(c[2])}" for c in CASES])
def test_refinement_of_the_compiled_model_equals_the_oracles(case, restated, reference):
    K, cs, ch, seed = case
    prob = scenes.densify_problem(N, K, seed)
    m, step = model_for(reference, prob, K, cs, ch)
    # no visible Gaussian in "this" iteration: the statistics enter the refinement as given
    refined, _ = m.after_train(step, np.zeros(N, np.int32), np.zeros((N, 2), np.float32),
                               (prob["xys_grad_norm"], prob["vis_counts"], prob["max_2d_size"]),
                               prob["height"], prob["width"], seed=seed)
    assert refined
    got = m.get_state()
    smp = torch_normal_samples(seed)
    for name, other in (("plain-C oracle", restated), ("restated under libtorch", reference)):
        r = other.densify_refine(prob, 0.0002, 0.01, cs, 0.05, ch, smp)
        assert r["new_n"] == m.N, name
        assert r["n_splits"] > 50 and r["n_dups"] > 50 and r["culled"] > r["n_splits"]
        for key in ("params", "exp_avg", "exp_avg_sq"):
            for i, (a, b) in enumerate(zip(r[key], got[key])):
                assert a.shape == b.shape, (name, key, i)
                if a.size == 0:
                    continue
                if key == "params" and i in (0, 1):     # split samples: exp, log and a 3 x 3 product
                    assert np.abs(a - b).max() <= 1e-6 * max(np.abs(b).max(), 1.0), (name, key, i)
                else:
                    assert np.array_equal(a, b), (name, key, i)


def test_statistics_of_the_compiled_model_equal_the_oracle(restated, reference):
    """model.cpp:317-337 over four iterations (no refinement: step % refineEvery != 0)."""
    prob = scenes.densify_problem(N, 4, 9)
    m = reference.model(prob["params"][0], np.zeros((N, 3), np.uint8), shDegree=1)
    m.set_state(prob["params"])
    rs = np.random.RandomState(5)
    g = np.zeros(N, np.float32); v = np.zeros(N, np.float32); m2 = np.zeros(N, np.float32)
    stats = None
    for it in range(4):
        grad = (rs.standard_normal((N, 2)) * 1e-4).astype(np.float32)
        rad = (rs.randint(0, 40, N) * (rs.rand(N) < 0.7)).astype(np.int32)
        grad[rad == 0] = 0
        restated.densify_stats(grad, rad, 480, 640, it == 0, g, v, m2)
        refined, stats = m.after_train(7 + it, rad, grad, stats, 480, 640)
        assert not refined
        assert np.array_equal(stats[1], v) and np.array_equal(stats[2], m2)
        assert np.abs(stats[0] - g).max() <= 2e-7 * np.abs(g).max()      # the norm: 1 ulp


def test_alpha_reset_of_the_compiled_model(restated, reference):
    """step % (resetAlphaEvery * refineEvery) == refineEvery (model.cpp:464-479): opacities clamped to
    logit(0.2); the reference builds a zeroed Adam state and never installs it — the moments the
    optimiser holds are unchanged, which is what gs_reset_opacity(NULL moments) does."""
    prob = scenes.densify_problem(N, 4, 11)
    m = reference.model(prob["params"][0], np.zeros((N, 3), np.uint8), shDegree=1)
    m.set_state(prob["params"], prob["exp_avg"], prob["exp_avg_sq"], adam_step=3)
    step = 3100                                          # 3100 % 3000 == 100; no densification (100 <= 10 + 100)
    refined, _ = m.after_train(step, np.zeros(N, np.int32), np.zeros((N, 2), np.float32),
                               (prob["xys_grad_norm"], prob["vis_counts"], prob["max_2d_size"]), 480, 640)
    assert refined and m.N == N
    got = m.get_state(moments=False)
    want = restated.reset_opacity(prob["params"][3].reshape(-1), 0.2).reshape(N, 1)
    assert np.abs(got["params"][3] - want).max() <= 1e-6
    for i in (0, 1, 2, 4, 5):
        assert np.array_equal(got["params"][i], prob["params"][i])


@pytest.mark.parametrize("keep_crs", [False, True])
def test_ply_and_splat_files_of_the_compiled_model(tmp_path, keep_crs, reference):
    prob = scenes.densify_problem(300, 4, 31)
    params = [np.ascontiguousarray(a, np.float32) for a in prob["params"]]
    tr = np.array([1.0, -2.0, 3.0], np.float32)
    m = reference.model(params[0], np.zeros((300, 3), np.uint8), shDegree=1, keepCrs=keep_crs, scale=2.5,
                        translation=tr)
    m.set_state(params)
    m.save(tmp_path / "ref.ply", 1234)
    io.save_ply(tmp_path / "ours.ply", *params, 1234, keep_crs=keep_crs, scale=2.5, translation=tr)
    a, b = (tmp_path / "ours.ply").read_bytes(), (tmp_path / "ref.ply").read_bytes()
    assert len(a) == len(b) and a[:a.find(b"end_header")] == b[:b.find(b"end_header")]
    if not keep_crs:
        assert a == b                                    # byte for byte
    else:                                                # means / scale, log(exp(s) / scale): libtorch's exp / log
        da, _ = io.load_ply(tmp_path / "ours.ply")
        db, _ = io.load_ply(tmp_path / "ref.ply")
        for k in da:
            assert np.abs(da[k] - db[k]).max() <= 1e-6 * max(np.abs(db[k]).max(), 1.0), k
    m.save(tmp_path / "ref.splat")
    io.save_splat(tmp_path / "ours.splat", params[0], params[1], params[2], params[3], params[4],
                  keep_crs=keep_crs, scale=2.5, translation=tr)
    ra = np.frombuffer((tmp_path / "ours.splat").read_bytes(), np.uint8).reshape(-1, 32)
    rb = np.frombuffer((tmp_path / "ref.splat").read_bytes(), np.uint8).reshape(-1, 32)
    assert ra.shape == rb.shape == (300, 32)
    # same order (sorted by size x opacity) and the same floats; the 8-bit fields within one count
    # where exp rounding decides a truncation
    fa, fb = ra[:, :24].copy().view(np.float32), rb[:, :24].copy().view(np.float32)
    assert np.abs(fa - fb).max() <= 1e-6 * np.abs(fb).max()
    assert np.abs(ra[:, 24:].astype(int) - rb[:, 24:].astype(int)).max() <= 1


def test_points_tensor_scales_stand_in(reference):
    """Model's constructor calls PointsTensor::scales() (kdtree_tensor.cpp needs nanoflann): the brute-
    force stand-in of the shim against numpy — and colmap.init_from_points, which restates it."""
    from opensplat_amd import colmap

    rs = np.random.RandomState(3)
    xyz = rs.uniform(-1, 1, (200, 3)).astype(np.float32)
    rgb = rs.randint(0, 256, (200, 3)).astype(np.uint8)
    m = reference.model(xyz, rgb, shDegree=2)
    st = m.get_state(moments=False)["params"]
    d = np.sqrt(((xyz[:, None, :] - xyz[None, :, :]) ** 2).sum(-1))
    knn = np.sort(d, axis=1)[:, 1:4].mean(1)
    assert np.abs(np.exp(st[1][:, 0]) - knn).max() <= 1e-5 * knn.max()
    assert np.array_equal(st[0], xyz) and m.K == 9
    # ... and Model's whole constructor (model.hpp:33-53), which colmap.init_from_points restates
    init = colmap.init_from_points(xyz, rgb, sh_degree=2)   # [means, log_scales, quats, logits, dc, rest]
    assert np.abs(init[1] - st[1]).max() < 1e-5
    assert np.abs(init[4] - st[4]).max() < 1e-6
    assert np.abs(init[3] - st[3]).max() < 1e-6
    assert np.abs(init[2] - st[2]).max() < 1e-6             # same generator stream (torch seed 42)
    assert np.array_equal(init[0], st[0]) and not st[5].any() and init[5].shape == st[5].shape
