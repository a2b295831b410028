"""GPU (MI355X): SURVEY.md §8 row f4 through the C ABI (include/gsplat_densify.h) — per-iteration
statistics, the split / duplicate / cull refinement with its optimiser-state surgery, the alpha
reset — against the CPU oracle (oracle/densify_oracle.c, pinned to the reference's statements under
libtorch by tests/test_densify_oracle.py) and the stored reference vectors."""
import os

import numpy as np
import pytest
import torch

from opensplat_amd import cabi, scenes

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "densify.npz"))
DEV = "cuda:0"
CASES = [(4, True, True, 1), (1, False, True, 2), (16, True, False, 3), (4, False, False, 4)]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def run_gpu(prob, cfg, samples_np):
    P = [dev(a) for a in prob["params"]]
    M = [dev(a) for a in prob["exp_avg"]]
    V = [dev(a) for a in prob["exp_avg_sq"]]
    fn = lambda n: dev(samples_np(n))
    p, m, v, c = cabi.densify(cfg, P, M, V, dev(prob["xys_grad_norm"]), dev(prob["vis_counts"]),
                              dev(prob["max_2d_size"]), fn)
    torch.cuda.synchronize()
    np_ = lambda lst: [t.cpu().numpy() for t in lst]
    return dict(params=np_(p), exp_avg=np_(m), exp_avg_sq=np_(v)), c


def check_set(r, ref):
    """Copied rows bit-equal; split samples' means / scales (exp, log, 3x3 product) 1e-6 relative."""
    for key in ("params", "exp_avg", "exp_avg_sq"):
        for i, (a, b) in enumerate(zip(r[key], ref[key])):
            assert a.shape == b.shape, (key, i, a.shape, b.shape)
            if a.size == 0:
                continue
            if key == "params" and i in (0, 1):
                assert np.abs(a - b).max() <= 1e-6 * max(np.abs(b).max(), 1.0), (key, i)
            else:
                assert np.array_equal(a, b), (key, i)


@pytest.mark.parametrize("case", CASES, ids=[f"K{c[0]}_screen{int(c[1])}_huge{int(c[2])}" for c in CASES])
def test_refine_matches_reference_vectors(case):
    K, cs, ch, seed = case
    prob = scenes.densify_problem(600, K, seed)
    smp = lambda n: np.random.RandomState(seed + 100).standard_normal((2 * n, 3)).astype(np.float32)
    cfg = cabi.densify_config(prob["width"], prob["height"], 0.0002, 0.01, cs, 0.05, ch)
    r, c = run_gpu(prob, cfg, smp)
    tag = f"c{seed}"
    assert [c["n_splits"], c["n_dups"], c["new_n"], c["culled"]] == list(GOLD[f"{tag}_counts"])
    assert c["added"] == 2 * c["n_splits"] + c["n_dups"]
    assert c["new_n"] == c["kept_orig"] + 2 * c["kept_split"] + c["kept_dup"]
    ref = {key: [GOLD[f"{tag}_{key}{i}"] for i in range(6)] for key in ("params", "exp_avg", "exp_avg_sq")}
    check_set(r, ref)


@pytest.mark.parametrize("N", [1, 63, 1024, 1025, 5000, 70001])
def test_refine_matches_oracle_ragged_sizes(N, restated):
    prob = scenes.densify_problem(N, 4, seed=N)
    smp = lambda n: np.random.RandomState(N).standard_normal((2 * n, 3)).astype(np.float32)
    cfg = cabi.densify_config(prob["width"], prob["height"], 0.0002, 0.01, True, 0.05, True)
    r, c = run_gpu(prob, cfg, smp)
    o = restated.densify_refine(prob, 0.0002, 0.01, True, 0.05, True, smp)
    assert (c["n_splits"], c["n_dups"], c["new_n"], c["culled"]) == \
           (o["n_splits"], o["n_dups"], o["new_n"], o["culled"])
    check_set(r, o)


def test_refine_degenerate_outcomes(restated):
    smp = lambda n: np.zeros((2 * n, 3), np.float32)
    # nothing to do: low gradients, healthy opacities -> identity
    prob = scenes.densify_problem(3000, 4, seed=5)
    prob["xys_grad_norm"][:] = 0.0
    prob["params"][3][:] = 2.0
    cfg = cabi.densify_config(prob["width"], prob["height"], cull_huge=False)
    r, c = run_gpu(prob, cfg, smp)
    assert c["n_splits"] == 0 and c["n_dups"] == 0 and c["new_n"] == 3000 and c["culled"] == 0
    for i in range(6):
        assert np.array_equal(r["params"][i], prob["params"][i])
        assert np.array_equal(r["exp_avg"][i], prob["exp_avg"][i])
    # everything faint: all culled, also the new ones (they inherit the opacity)
    prob = scenes.densify_problem(2500, 4, seed=6)
    prob["params"][3][:] = -5.0
    r, c = run_gpu(prob, cabi.densify_config(prob["width"], prob["height"]), smp)
    assert c["new_n"] == 0 and c["culled"] == 2500 + c["added"]
    assert all(a.shape[0] == 0 for a in r["params"])
    # no optimiser state given: moments of the new set are not produced
    prob = scenes.densify_problem(1500, 4, seed=7)
    P = [dev(a) for a in prob["params"]]
    p, m, v, c = cabi.densify(cabi.densify_config(prob["width"], prob["height"]), P, None, None,
                              dev(prob["xys_grad_norm"]), dev(prob["vis_counts"]),
                              dev(prob["max_2d_size"]), lambda n: torch.zeros((2 * n, 3), device=DEV))
    assert m is None and v is None and p[0].shape[0] == c["new_n"]


def test_refine_one_million_properties():
    """BASELINE size: structural properties instead of the (slow) serial oracle — counts are
    consistent, survivors are an order-preserving subsequence of the originals with their moments,
    new rows have zero moments, duplicates are exact copies."""
    N, K = 1_000_000, 16
    prob = scenes.densify_problem(N, K, seed=11)
    # tag every Gaussian through featuresDc[:, 0] so that rows can be traced
    prob["params"][4][:, 0] = np.arange(N, dtype=np.float32)
    smp = lambda n: np.random.RandomState(1).standard_normal((2 * n, 3)).astype(np.float32)
    cfg = cabi.densify_config(prob["width"], prob["height"])
    r, c = run_gpu(prob, cfg, smp)
    new_n, ko, ks, kd = c["new_n"], c["kept_orig"], c["kept_split"], c["kept_dup"]
    assert new_n == ko + 2 * ks + kd and c["culled"] == N + c["added"] - new_n
    tag = r["params"][4][:, 0].astype(np.int64)
    assert np.all(np.diff(tag[:ko]) > 0)                                    # originals keep their order
    assert np.array_equal(tag[ko:ko + ks], tag[ko + ks:ko + 2 * ks])         # sample-major pairs
    assert np.all(np.diff(tag[ko:ko + ks]) > 0) and np.all(np.diff(tag[ko + 2 * ks:]) > 0)
    assert not np.intersect1d(tag[:ko], tag[ko:ko + ks]).size               # split sources are culled
    for i in range(6):
        src_rows = prob["params"][i][tag]
        if i not in (0, 1):
            assert np.array_equal(r["params"][i], src_rows)
        assert np.array_equal(r["exp_avg"][i][:ko], prob["exp_avg"][i][tag[:ko]])
        assert not r["exp_avg"][i][ko:].any() and not r["exp_avg_sq"][i][ko:].any()
    assert np.array_equal(r["params"][0][ko + 2 * ks:], prob["params"][0][tag[ko + 2 * ks:]])  # dups
    # split samples: scale = log(exp(s) / 1.6), means within a few sigma of the source
    s_src = prob["params"][1][tag[ko:ko + ks]]
    assert np.abs(r["params"][1][ko:ko + ks] - (s_src - np.log(np.float32(1.6)))).max() < 1e-5
    d = np.abs(r["params"][0][ko:ko + ks] - prob["params"][0][tag[ko:ko + ks]]).max(axis=1)
    assert np.all(d <= 6.0 * np.exp(s_src).max(axis=1) + 1e-6)


def test_stats_match_oracle_and_reference_vectors(restated):
    N = 600
    rs = np.random.RandomState(5)
    g = torch.zeros(N, device=DEV); v = torch.zeros(N, device=DEV); m = torch.zeros(N, device=DEV)
    go = np.zeros(N, np.float32); vo = np.zeros(N, np.float32); mo = np.zeros(N, np.float32)
    for it in range(4):
        grad = (rs.standard_normal((N, 2)) * 1e-4).astype(np.float32)
        rad = (rs.randint(0, 40, N) * (rs.rand(N) < 0.7)).astype(np.int32)
        grad[rad == 0] = 0
        cabi.densify_stats(dev(grad), dev(rad), 640.0, it == 0, g, v, m)
        restated.densify_stats(grad, rad, 480, 640, it == 0, go, vo, mo)
    torch.cuda.synchronize()
    assert np.array_equal(g.cpu().numpy(), go)      # same IEEE operations as the C oracle
    assert np.array_equal(v.cpu().numpy(), vo) and np.array_equal(m.cpu().numpy(), mo)
    assert np.array_equal(v.cpu().numpy(), GOLD["stats_vis"])
    assert np.array_equal(m.cpu().numpy(), GOLD["stats_m2d"])
    assert np.abs(g.cpu().numpy() - GOLD["stats_gnorm"]).max() <= 2e-7 * np.abs(go).max()


def test_reset_opacity(restated):
    x = np.linspace(-6, 6, 1001, dtype=np.float32)
    t = dev(x)
    m1, m2 = torch.ones_like(t), torch.ones_like(t)
    cabi.reset_opacity(t, 0.2, m1, m2)
    assert np.array_equal(t.cpu().numpy(), restated.reset_opacity(x, 0.2))
    assert not m1.any() and not m2.any()
    t2 = dev(x)
    cabi.reset_opacity(t2, 0.2)          # the reference's actual behaviour: moments untouched
    assert np.array_equal(t2.cpu().numpy(), t.cpu().numpy())


def test_trainer_after_train_schedule_and_refinement(restated):
    """Model::afterTrain's schedule driven by Trainer on a small scene with a fast schedule: the
    statistics accumulate, step 12 refines (checked against the oracle on the captured state, same
    normal samples), the alpha reset fires at step % reset_interval == refine_every, training
    continues on the grown set."""
    from opensplat_amd import train

    s = scenes.camera_scene(4000, 256, 192, K=4, seed=81, znear=1.0, zfar=100.0, degrees_to_use=1)
    raw = scenes.raw_parameters(s)
    cam = dict(viewmat=s.viewmat, projmat=s.projmat, fx=s.fx, fy=s.fy, cx=s.cx, cy=s.cy, W=s.W, H=s.H)
    _, gt_np = scenes.loss_images(s.W, s.H, seed=3)
    gt = dev(gt_np)
    T = train.Trainer(*raw, torch.device(DEV), max_steps=200, refine_every=4, warmup_length=5,
                      reset_alpha_every=5, densify_grad_thresh=2e-6, densify_size_thresh=0.01,
                      stop_screen_size_at=100, num_cameras=1)
    # reset_interval = 20; densify when step % 20 > 5: steps 8, 12, 16; alpha reset at step 24
    events = {}
    for step in range(1, 26):
        T.train_step(cam, gt, s.background, s.degrees_to_use)
        if step == 8:   # capture the state the refinement will see
            torch.cuda.synchronize()
            np_ = lambda lst: [t.detach().cpu().numpy().copy() if t is not None else None for t in lst]
            snap = dict(params=np_(T._param_list(T.params)), exp_avg=np_(T._param_list(T.exp_avg)),
                        exp_avg_sq=np_(T._param_list(T.exp_avg_sq)),
                        stats=[t.cpu().numpy().copy() for t in T._stats],
                        v_xy=T.rgrads["v_xy"].cpu().numpy().copy(),
                        radii=T.proj["radii"].cpu().numpy().copy())
        c = T.after_train(step)
        if c is not None:
            events[step] = (c, T.N)
    torch.cuda.synchronize()
    assert sorted(events) == [8, 12, 16], sorted(events)   # 20 % 20 = 0 and 24 % 20 = 4 are not > 5
    c8, n8 = events[8]
    assert c8["added"] > 0 and n8 == c8["new_n"]
    # oracle on the captured state
    gn, vc, m2 = [a.copy() for a in snap["stats"]]
    restated.densify_stats(snap["v_xy"], snap["radii"], s.H, s.W, False, gn, vc, m2)
    prob = dict(params=snap["params"], exp_avg=snap["exp_avg"], exp_avg_sq=snap["exp_avg_sq"],
                xys_grad_norm=gn, vis_counts=vc, max_2d_size=m2, width=s.W, height=s.H, K=4, N=4000)
    gen = torch.Generator(device=DEV).manual_seed(1_000_003 * 8)
    smp = lambda n: torch.randn((2 * n, 3), device=DEV, generator=gen).cpu().numpy()
    o = restated.densify_refine(prob, 2e-6, 0.01, True, 0.05, False, smp)
    assert (o["n_splits"], o["n_dups"], o["new_n"]) == (c8["n_splits"], c8["n_dups"], c8["new_n"])
    # ... and the run continued on the refined set with finite losses and consistent buffers
    assert T.N == events[16][1]
    for buf in (T.params, T.grads, T.exp_avg, T.exp_avg_sq):
        assert buf.flat.numel() == T.N * (3 * 4 + 11) and torch.isfinite(buf.flat).all()
    assert T.opacity_logits.max().item() <= np.log(0.2 / 0.8) + 1.0   # reset at step 24, one Adam step since


def test_libtorch_densify_operators_equal_the_cabi_path():
    """torch_ops.cpp: densifyStats / densify (C++ host, torch::randn for the samples) against the
    ctypes-level path fed the same random stream."""
    from opensplat_amd import ops

    prob = scenes.densify_problem(5000, 9, seed=41)
    P = [dev(a) for a in prob["params"]]
    M = [dev(a) for a in prob["exp_avg"]]
    V = [dev(a) for a in prob["exp_avg_sq"]]
    stats = (dev(prob["xys_grad_norm"]), dev(prob["vis_counts"]), dev(prob["max_2d_size"]))
    torch.manual_seed(123)
    p1, m1, v1, c1 = ops.densify(P, M, V, stats, prob["width"], prob["height"])
    torch.manual_seed(123)
    cfg = cabi.densify_config(prob["width"], prob["height"])
    p2, m2, v2, c2 = cabi.densify(cfg, P, M, V, *stats)
    torch.cuda.synchronize()
    assert (c1["n_splits"], c1["n_dups"], c1["added"], c1["culled"]) == \
           (c2["n_splits"], c2["n_dups"], c2["added"], c2["culled"])
    for a, b in zip(p1 + m1 + v1, p2 + m2 + v2):
        assert torch.equal(a, b)
    # no optimiser state
    p3, m3, v3, c3 = ops.densify(P, None, None, stats, prob["width"], prob["height"])
    assert m3 is None and p3[0].shape[0] == c2["new_n"]
    # statistics: first call allocates, second accumulates
    N = 700
    rs = np.random.RandomState(2)
    g = dev((rs.standard_normal((N, 2)) * 1e-4).astype(np.float32))
    r = dev((rs.randint(0, 30, N) * (rs.rand(N) < 0.6)).astype(np.int32))
    st = ops.densify_stats(g, r, 480, 640)
    st = ops.densify_stats(g, r, 480, 640, st)
    a, b, c = [torch.zeros(N, device=DEV) for _ in range(3)]
    cabi.densify_stats(g, r, 640.0, True, a, b, c)
    cabi.densify_stats(g, r, 640.0, False, a, b, c)
    assert torch.equal(st[0], a) and torch.equal(st[1], b) and torch.equal(st[2], c)


def test_morton_reorder_at_refinement_renders_the_same_image():
    """Trainer(morton_order=True): the refinement output is permuted along a Z-order curve; the
    rendered image does not depend on the order of the Gaussians."""
    from opensplat_amd import train

    s = scenes.camera_scene(3000, 200, 150, K=4, seed=83, znear=1.0, zfar=100.0, degrees_to_use=1)
    raw = scenes.raw_parameters(s)
    cam = dict(viewmat=s.viewmat, projmat=s.projmat, fx=s.fx, fy=s.fy, cx=s.cx, cy=s.cy, W=s.W, H=s.H)
    T = train.Trainer(*raw, torch.device(DEV))
    img = T.render(cam, s.background, 1).clone()
    perm = train.morton_permutation(T.means)
    assert sorted(perm.tolist()) == list(range(3000))
    P = [t[perm.cpu().numpy()] if t is not None else None for t in raw]
    T2 = train.Trainer(*P, torch.device(DEV))
    img2 = T2.render(cam, s.background, 1)
    assert torch.equal(img, img2)
    # neighbours in memory are neighbours in space
    m = T2.means
    d_sorted = (m[1:] - m[:-1]).norm(dim=1).mean().item()
    d_given = (T.means[1:] - T.means[:-1]).norm(dim=1).mean().item()
    assert d_sorted < 0.3 * d_given


def test_rank_merged_statistics_take_the_single_camera_decisions():
    """with one camera per rank each rank's cotangent is scaled by 1/world, so the merged
    statistics must be judged against world x half_max_side (train.rank_merged_config).  Two cameras
    processed (a) by ONE rank in two iterations with unscaled gradients and (b) by TWO ranks with
    gradients scaled by 1/2 — rank 0 taking the "first iteration" branch, rank 1 starting from zero
    accumulators, merged by sum / sum / max as Trainer._refine does — must lead to the SAME plan."""
    import ctypes as C

    import torch

    from opensplat_amd import cabi
    from opensplat_amd.train import rank_merged_config

    prob = scenes.densify_problem(20000, K=4, seed=5)
    N, W, H = prob["N"], prob["width"], prob["height"]
    rs = np.random.RandomState(9)
    dev = "cuda"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    radii = [t((rs.randint(1, 40, N) * (rs.rand(N) < 0.7)).astype(np.int32)) for _ in range(2)]
    mag = 10.0 ** rs.uniform(-5.5, -2.0, (N, 1))
    v_xy = [t((mag * rs.standard_normal((N, 2)) / (0.5 * max(W, H))).astype(np.float32)) for _ in range(2)]
    for g, r in zip(v_xy, radii):
        g[r == 0] = 0.0                      # an invisible Gaussian receives no gradient

    def stats(grads, rads, firsts):
        acc = tuple(torch.zeros(N, device=dev) for _ in range(3))
        for g, r, first in zip(grads, rads, firsts):
            cabi.densify_stats(g, r, float(max(H, W)), first, *acc)
        return acc

    def plan(cfg, acc):
        need = cabi.lib().gs_densify_workspace_bytes(N)
        ws = torch.empty(need, device=dev, dtype=torch.uint8)
        counts = torch.zeros(8, dtype=torch.int32).pin_memory()
        # (kept in variables: a temporary handed to _p() is freed before the launch, and the second upload can be
        # given the first one's memory — the plan then read opacities where the scales should be, whenever the
        # allocator's state made it so: failed in the full suite of round 6, passed alone)
        log_scales, logits = t(prob["params"][1]), t(prob["params"][3]).reshape(-1)
        cabi._check(cabi.lib().gs_densify_plan(
            C.c_int(N), C.byref(cfg), cabi._p(acc[0]), cabi._p(acc[1]), cabi._p(acc[2]),
            cabi._p(log_scales), cabi._p(logits),
            C.c_void_p(counts.data_ptr()), cabi._p(ws), C.c_size_t(need), cabi._stream()), "plan")
        torch.cuda.synchronize()
        return dict(zip(cabi.COUNT_NAMES, [int(x) for x in counts]))

    one = plan(cabi.densify_config(W, H), stats(v_xy, radii, [True, False]))
    r0 = stats([0.5 * v_xy[0]], [radii[0]], [True])
    r1 = stats([0.5 * v_xy[1]], [radii[1]], [False])
    merged_acc = (r0[0] + r1[0], r0[1] + r1[1], torch.maximum(r0[2], r1[2]))
    merged = plan(rank_merged_config(cabi.densify_config(W, H), 2), merged_acc)
    assert merged == one, (merged, one)
    assert one["n_splits"] > 100 and one["n_dups"] > 100 and one["culled"] > 0
    # without the correction the scaled gradients would hardly ever pass the threshold
    unscaled = plan(cabi.densify_config(W, H), merged_acc)
    assert unscaled["n_splits"] + unscaled["n_dups"] < 0.9 * (one["n_splits"] + one["n_dups"])
