"""-m gpu: Trainer(graph=True) — the training iteration replayed as ONE captured HIP graph (VERDICT r03
"next" 7; the loop being replaced: opensplat.cpp:151-170).

With deterministic=True the compositing backward sums in fixed point, Adam is bit-exact by construction
(gs_adam_step_scheduled reads the very scalars gs_adam_step computes), so a captured run must leave the SAME
BITS in every parameter as the launch-by-launch run: over camera changes, an id-list overflow inside a
replayed graph (the guarded Adam step must not move anything; the iteration is repeated), refinements that
replace every buffer, an SH-degree change and a resolution change.  Also: gs_adam_step_scheduled against
gs_adam_step, and its guard.
"""
import math
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def _capture(n_gt=4000, n_init=1500, W=160, H=96, n_cams=5, K=4, radius=3.5):
    import torch

    from opensplat_amd import train
    from train_synthetic_inputs import ground_truth, make_camera, sfm_like_init

    rs = np.random.RandomState(0)
    dev = torch.device("cuda", 0)
    cams = [make_camera((radius * math.cos(a), 0.4 * math.sin(2 * a), radius * math.sin(a)), W, H)
            for a in np.linspace(0.0, 2.0 * math.pi, n_cams, endpoint=False)]
    gt = ground_truth(n_gt, K, rs)
    G = train.Trainer(*gt, dev)
    bg = np.zeros(3, np.float32)
    images = [G.render(c, bg, 1).clone() for c in cams]
    init = sfm_like_init(gt, n_init, K, rs)
    return dev, cams, images, init, bg


def _params(T):
    import torch

    torch.cuda.synchronize()
    return {k: getattr(T, k).clone() for k in ("means", "log_scales", "quats", "opacity_logits", "features_dc",
                                                "features_rest")}


def _same(a, b):
    import torch

    return all(torch.equal(a[k], b[k]) for k in a)


def _run(T, cams, images, bg, steps, order, after_train=True, deg=lambda s: 1):
    counts = []
    for step in range(1, steps + 1):
        ci = order[(step - 1) % len(order)]
        T.train_step(cams[ci], images[ci], bg, deg(step))
        if after_train:
            c = T.after_train(step)
            if c is not None:
                counts.append((step, c["new_n"]))
    return counts


def test_scheduled_adam_moves_the_same_bits_and_obeys_its_guard():
    import torch

    from opensplat_amd import cabi

    g = torch.Generator(device="cuda").manual_seed(3)
    sizes = [1000, 37, 4096]
    lrs = [0.00016, 0.005, 0.05]

    def fresh():
        gg = torch.Generator(device="cuda").manual_seed(4)
        return [[torch.randn(n, device="cuda", generator=gg) for _ in range(2)] +
                [torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")] for n in sizes]
    A, B = fresh(), fresh()
    first = 7
    rows = torch.from_numpy(cabi.adam_schedule_rows([lrs] * 6, first)).cuda()
    idx = torch.zeros(1, dtype=torch.int32, device="cuda")
    guard = torch.tensor([10], dtype=torch.int32, device="cuda")
    for r in range(4):
        for t in A + B:
            t[1].copy_(torch.randn(t[1].shape, device="cuda", generator=g))
        for a, b in zip(A, B):
            b[1].copy_(a[1])
        cabi.adam_step([(p, gr, m, v, lr) for (p, gr, m, v), lr in zip(A, lrs)], first + r)
        grp = [(p, gr, m, v, 0.0) for (p, gr, m, v) in B]
        cabi.adam_step_scheduled(grp, rows, idx, guard, 10)      # *guard == guard_max: the step counts
        cabi.adam_advance(idx, guard, 10)
        assert int(idx.item()) == r + 1
        for a, b in zip(A, B):
            assert all(torch.equal(x, y) for x, y in zip(a, b)), r
    # guard exceeded: nothing moves, the row index stays
    before = [[x.clone() for x in b] for b in B]
    guard.fill_(11)
    cabi.adam_step_scheduled([(p, gr, m, v, 0.0) for (p, gr, m, v) in B], rows, idx, guard, 10)
    cabi.adam_advance(idx, guard, 10)
    assert int(idx.item()) == 4
    assert all(torch.equal(x, y) for b, c in zip(B, before) for x, y in zip(b, c))
    # a row index outside the table: nothing moves either
    guard.fill_(0)
    idx.fill_(6)
    cabi.adam_step_scheduled([(p, gr, m, v, 0.0) for (p, gr, m, v) in B], rows, idx, guard, 10)
    assert all(torch.equal(x, y) for b, c in zip(B, before) for x, y in zip(b, c))


def test_captured_training_equals_launch_by_launch_training_bit_for_bit():
    from opensplat_amd import train

    dev, cams, images, init, bg = _capture()
    kw = dict(max_steps=200, deterministic=True, refine_every=6, warmup_length=5, reset_alpha_every=4,
              sh_degree_interval=9)
    order = [0, 1, 2, 3, 4, 2, 0]
    deg = lambda s: min(s // 9, 1)
    ref = train.Trainer(*init, dev, **kw)
    got = train.Trainer(*init, dev, graph=True, **kw)
    n_ref = _run(ref, cams, images, bg, 40, order, deg=deg)
    n_got = _run(got, cams, images, bg, 40, order, deg=deg)
    assert n_ref == n_got and len(n_ref) >= 2, (n_ref, n_got)          # refinements happened, identically
    assert ref.N == got.N and ref.step_count == got.step_count == 40
    assert _same(_params(ref), _params(got))
    st = got.graph_stats
    assert st["replays"] > 20 and st["captures"] >= 3, st              # (N changed, the SH degree changed)
    assert abs(ref.means_lr - got.means_lr) == 0.0


def test_overflow_inside_a_replayed_graph_changes_nothing_and_is_repeated():
    import torch

    from opensplat_amd import train
    from train_synthetic_inputs import make_camera

    dev, cams, images, init, bg = _capture()
    W, H = cams[0]["W"], cams[0]["H"]
    # a camera five times closer to the blob: many more (tile, Gaussian) pairs than the ring cameras
    near = make_camera((0.55, 0.05, 0.3), W, H)
    kw = dict(max_steps=100, deterministic=True)
    ref = train.Trainer(*init, dev, **kw)
    got = train.Trainer(*init, dev, graph=True, **kw)
    near_img = ref.render(near, bg, 1).clone() * 0.5
    seq = [(cams[0], images[0]), (cams[0], images[0]), (cams[1], images[1]), (near, near_img),
           (cams[2], images[2]), (near, near_img), (cams[0], images[0])]
    for T in (ref, got):
        for c, img in seq:
            T.train_step(c, img, bg, 1)
    st = got.graph_stats
    assert st["overflows"] >= 1, st         # the near camera ran into the capacity the ring cameras had set
    assert st["replays"] >= 2, st
    assert got.step_count == ref.step_count == len(seq)
    assert _same(_params(ref), _params(got))
    torch.cuda.synchronize()


def test_resolution_change_recaptures():
    from opensplat_amd import train

    dev, cams, images, init, bg = _capture()
    import torch

    def reduced(cam):
        c = dict(cam)
        c.update(fx=cam["fx"] / 2, fy=cam["fy"] / 2, cx=cam["cx"] / 2, cy=cam["cy"] / 2, W=cam["W"] // 2,
                 H=cam["H"] // 2)
        return c
    half = [torch.nn.functional.avg_pool2d(im.permute(2, 0, 1)[None], 2)[0].permute(1, 2, 0).contiguous()
            for im in images]
    kw = dict(max_steps=100, deterministic=True)
    ref = train.Trainer(*init, dev, **kw)
    got = train.Trainer(*init, dev, graph=True, **kw)
    for T in (ref, got):
        for s in range(10):
            ci = s % len(cams)
            if s < 5:
                T.train_step(reduced(cams[ci]), half[ci], bg, 0)
            else:
                T.train_step(cams[ci], images[ci], bg, 1)
    assert _same(_params(ref), _params(got))
    assert got.graph_stats["captures"] >= 2
