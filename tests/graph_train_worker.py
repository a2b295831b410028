"""Worker of tests/test_gpu_train_graph.py (not a test module): one case of captured (HIP graph) training per
process — a GPU memory fault aborts the process that hit it, and must not take the test session with it.

    python tests/graph_train_worker.py <case>      prints "OK <case> <graph_stats>" and exits 0
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.test_gpu_train_blocks import _capture, _params, _run, _same  # noqa: E402


def case_bit_for_bit():
    from opensplat_amd import train

    dev, cams, images, init, bg = _capture()
    kw = dict(max_steps=200, deterministic=True, refine_every=6, warmup_length=5, reset_alpha_every=4,
              sh_degree_interval=9)
    order = [0, 1, 2, 3, 4, 2, 0]
    deg = lambda s: min(s // 9, 1)
    ref = train.Trainer(*init, dev, segmented=False, **kw)   # (the captured iteration runs the one-pass backward)
    got = train.Trainer(*init, dev, experimental_graph=True, **kw)
    n_ref = _run(ref, cams, images, bg, 40, order, deg=deg)
    n_got = _run(got, cams, images, bg, 40, order, deg=deg)
    assert n_ref == n_got and len(n_ref) >= 2, (n_ref, n_got)          # refinements happened, identically
    assert ref.N == got.N and ref.step_count == got.step_count == 40
    assert _same(_params(ref), _params(got))
    st = got.graph_stats
    assert st["replays"] > 20 and st["captures"] >= 3, st              # (N changed, the SH degree changed)
    assert abs(ref.means_lr - got.means_lr) == 0.0


def case_overflow():
    import torch

    from opensplat_amd import train
    from train_synthetic_inputs import make_camera

    dev, cams, images, init, bg = _capture()
    W, H = cams[0]["W"], cams[0]["H"]
    # a camera five times closer to the blob: many more (tile, Gaussian) pairs than the ring cameras
    near = make_camera((0.55, 0.05, 0.3), W, H)
    kw = dict(max_steps=100, deterministic=True)
    ref = train.Trainer(*init, dev, segmented=False, **kw)   # (the captured iteration runs the one-pass backward)
    got = train.Trainer(*init, dev, experimental_graph=True, **kw)
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


def case_resolution():
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
    ref = train.Trainer(*init, dev, segmented=False, **kw)   # (the captured iteration runs the one-pass backward)
    got = train.Trainer(*init, dev, experimental_graph=True, **kw)
    for T in (ref, got):
        for s in range(10):
            ci = s % len(cams)
            if s < 5:
                T.train_step(reduced(cams[ci]), half[ci], bg, 0)
            else:
                T.train_step(cams[ci], images[ci], bg, 1)
    assert _same(_params(ref), _params(got))
    assert got.graph_stats["captures"] >= 2


def case_eager_renders():
    """render() for evaluation — eager launches on the caller's stream, same image size, hence the same
    buffers — between replays of a live graph: what faulted when replays shared that stream."""
    import torch

    from opensplat_amd import train

    dev, cams, images, init, bg = _capture(K=16)
    kw = dict(max_steps=400, deterministic=True)
    ref = train.Trainer(*init, dev, segmented=False, **kw)   # (the captured iteration runs the one-pass backward)
    got = train.Trainer(*init, dev, experimental_graph=True, **kw)
    for T in (ref, got):
        pcs = [T.prepare_camera(c) for c in cams] if T.graph else cams
        for s in range(1, 241):
            ci = s % len(cams)
            T.train_step(pcs[ci], images[ci], bg, min(s // 30, 3))
            T.after_train(s)
            if s % 20 == 0:
                for c in cams[:3]:
                    T.render(c, bg, 3)
    torch.cuda.synchronize()
    assert _same(_params(ref), _params(got))
    assert got.graph_stats["replays"] > 200


CASES = {"bit_for_bit": case_bit_for_bit, "overflow": case_overflow, "resolution": case_resolution,
         "eager_renders": case_eager_renders}

if __name__ == "__main__":
    name = sys.argv[1]
    CASES[name]()
    import torch

    torch.cuda.synchronize()
    print("OK", name, flush=True)
