"""CPU: SURVEY.md §8 row f2 — the restated loss / optimiser oracle (oracle/train_oracle.c) against
the stored vectors generated from the reference's own SSIM + libtorch (tests/golden/train_*.npz,
tests/golden/make_golden_train.py), live against oracle/_ref where that is built, and the host-side
entry points of the C ABI (gs_ssim_window, gs_sched_lr) that need no GPU."""
import ctypes
import hashlib
import os
import re

import numpy as np
import pytest

from opensplat_amd import _build, cabi, scenes

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LOSS = np.load(os.path.join(HERE, "golden", "train_loss.npz"))
ADAM = np.load(os.path.join(HERE, "golden", "train_adam.npz"))
LOSS_CASES = [("ragged", 75, 53, 11), ("even", 96, 64, 12)]
WEIGHTS = [0.2, 0.0, 1.0]
ADAM_N, ADAM_STEPS, ADAM_SEED, ADAM_LR = 4099, (1, 2, 6), 21, 0.005
SCHED = (0.00016, 0.0000016, 30000)
SCHED_STEPS = [0, 1, 2, 100, 7000, 15000, 29999, 30000, 31000]


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def test_window_is_the_references_asymmetric_one(restated):
    g, w2 = restated.ssim_window()
    assert np.array_equal(g, LOSS["window_1d"])          # bit for bit
    assert np.array_equal(w2, LOSS["window_2d"])
    assert g[0] < 1e-3 and g[10] > 0.29 and g[9] == g[10]  # mass at the far end (ssim.cpp:42)
    assert abs(float(g.sum()) - 1.0) < 1e-6


@pytest.mark.parametrize("case", LOSS_CASES, ids=[c[0] for c in LOSS_CASES])
@pytest.mark.parametrize("w", WEIGHTS)
def test_restated_main_loss_matches_golden(restated, case, w):
    name, W, H, seed = case
    rendered, gt = scenes.loss_images(W, H, seed)
    assert digest(rendered, gt) == str(LOSS[f"{name}_digest"]), "input generator drifted"
    loss, v = restated.main_loss(rendered, gt, w)
    ref_loss, ref_v = LOSS[f"{name}_w{w}_loss"], LOSS[f"{name}_w{w}_grad"]
    assert np.abs(loss - ref_loss).max() < 3e-7           # fp32 mean of ~1e4 terms
    scale = np.abs(ref_v).max()
    # conv2d summation order; sigma = E[xx] - mu^2 cancels in fp32 where rendered == gt exactly
    assert np.abs(v - ref_v).max() < 5e-5 * scale
    if w == 0.0:
        assert np.array_equal(v, ref_v)                   # pure sign / (3P)
        h4, w4 = H // 4, W // 4
        assert np.all(v[:h4, :w4] == 0.0)                 # sign(0) = 0


def assert_adam_param_close(p, ref, lr):
    """libtorch's vectorised CPU sqrt is not correctly rounded (see oracle/train_oracle.c): the
    parameter may be one ulp (of the parameter, or of the ~lr-sized update when the parameter is
    smaller than that) off in a fraction of a percent of the elements."""
    bad = p != ref
    assert bad.mean() < 5e-3
    assert np.all(np.abs(p - ref) <= 1.5 * np.spacing(np.maximum(np.abs(ref), np.float32(4 * lr))))


def test_restated_adam_moments_bit_identical_to_libtorch(restated):
    p0, grads = scenes.adam_problem(ADAM_N, max(ADAM_STEPS), ADAM_SEED)
    assert digest(p0, *grads) == str(ADAM["digest"]), "input generator drifted"
    p, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for s, g in enumerate(grads, start=1):
        restated.adam_step(p, g, m, v, ADAM_LR, s)
        if s in ADAM_STEPS:
            assert np.array_equal(m, ADAM[f"m{s}"])
            assert np.array_equal(v, ADAM[f"v{s}"])
            assert_adam_param_close(p, ADAM[f"p{s}"], ADAM_LR)


def test_scheduler_matches_golden(restated):
    got = np.array([restated.sched_lr(*SCHED, s) for s in SCHED_STEPS], np.float32)
    assert np.array_equal(got, ADAM["sched"])
    assert got[0] == np.float32(SCHED[0]) or abs(got[0] / SCHED[0] - 1) < 1e-6
    assert abs(got[-1] / SCHED[1] - 1) < 1e-6 and got[-1] == got[-2]   # clamped past max_steps


def test_restated_vs_live_reference(restated, reference):
    rendered, gt = scenes.loss_images(61, 47, 5, noise=0.3)
    for w in (0.2, 0.7):
        a, va = reference.main_loss(rendered, gt, w)
        b, vb = restated.main_loss(rendered, gt, w)
        assert np.abs(a - b).max() < 3e-7
        assert np.abs(va - vb).max() < 5e-5 * np.abs(va).max()
    p0, grads = scenes.adam_problem(1000, 4, 9)
    pr, mr, vr = reference.adam_steps(p0, grads, 0.05)
    p, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for s, g in enumerate(grads, start=1):
        restated.adam_step(p, g, m, v, 0.05, s)
    assert np.array_equal(m, mr) and np.array_equal(v, vr)
    assert_adam_param_close(p, pr, 0.05)


# ---- C ABI: symbols and the host-side entry points ---------------------------------------------

def declared_functions():
    text = open(os.path.join(ROOT, "include", "gsplat_train.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", text)))


def test_train_symbols_exported():
    names = declared_functions()
    l = ctypes.CDLL(_build.HIP_LIB)
    assert not [n for n in names if not hasattr(l, n)]
    assert sorted(cabi.TRAIN_SYMBOLS) == names, "cabi.TRAIN_SYMBOLS out of sync with the header"


def test_abi_window_and_scheduler_on_host():
    assert np.array_equal(np.array(cabi.ssim_window(), np.float32), LOSS["window_1d"])
    got = np.array([cabi.sched_lr(*SCHED, s) for s in SCHED_STEPS], np.float32)
    assert np.array_equal(got, ADAM["sched"])


def test_train_argument_validation_without_gpu():
    l = cabi.lib()
    null, one = ctypes.c_void_p(0), ctypes.c_void_p(256)
    assert l.gs_loss_workspace_bytes(0, 10) == 0
    need = l.gs_loss_workspace_bytes(64, 48)
    assert need >= 9 * 64 * 48 * 4
    f = ctypes.c_float
    assert l.gs_main_loss(64, 48, null, one, f(0.2), f(1.0), one, one, one, ctypes.c_size_t(need), null) == -1
    assert l.gs_main_loss(64, 48, one, one, f(0.2), f(1.0), one, one, one, ctypes.c_size_t(need - 1), null) == -3
    assert l.gs_main_loss(70000, 48, one, one, f(0.2), f(1.0), one, one, one, ctypes.c_size_t(1 << 40), null) == -2
    grp = (cabi.GsAdamGroup * 1)()
    d = ctypes.c_double
    assert l.gs_adam_step(9, grp, ctypes.c_int64(1), d(0.9), d(0.999), d(1e-8), null) == -1
    assert l.gs_adam_step(1, grp, ctypes.c_int64(0), d(0.9), d(0.999), d(1e-8), null) == -1
    grp[0].n = 5  # null pointers with n > 0
    assert l.gs_adam_step(1, grp, ctypes.c_int64(1), d(0.9), d(0.999), d(1e-8), null) == -1
    grp[0].n = 0  # nothing to do: fine without touching the device
    assert l.gs_adam_step(1, grp, ctypes.c_int64(1), d(0.9), d(0.999), d(1e-8), null) == 0
    assert l.gs_adam_step(0, null, ctypes.c_int64(1), d(0.9), d(0.999), d(1e-8), null) == 0
