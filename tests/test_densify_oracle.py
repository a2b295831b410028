"""CPU: SURVEY.md §8 row f4 — the plain-C restatement of Model::afterTrain's tensor work
(oracle/densify_oracle.c) against the stored outputs of the same statements run under libtorch
with the reference's own quatToRotMat (tests/golden/densify.npz), live against oracle/_ref where
built, plus the C ABI's symbols and argument validation."""
import ctypes
import os
import re

import numpy as np
import pytest

from opensplat_amd import _build, cabi, scenes

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = np.load(os.path.join(HERE, "golden", "densify.npz"))
CASES = [(4, True, True, 1), (1, False, True, 2), (16, True, False, 3), (4, False, False, 4)]
N = 600


def samples_for(seed):
    return lambda n: np.random.RandomState(seed + 100).standard_normal((2 * n, 3)).astype(np.float32)


def check_set(r, ref, n_splits_kept_hint=None):
    """Copied rows are bit-equal (pure data movement); the split samples' means / scales go
    through exp, log and a 3x3 product: 1e-6 relative (libtorch's vectorised exp/log vs libm)."""
    for key in ("params", "exp_avg", "exp_avg_sq"):
        for i, (a, b) in enumerate(zip(r[key], ref[key])):
            assert a.shape == b.shape, (key, i)
            if a.size == 0:
                continue
            if key == "params" and i in (0, 1):
                assert np.abs(a - b).max() <= 1e-6 * max(np.abs(b).max(), 1.0), (key, i)
            else:
                assert np.array_equal(a, b), (key, i)


@pytest.mark.parametrize("case", CASES, ids=[f"K{c[0]}_screen{int(c[1])}_huge{int(c[2])}" for c in CASES])
def test_restated_refine_matches_golden(restated, case):
    K, cs, ch, seed = case
    prob = scenes.densify_problem(N, K, seed)
    r = restated.densify_refine(prob, 0.0002, 0.01, cs, 0.05, ch, samples_for(seed))
    tag = f"c{seed}"
    assert [r["n_splits"], r["n_dups"], r["new_n"], r["culled"]] == list(GOLD[f"{tag}_counts"])
    ref = {key: [GOLD[f"{tag}_{key}{i}"] for i in range(6)] for key in ("params", "exp_avg", "exp_avg_sq")}
    check_set(r, ref)
    # every branch is populated by the generator
    assert r["n_splits"] > 50 and r["n_dups"] > 50 and r["culled"] > r["n_splits"]


def test_restated_stats_match_golden(restated):
    rs = np.random.RandomState(5)
    g = np.zeros(N, np.float32); v = np.zeros(N, np.float32); m = np.zeros(N, np.float32)
    for it in range(4):
        grad = (rs.standard_normal((N, 2)) * 1e-4).astype(np.float32)
        rad = (rs.randint(0, 40, N) * (rs.rand(N) < 0.7)).astype(np.int32)
        grad[rad == 0] = 0
        restated.densify_stats(grad, rad, 480, 640, it == 0, g, v, m)
    assert np.array_equal(v, GOLD["stats_vis"]) and np.array_equal(m, GOLD["stats_m2d"])
    assert np.abs(g - GOLD["stats_gnorm"]).max() <= 2e-7 * np.abs(g).max()   # norm: 1 ulp
    assert v.min() >= 1.0   # invisible Gaussians still start at one (model.cpp:323)


def test_restated_vs_live_reference(restated, reference):
    prob = scenes.densify_problem(2000, 9, 17)
    smp = samples_for(17)
    a = reference.densify_refine(prob, 0.0003, 0.02, True, 0.08, True, smp)
    b = restated.densify_refine(prob, 0.0003, 0.02, True, 0.08, True, smp)
    assert (a["n_splits"], a["n_dups"], a["new_n"], a["culled"]) == \
           (b["n_splits"], b["n_dups"], b["new_n"], b["culled"])
    check_set(b, a)


def test_reset_opacity_restated(restated):
    x = np.linspace(-6, 6, 101, dtype=np.float32)
    y = restated.reset_opacity(x, 0.2)
    mx = np.float32(np.log(0.2 / 0.8))
    assert np.allclose(y, np.minimum(x, mx), atol=1e-7) and y.max() <= mx + 1e-6


def declared_functions():
    text = open(os.path.join(ROOT, "include", "gsplat_densify.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", text)))


def test_densify_symbols_exported():
    names = declared_functions()
    l = ctypes.CDLL(_build.HIP_LIB)
    assert not [n for n in names if not hasattr(l, n)]
    assert sorted(cabi.DENSIFY_SYMBOLS) == names


def test_densify_argument_validation_without_gpu():
    l = cabi.lib()
    null, one = ctypes.c_void_p(0), ctypes.c_void_p(256)
    f = ctypes.c_float
    assert l.gs_densify_workspace_bytes(0) == 0
    need = l.gs_densify_workspace_bytes(1000)
    assert need >= 1000 * 4 * 3 * 4
    assert l.gs_densify_stats(-1, one, one, f(640), 1, one, one, one, null) == -1
    assert l.gs_densify_stats(10, one, one, f(0), 1, one, one, one, null) == -1
    assert l.gs_densify_stats(0, null, null, f(640), 1, null, null, null, null) == 0
    cfg = cabi.densify_config(640, 480)
    assert l.gs_densify_plan(10, ctypes.byref(cfg), one, one, one, one, null, one, one,
                             ctypes.c_size_t(1 << 30), null) == -1
    assert l.gs_densify_plan(1000, ctypes.byref(cfg), one, one, one, one, one, one, one,
                             ctypes.c_size_t(need - 1), null) == -3
    sets = (cabi.GsGaussianSet * 3)()
    assert l.gs_densify_apply(1000, 0, 10, null, sets, sets, one, ctypes.c_size_t(need), null) == -1
    assert l.gs_densify_apply(1000, 4, 5000, null, sets, sets, one, ctypes.c_size_t(need), null) == -1
    assert l.gs_densify_apply(1000, 4, 0, null, sets, sets, one, ctypes.c_size_t(need), null) == 0
    assert l.gs_reset_opacity(10, f(1.5), one, null, null, null) == -1
    assert l.gs_reset_opacity(0, f(0.2), null, null, null, null) == 0
