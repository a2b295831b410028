"""-m gpu: the multi-rank path THROUGH THE REAL PIPELINE.  Two ranks (gloo, sharing the one GPU of
the test box) render two different C4-style cameras over the same Gaussians with opensplat_amd.pipeline.HotPath —
gs_gaussian_forward ... gs_gaussian_backward writing into the flat GradBuffer, then the gradient
exchange (flat all-reduce, or the factored exchange: geometry all-reduce + colour-cotangent
all-gather + local SH backward over all cameras) — and the exchanged buffer must equal the sum of
single-rank runs of the same cameras.  Also: `python bench.py --gpus 2` starts its own ranks and reports n_gpus = 2."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**kw):
    env = dict(os.environ, **kw)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("mode,cpr", [("flat", 1), ("factored", 1), ("factored", 2)])
def test_two_ranks_allreduced_gradients_equal_the_sum_of_single_rank_runs(tmp_path, mode, cpr):
    import socket

    import torch

    from opensplat_amd.pipeline import HotPath
    from tests.dist_pipeline_worker import small_c4

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    prefix = str(tmp_path / "flat")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tests", "dist_pipeline_worker.py"), prefix],
                       env=_clean_env(GSPLAT_DIST_BACKEND="gloo", GSPLAT_TEST_EXCHANGE=mode,
                                      GSPLAT_TEST_CPR=str(cpr)),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got0, got1 = np.load(prefix + "_rank0.npy"), np.load(prefix + "_rank1.npy")
    assert np.array_equal(got0, got1), "ranks disagree after the exchange"
    # single-rank runs of the same cameras in THIS process (world = 1: no exchange)
    dev = torch.device("cuda", 0)
    flats = []
    for cam in range(2 * cpr):
        pipe = HotPath(small_c4(cam), dev, 0)
        pipe.step()
        pipe.step()
        torch.cuda.synchronize()
        flats.append(pipe.grads.flat.cpu().numpy().astype(np.float64))
    want = sum(flats)
    assert np.abs(want).max() > 0
    # every block of the flat buffer (the six gradient tensors), relative to its own magnitude:
    # the compositing backward sums with atomics, so two runs differ by summation order (2e-5)
    gb = pipe.grads
    o = 0
    for name in ("v_rest", "v_dc", "v_means", "v_scales", "v_quats", "v_opacity"):
        n = gb.views[name].numel()
        a, b = got0[o:o + n], want[o:o + n]
        assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max(), name
        o += n
    assert o == got0.size


def test_bench_gpus_2_starts_two_ranks_and_reports_them():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--gaussians", "100000", "--cameras-per-rank", "2"],
                       env=_clean_env(GSPLAT_DIST_BACKEND="gloo"), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1]
    assert line["n_gpus"] == 2 and line["config"]["cameras_per_rank_per_step"] == 2
    # default exchange at 2 ranks x 2 cameras: factored (geometry all-reduced, colours all-gathered)
    assert line["exchange"]["mode"] == "factored"
    assert line["grad_bytes_allreduced"] == 100000 * 11 * 4
    assert line["exchange"]["allgather_message_bytes"] == (2 * 100000 * 3 + 8) * 4
    assert line["value"] > 0 and line["stage_ms"]["allreduce"] >= 0


def test_bench_flat_exchange_is_still_selectable():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--gaussians", "50000", "--exchange", "flat"],
                       env=_clean_env(GSPLAT_DIST_BACKEND="gloo"), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1]
    assert line["exchange"]["mode"] == "flat"
    assert line["grad_bytes_allreduced"] == 50000 * (3 * 16 + 11) * 4


def test_gradient_accumulation_over_a_rank_s_camera_batch():
    """GS_FLAG_ACCUMULATE_GRADS: rendering camera 1 after camera 0 with accumulation leaves the sum
    of both cameras' gradients in the flat buffer (what ONE all-reduce then exchanges)."""
    import torch

    from opensplat_amd.pipeline import HotPath
    from opensplat_amd import scenes
    from tests.dist_pipeline_worker import small_c4

    dev = torch.device("cuda", 0)
    s0 = small_c4(0)
    pipe = HotPath(s0, dev, 0)
    cams = [scenes.yaw_camera(s0.W, s0.H, y, 1.0, 100.0) for y in scenes.C4_YAWS[:2]]
    singles = []
    for c in cams:
        pipe.set_camera(*c)
        pipe.step()
        pipe.step()
        torch.cuda.synchronize()
        singles.append(pipe.grads.flat.cpu().numpy().astype(np.float64))
    assert np.abs(singles[0] - singles[1]).max() > 0
    pipe.set_camera(*cams[0])
    pipe.step(accumulate=False, exchange=False)
    pipe.set_camera(*cams[1])
    pipe.step(accumulate=True, exchange=True)
    torch.cuda.synchronize()
    got = pipe.grads.flat.cpu().numpy()
    want = singles[0] + singles[1]
    o = 0
    for name in ("v_rest", "v_dc", "v_means", "v_scales", "v_quats", "v_opacity"):
        n = pipe.grads.views[name].numel()
        assert np.abs(got[o:o + n] - want[o:o + n]).max() <= 2e-5 * np.abs(want[o:o + n]).max(), name
        o += n


def test_two_rank_training_flat_and_factored_exchange_agree(tmp_path):
    """opensplat_amd.train.Trainer with one camera per rank: the exchanged gradients of the factored
    exchange (default) equal those of the flat bucketed all-reduce, both ranks end every iteration
    with identical parameters."""
    import socket

    out = {}
    for mode in ("flat", "factored"):
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        prefix = str(tmp_path / mode)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", str(port),
                            os.path.join(ROOT, "tests", "dist_train_worker.py"), prefix],
                           env=_clean_env(GSPLAT_DIST_BACKEND="gloo", GSPLAT_TEST_EXCHANGE=mode),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        g0, g1 = np.load(prefix + "_grads_rank0.npy"), np.load(prefix + "_grads_rank1.npy")
        p0, p1 = np.load(prefix + "_params_rank0.npy"), np.load(prefix + "_params_rank1.npy")
        assert np.array_equal(g0, g1) and np.array_equal(p0, p1), mode
        assert np.isfinite(p0).all() and np.abs(g0).max() > 0
        out[mode] = g0.astype(np.float64)
    N, K = 6000, 16
    o = 0
    for name, n in (("v_rest", N * (K - 1) * 3), ("v_dc", N * 3), ("v_means", N * 3), ("v_scales", N * 3),
                    ("v_quats", N * 4), ("v_opacity", N)):
        a, b = out["factored"][o:o + n], out["flat"][o:o + n]
        assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max(), name
        o += n
    assert o == out["flat"].size
