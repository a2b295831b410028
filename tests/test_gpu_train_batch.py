"""-m gpu: Trainer.train_step_batch — ONE optimiser step over a batch of cameras on this rank with two of them in
flight (opensplat_amd/train.py, opensplat_amd/pipeline.py: two_in_flight).  Generalises the per-image body of
opensplat.cpp:151-170; the reference has one camera per step, so the checks are against this library's own
single-camera step (oracle-checked in tests/test_gpu_train.py) and the serial camera loop:

  * deterministic mode: gradients, per-camera losses and densification statistics of the two-in-flight batch equal
    the serial loop's bit for bit, for two, three and five cameras (lanes re-used);
  * the batch gradient equals the sum of the single-camera gradients (render + loss scaled by 1 / c + backward) to the
    rounding of the accumulation;
  * one batch = one optimiser step: step count + 1, parameters equal to Adam on the batch gradient;
  * two ranks x two cameras (gloo on one GPU), flat and factored exchange: both ranks end with identical gradients
    and parameters, and the exchanges agree with each other.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from opensplat_amd import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(ncam, deterministic=True, N=12000, W=320, H=192):
    import torch

    from opensplat_amd.train import Trainer

    dev = torch.device("cuda:0")
    s = scenes.camera_scene(N, W, H, K=16, seed=7, sigma_px=(0.6, 5.0), znear=1.0, zfar=100.0)
    T = Trainer(*scenes.raw_parameters(s), device=dev, deterministic=deterministic, sh_degree_interval=1)
    cams, gts = [], []
    for j, yaw in enumerate((-8.0, -3.0, 2.0, 6.0, 10.0)[:ncam]):
        vm, pm = scenes.yaw_camera(W, H, yaw, 1.0, 100.0)
        cams.append(dict(viewmat=vm, projmat=pm, fx=s.fx, fy=s.fy, cx=s.cx, cy=s.cy, W=W, H=H))
        gts.append(torch.from_numpy(np.random.RandomState(40 + j).uniform(0, 1, (H, W, 3)).astype(np.float32)).to(dev))
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    return T, cams, gts, bg


@pytest.mark.parametrize("ncam", [2, 3, 5])
def test_batch_with_two_cameras_in_flight_equals_the_serial_loop_bit_for_bit(ncam):
    import torch

    T, cams, gts, bg = _setup(ncam)

    def run(serial):
        T._stats = None
        losses = T.train_step_batch(cams, gts, bg, 3, step=7, step_optimizer=False, serial=serial)
        torch.cuda.synchronize()
        return T.grads.flat.clone(), losses.clone(), [t.clone() for t in T._stats]

    g_ref, l_ref, st_ref = run(True)
    assert torch.isfinite(g_ref).all() and float(g_ref.abs().max()) > 0 and float(l_ref[:, 0].min()) > 0
    for _ in range(3):
        g, l, st = run(False)
        assert torch.equal(g, g_ref) and torch.equal(l, l_ref)
        assert all(torch.equal(a, b) for a, b in zip(st, st_ref))
    assert T.step_count == 0 and T._batch_stats_step == 7
    # visCounts: rank 0's first camera counts every Gaussian once, the others the visible ones (model.cpp:321-326)
    assert float(st_ref[1].min()) >= 1 and float(st_ref[1].max()) <= ncam


def test_batch_gradient_is_the_sum_of_the_single_camera_gradients():
    import torch

    from opensplat_amd import cabi

    T, cams, gts, bg = _setup(3)
    T.train_step_batch(cams, gts, bg, 3, step_optimizer=False)
    torch.cuda.synchronize()
    got = T.grads.flat.double().cpu().numpy()
    want = np.zeros_like(got)
    for cam, gt in zip(cams, gts):
        rgb = T.render(cam, bg, 3)
        _, v_rgb = cabi.main_loss(rgb, gt, T.ssim_weight, 1.0 / 3, True, out=T.loss_out, workspace=T.loss_ws)
        T.backward(v_rgb)
        torch.cuda.synchronize()
        want += T.grads.flat.double().cpu().numpy()
    o = 0
    for name in ("v_rest", "v_dc", "v_means", "v_scales", "v_quats", "v_opacity"):
        n = T.grads.views[name].numel()
        assert np.abs(got[o:o + n] - want[o:o + n]).max() <= 2e-6 * np.abs(want[o:o + n]).max(), name
        o += n


def test_one_batch_is_one_optimiser_step():
    import torch

    from opensplat_amd import cabi

    T, cams, gts, bg = _setup(2)
    p0 = T.params.flat.clone()
    losses = T.train_step_batch(cams, gts, bg, 3)
    torch.cuda.synchronize()
    assert T.step_count == 1 and tuple(losses.shape) == (2, 3)
    g = T.grads.flat.clone()
    # Adam's first step (bias-corrected moments = g, g^2): every parameter moves by lr g / (|g| + eps)
    moved = (T.params.flat - p0)
    lr = torch.zeros_like(moved)
    o = 0
    for v, n in (("v_rest", "features_rest"), ("v_dc", "features_dc"), ("v_means", "means"), ("v_scales", "scales"),
                 ("v_quats", "quats"), ("v_opacity", "opacities")):
        cnt = T.params.views[v].numel()
        lr[o:o + cnt] = T.LR[n]
        o += cnt
    big = g.abs() > 1e-6
    want = -lr * g / (g.abs() + 1e-8)
    # (fp32: the update is rounded into the parameter, |p| up to a few units against steps of 1e-4)
    assert bool(((moved - want).abs()[big] <= 2e-3 * lr[big] + 2.5e-7 * p0.abs()[big].clamp_min(1.0)).all())
    assert int(big.sum()) > 1000
    # a second batch continues from there (lanes and buffers re-used), and a plain train_step still works afterwards
    T.train_step_batch(cams, gts, bg, 3)
    T.train_step(cams[0], gts[0], bg, 3)
    torch.cuda.synchronize()
    assert T.step_count == 3 and torch.isfinite(T.params.flat).all()


def test_two_ranks_times_two_cameras_flat_and_factored(tmp_path):
    out = {}
    for mode in ("flat", "factored"):
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        prefix = str(tmp_path / mode)
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        env.update(GSPLAT_DIST_BACKEND="gloo", GSPLAT_TEST_EXCHANGE=mode, GSPLAT_TEST_CPR="2")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", str(port),
                            os.path.join(ROOT, "tests", "dist_train_worker.py"), prefix],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        g0, g1 = np.load(prefix + "_grads_rank0.npy"), np.load(prefix + "_grads_rank1.npy")
        p0, p1 = np.load(prefix + "_params_rank0.npy"), np.load(prefix + "_params_rank1.npy")
        assert np.array_equal(g0, g1) and np.array_equal(p0, p1), mode
        assert np.isfinite(p0).all() and np.abs(g0).max() > 0
        out[mode] = g0.astype(np.float64)
    a, b = out["factored"], out["flat"]
    assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max()
