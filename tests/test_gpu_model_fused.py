"""-m gpu: OpenSplat's OWN `Model` training on the fused MI355X operators (a C++ caller at the
opensplat.cpp:151-170 level; Model-level integration).

oracle/_ref/model_fused_shim = the reference's model.cpp patched by `integration/apply_hip_native.py
--fused` (five guarded call sites into opensplat_amd/csrc/model_fused.inl) + tests/integration/
model_fused_tu.cpp as the training loop; built by build() where /root/reference exists, travels to the GPU
box.  The same binary runs the patched Model on the GPU (SplatRender, MainLoss, one-launch Adam, device
densification) and on the CPU (the reference's original statements on gsplat-cpu: the patch leaves that
device alone).  Compared here: the two devices with each other and with the Python Trainer, from the same
starting point, iteration by iteration.

The reference's CPU chain composites in the order of its as-read keys; the comparison
scene therefore keeps the Gaussians apart (no pixel sees two of them), where the order cannot matter."""
import os
import struct
import subprocess

import numpy as np
import pytest

from opensplat_amd import colmap, scenes
from tests.util import np_, to_dev

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "oracle", "_ref", "model_fused_shim")
needs_shim = pytest.mark.skipif(not os.path.exists(SHIM), reason="oracle/_ref/model_fused_shim not built "
                                "(needs /root/reference at build time)")


def write_arrays(path, arrays):
    with open(path, "wb") as f:
        for name, a in arrays.items():
            a = np.ascontiguousarray(a, np.float32)
            f.write(struct.pack("<I", len(name)) + name.encode() + struct.pack("<I", a.ndim))
            f.write(struct.pack("<%dq" % a.ndim, *a.shape))
            f.write(a.tobytes())


def read_arrays(path):
    out, blob, pos = {}, open(path, "rb").read(), 0
    while pos < len(blob):
        (nl,) = struct.unpack_from("<I", blob, pos); pos += 4
        name = blob[pos:pos + nl].decode(); pos += nl
        (nd,) = struct.unpack_from("<I", blob, pos); pos += 4
        dims = struct.unpack_from("<%dq" % nd, blob, pos); pos += 8 * nd
        n = int(np.prod(dims)) if nd else 1
        out[name] = np.frombuffer(blob, np.float32, n, pos).reshape(dims).copy(); pos += 4 * n
    return out


def separated_scene(W=160, H=112, K=4, seed=3):
    """Gaussians on a 10-pixel grid, sigma ~ 1.1 px: every pixel is reached by at most one of them."""
    rs = np.random.RandomState(seed)
    gx, gy = np.meshgrid(np.arange(8, W - 4, 10.0), np.arange(8, H - 4, 10.0))
    px, py = gx.ravel() + rs.uniform(-0.4, 0.4, gx.size), gy.ravel() + rs.uniform(-0.4, 0.4, gx.size)
    n = px.size
    fx = fy = 0.5 * W
    z = rs.uniform(3.0, 6.0, n)
    means = np.stack([(px - W / 2) * z / fx, (py - H / 2) * z / fy, z], -1).astype(np.float32)
    log_scales = np.log(np.stack([1.1 * z / fx] * 3, -1) * rs.uniform(0.8, 1.0, (n, 3))).astype(np.float32)
    quats = rs.standard_normal((n, 4)).astype(np.float32)
    logits = rs.uniform(0.0, 2.5, (n, 1)).astype(np.float32)
    dc = rs.uniform(-1.0, 1.5, (n, 3)).astype(np.float32)
    rest = (0.1 * rs.standard_normal((n, K - 1, 3))).astype(np.float32)
    return [means, log_scales, quats, logits, dc, rest], fx, fy


def cam_to_world(yaw_deg):
    """camToWorld (OpenGL axes) whose view matrix (model.cpp:93-106) is a rotation about +y at the origin."""
    a = np.radians(yaw_deg)
    Ry = np.array([[np.cos(a), 0, -np.sin(a)], [0, 1, 0], [np.sin(a), 0, np.cos(a)]], np.float32)
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, :3] = Ry.T @ np.diag(np.array([1.0, -1.0, -1.0], np.float32))
    return c2w


def run_shim(tmp_path, case, device, tag, shim=SHIM):
    cpath, opath = str(tmp_path / ("case_%s.bin" % tag)), str(tmp_path / ("out_%s_%s.bin" % (tag, device)))
    write_arrays(cpath, case)
    r = subprocess.run([shim, cpath, opath, "--device", device], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    return read_arrays(opath), r.stdout


def make_case(params, fx, fy, W, H, yaws, iters, shDegree, **cfg):
    c = dict(numCameras=len(yaws), numDownscales=0, resolutionSchedule=3000, shDegreeInterval=2,
             refineEvery=100, warmupLength=500, resetAlphaEvery=30, stopScreenSizeAt=4000, maxSteps=30000,
             firstStep=1, densifyGradThresh=0.0002, densifySizeThresh=0.01, splitScreenSize=0.05, ssimWeight=0.2)
    c.update(cfg)
    cams = np.stack([np.concatenate([[fx, fy, W / 2.0, H / 2.0], cam_to_world(y).ravel()]) for y in yaws])
    gts = np.stack([scenes.loss_images(W, H, seed=11 + i)[1] for i in range(len(yaws))])
    case = {"cfg": np.array([c["numCameras"], c["numDownscales"], c["resolutionSchedule"], shDegree,
                             c["shDegreeInterval"], c["refineEvery"], c["warmupLength"], c["resetAlphaEvery"],
                             c["stopScreenSizeAt"], c["maxSteps"], iters, c["firstStep"], W, H], np.float32),
            "cfgf": np.array([c["densifyGradThresh"], c["densifySizeThresh"], c["splitScreenSize"], c["ssimWeight"]],
                             np.float32),
            "cams": cams.astype(np.float32), "gt": gts.astype(np.float32)}
    for i, p in enumerate(params):
        case["p%d" % i] = p
    return case, c, gts


def run_trainer(params, fx, fy, W, H, yaws, iters, gts, c, **kw):
    import torch

    from opensplat_amd import train

    dev = torch.device("cuda", 0)
    tr = train.Trainer(*params, dev, max_steps=c["maxSteps"], ssim_weight=c["ssimWeight"],
                       refine_every=c["refineEvery"], warmup_length=c["warmupLength"],
                       reset_alpha_every=c["resetAlphaEvery"], densify_grad_thresh=c["densifyGradThresh"],
                       densify_size_thresh=c["densifySizeThresh"], stop_screen_size_at=c["stopScreenSizeAt"],
                       split_screen_size=c["splitScreenSize"], num_cameras=c["numCameras"],
                       num_downscales=c["numDownscales"], resolution_schedule=c["resolutionSchedule"],
                       sh_degree_interval=c["shDegreeInterval"], **kw)
    cams = [colmap.render_camera(colmap.Camera(width=W, height=H, fx=fx, fy=fy, cx=W / 2.0, cy=H / 2.0,
                                               cam_to_world=cam_to_world(y))) for y in yaws]
    bg = np.array(scenes.BACKGROUND, np.float32)
    losses = []
    for it in range(iters):
        step = c["firstStep"] + it
        loss = tr.train_step(cams[it % len(yaws)], to_dev(gts[it % len(yaws)]), bg, tr.degrees_to_use(step))
        tr.after_train(step)
        losses.append(float(loss[0]))
    torch.cuda.synchronize()
    P = tr.params
    return dict(losses=np.array(losses, np.float32), p0=np_(P.v_means), p1=np_(P.v_scales), p2=np_(P.v_quats),
                p3=np_(P.v_opacity).reshape(-1, 1), p4=np_(P.v_dc), p5=np_(P.v_rest),
                m0=np_(tr.exp_avg.v_means), v0=np_(tr.exp_avg_sq.v_means), means_lr=tr.means_lr,
                m3=np_(tr.exp_avg.v_opacity).reshape(-1, 1), v3=np_(tr.exp_avg_sq.v_opacity).reshape(-1, 1),
                stats=None if tr._stats is None else [np_(t) for t in tr._stats], N=tr.N)


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


@needs_shim
def test_patched_reference_model_trains_alike_on_gpu_cpu_and_in_the_python_trainer(tmp_path):
    """Six iterations over two cameras, no refinement (warm-up): losses, parameters, Adam moments, the
    densification statistics and the scheduled learning rate."""
    W, H, iters, yaws = 160, 112, 6, [0.0, 2.0]
    params, fx, fy = separated_scene(W, H, K=4)
    case, c, gts = make_case(params, fx, fy, W, H, yaws, iters, shDegree=1)
    gpu, _ = run_shim(tmp_path, case, "gpu", "a")
    cpu, _ = run_shim(tmp_path, case, "cpu", "a")
    py = run_trainer(params, fx, fy, W, H, yaws, iters, gts, c)
    # GPU (fused operators) against the Python Trainer: the same kernels underneath
    assert np.abs(gpu["losses"] - py["losses"]).max() < 2e-6, (gpu["losses"], py["losses"])
    for k in ("p0", "p1", "p2", "p3", "p4", "p5", "m0", "v0"):
        assert rel(gpu[k], py[k].reshape(gpu[k].shape)) < 2e-5, k
    assert abs(float(gpu["means_lr"][0]) - py["means_lr"]) < 1e-10
    for a, b in zip((gpu["xysGradNorm"], gpu["visCounts"], gpu["max2DSize"]), py["stats"]):
        assert rel(a, b) < 1e-5
    # GPU against the reference's own CPU statements (gsplat-cpu, torch::optim::Adam, libtorch SSIM)
    assert np.abs(gpu["losses"] - cpu["losses"]).max() < 2e-5, (gpu["losses"], cpu["losses"])
    assert gpu["losses"][-1] < gpu["losses"][0]                       # and it trains
    for k in ("p0", "p1", "p2", "p3", "p4", "p5"):
        d = np.abs(gpu[k] - cpu[k]).max()
        moved = np.abs(cpu[k] - case[k]).max()
        assert d <= 0.02 * moved + 1e-7, (k, d, moved)               # Adam normalises: compare to the displacement
    assert rel(gpu["m0"], cpu["m0"]) < 2e-3 and rel(gpu["v0"], cpu["v0"]) < 4e-3
    assert np.array_equal(gpu["visCounts"], cpu["visCounts"])
    assert rel(gpu["xysGradNorm"], cpu["xysGradNorm"]) < 2e-3
    assert rel(gpu["max2DSize"], cpu["max2DSize"]) < 1e-6
    assert np.abs(gpu["rgb"] - cpu["rgb"]).max() < 1e-4


@needs_shim
def test_patched_reference_model_refines_on_the_device(tmp_path):
    """A refinement inside the loop (warm-up 2, refine every 3): the patched Model's afterTrain runs the
    device densification, re-registers the six parameters with their moments in the torch::optim::Adam
    objects and keeps training; the Python Trainer takes the same split / duplicate / cull decisions."""
    W, H, iters, yaws = 160, 112, 9, [0.0]
    params, fx, fy = separated_scene(W, H, K=4, seed=8)
    params[3][::5] = -4.0                       # faint ones: culled (sigmoid < 0.1)
    params[1][::7] += 1.5                       # large ones: split candidates
    case, c, gts = make_case(params, fx, fy, W, H, yaws, iters, shDegree=1, refineEvery=3, warmupLength=2,
                             resetAlphaEvery=30, numCameras=0, densifyGradThresh=1e-7)
    gpu, out = run_shim(tmp_path, case, "gpu", "b")
    n0 = params[0].shape[0]
    assert gpu["counts"][1] == n0 and gpu["counts"][-1] != n0, gpu["counts"]     # refined at steps 6 and 9
    assert "Added" in out
    N = int(gpu["counts"][-1])
    for i, last in enumerate((3, 3, 4, 1, 3)):
        assert gpu["p%d" % i].shape == (N, last) and np.isfinite(gpu["p%d" % i]).all()
        assert gpu["m%d" % i].shape == (N, last) and gpu["v%d" % i].shape == (N, last)
    assert gpu["p5"].shape == (N, 3, 3)
    assert np.isfinite(gpu["losses"]).all()
    assert "xysGradNorm" not in gpu                                   # cleared by the refinement at step 9
    py = run_trainer(params, fx, fy, W, H, yaws, iters, gts, c)
    # the same counts after every refinement (decisions do not depend on the split samples)
    assert py["N"] == N
    assert np.abs(gpu["losses"][:6] - py["losses"][:6]).max() < 2e-6   # identical until the first split samples


SHIM_REFALPHA = SHIM + "_refalpha"


@needs_shim
@pytest.mark.skipif(not os.path.exists(SHIM_REFALPHA), reason="oracle/_ref/model_fused_shim_refalpha not built")
def test_patched_reference_model_alpha_reset_in_both_modes(tmp_path):
    """The alpha reset (model.cpp:464-479) inside the loop, in both builds of model_fused.inl: refine every
    3, reset interval 9 -> reset at step 3, nothing else until step 6.  Default build: the registered
    parameter is clamped in place and its moments zeroed (the evident intent), the opacities keep training.
    -DGS_FUSED_REFERENCE_ALPHA_RESET: what the reference's statements do — `opacities` re-bound to a clamped
    copy the optimiser does not know, moments left alone, nothing updates them until a refinement
    re-registers the tensor.  Each against the Python Trainer in the same mode; the reference mode also
    against the reference's OWN statements (the CPU device of the same patched model.cpp)."""
    W, H, yaws = 160, 112, [0.0]
    params, fx, fy = separated_scene(W, H, K=4, seed=5)
    kw = dict(shDegree=1, refineEvery=3, warmupLength=2, resetAlphaEvery=3, numCameras=0)
    logit = np.float32(np.log(np.float32(0.2) / (np.float32(1.0) - np.float32(0.2))))
    assert (params[3] > logit + 0.5).all()       # every opacity is above the reset value: all get clamped

    case, c, gts = make_case(params, fx, fy, W, H, yaws, 5, **kw)
    # ---- the reference's actual behaviour -----------------------------------------------------------
    ref_gpu, out = run_shim(tmp_path, case, "gpu", "ra", SHIM_REFALPHA)
    assert "Alpha reset" in out
    ref_cpu, _ = run_shim(tmp_path, case, "cpu", "ra", SHIM_REFALPHA)   # model.cpp:464-479 as written
    ref_py = run_trainer(params, fx, fy, W, H, yaws, 5, gts, c, reference_alpha_reset=True)
    for r in (ref_gpu, ref_cpu, ref_py):          # frozen at the reset value through steps 4 and 5
        assert np.abs(r["p3"] - logit).max() < 1e-6
    assert rel(ref_gpu["m3"], ref_py["m3"]) < 2e-5 and rel(ref_gpu["v3"], ref_py["v3"]) < 2e-5
    assert np.abs(ref_gpu["m3"]).max() > 0        # ... and not zeroed
    assert rel(ref_gpu["m3"], ref_cpu["m3"]) < 2e-3 and rel(ref_gpu["v3"], ref_cpu["v3"]) < 4e-3
    assert np.abs(ref_gpu["losses"] - ref_py["losses"]).max() < 2e-6
    assert np.abs(ref_gpu["losses"] - ref_cpu["losses"]).max() < 2e-5
    for k in ("p0", "p1", "p2", "p4", "p5"):
        assert rel(ref_gpu[k], ref_py[k].reshape(ref_gpu[k].shape)) < 2e-5, k
    # ---- the evident intent (default build) -----------------------------------------------------------
    gpu, out = run_shim(tmp_path, case, "gpu", "da")
    assert "Alpha reset" in out
    py = run_trainer(params, fx, fy, W, H, yaws, 5, gts, c)
    assert np.abs(gpu["p3"] - logit).max() > 1e-3                       # two Adam steps away from the clamp
    assert rel(gpu["p3"], py["p3"]) < 2e-5
    assert rel(gpu["m3"], py["m3"]) < 2e-5 and rel(gpu["v3"], py["v3"]) < 2e-5
    assert np.abs(gpu["losses"] - py["losses"]).max() < 2e-6
    # the moments restarted from zero at step 3: two steps of history only
    assert np.abs(gpu["v3"]).max() < np.abs(ref_gpu["v3"]).max()
    # both modes render the same first four iterations (the reset shows from step 4's loss on ... equal too:
    # the rendered opacities are the clamped ones in both; they part at step 5, after one more update)
    assert np.abs(gpu["losses"][:4] - ref_gpu["losses"][:4]).max() < 2e-6
    assert abs(gpu["losses"][4] - ref_gpu["losses"][4]) > 0

    # ---- the refinement at step 6 re-registers the opacities in the reference mode too ----------------
    case8, c8, gts8 = make_case(params, fx, fy, W, H, yaws, 8, densifyGradThresh=1e-7, **kw)
    ref8, out = run_shim(tmp_path, case8, "gpu", "rb", SHIM_REFALPHA)
    assert "Added" in out and ref8["counts"][4] == params[0].shape[0] and ref8["counts"][5] != params[0].shape[0]
    py8 = run_trainer(params, fx, fy, W, H, yaws, 8, gts8, c8, reference_alpha_reset=True)
    assert py8["N"] == int(ref8["counts"][-1])
    assert ref8["m3"].shape[0] == py8["N"]
    assert np.abs(ref8["p3"] - logit).max() > 1e-3                      # training again after step 6
