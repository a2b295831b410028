#!/usr/bin/env python3
"""End-to-end training on a synthetic multi-view capture — the stand-in for BASELINE config 5
(banana COLMAP to 7 k iterations: the dataset is not available offline).

A ground-truth Gaussian set is rendered from a ring of cameras with this repo's renderer; training
starts, like OpenSplat does from SfM points (model.hpp:33-60), from a noisy subsample of positions
and colours with kNN-derived scales, and runs the reference's loop (opensplat.cpp:151-170) through
`opensplat_amd.train.Trainer`: Model::forward -> mainLoss (L1 + SSIM) -> backward -> Adam ->
scheduler -> afterTrain (densification schedule of the CLI defaults).  Reports PSNR on held-out
cameras, Gaussian count, iterations/s and a PLY save / load round trip (the reference's CPU path on
the same initial set is timed by `python bench.py --train-cpu-baselines`).

  python scripts/train_synthetic.py [--iters 3000] [--via-colmap]   -> one JSON object on stdout
"""
import argparse
import json
import math
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from opensplat_amd import io, scenes, train  # noqa: E402
from train_synthetic_inputs import C0, ground_truth, make_camera, sfm_like_init  # noqa: E402  (scripts/)

DEV = torch.device("cuda:0")


def rotmat_to_quat(R):
    """[w, x, y, z] of a rotation matrix (COLMAP's qvec)."""
    R = np.asarray(R, np.float64)
    w = math.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2])) / 2.0
    x = math.sqrt(max(0.0, 1.0 + R[0, 0] - R[1, 1] - R[2, 2])) / 2.0
    y = math.sqrt(max(0.0, 1.0 - R[0, 0] + R[1, 1] - R[2, 2])) / 2.0
    z = math.sqrt(max(0.0, 1.0 - R[0, 0] - R[1, 1] + R[2, 2])) / 2.0
    x = math.copysign(x, R[2, 1] - R[1, 2]); y = math.copysign(y, R[0, 2] - R[2, 0])
    z = math.copysign(z, R[1, 0] - R[0, 1])
    return np.array([w, x, y, z])


def psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return 10.0 * math.log10(1.0 / max(mse, 1e-12))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3000)
    ap.add_argument("--gt-gaussians", type=int, default=20000)
    ap.add_argument("--init-points", type=int, default=6000)
    ap.add_argument("--width", type=int, default=384)
    ap.add_argument("--height", type=int, default=288)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--num-downscales", type=int, default=2)
    ap.add_argument("--via-colmap", action="store_true",
                    help="write the capture as a COLMAP project (sparse/0/*.bin + images/*.npy) and train "
                         "from what opensplat_amd.colmap reads back: poses normalised like the reference, "
                         "Model-style initialisation from the sparse points (row f3)")
    ap.add_argument("--graph", action="store_true",
                    help="Trainer(experimental_graph=True): every iteration replayed as one captured HIP graph")
    ap.add_argument("--no-segments", action="store_true",
                    help="Trainer(segmented=False): the one-pass compositing backward on every frame")
    ap.add_argument("--reference-schedules", action="store_true",
                    help="the reference's CLI defaults as they are (--sh-degree-interval 1000, "
                         "--resolution-schedule 3000: what `opensplat -n 7000` runs) instead of schedules "
                         "scaled to the run's length")
    ap.add_argument("--cpu-baseline", action="store_true",
                    help="also time the reference's own CPU chain (oracle/_ref) on one iteration of the initial "
                         "set at every resolution of the schedule, and extrapolate the run")
    a = ap.parse_args()
    rs = np.random.RandomState(0)
    K, W, H = 16, a.width, a.height
    bg = np.array([0.0, 0.0, 0.0], np.float32)

    n_train, n_test = 24, 4
    cams = []
    for i in range(n_train + n_test):
        ang = 2.0 * math.pi * (i + (0.5 if i >= n_train else 0.0)) / (n_train if i < n_train else n_test)
        h = 0.8 * math.sin(3.0 * ang) if i < n_train else 0.3
        cams.append(make_camera((3.5 * math.cos(ang), h, 3.5 * math.sin(ang)), W, H))
    gt_params = ground_truth(a.gt_gaussians, K, rs)
    G = train.Trainer(*gt_params, DEV)
    images = [G.render(c, bg, 3).clone() for c in cams]
    torch.cuda.synchronize()

    init = sfm_like_init(gt_params, a.init_points, K, rs)
    if a.via_colmap:
        from opensplat_amd import colmap
        tmp = tempfile.mkdtemp(prefix="capture_")
        os.makedirs(os.path.join(tmp, "images"))
        ccams, w2c = [], []
        for i, (c, img) in enumerate(zip(cams, images)):
            np.save(os.path.join(tmp, "images", "%05d.npy" % i),
                    np.clip(np.rint(img.cpu().numpy() * 255.0), 0, 255).astype(np.uint8))
            ccams.append(colmap.Camera(id=i + 1, width=W, height=H, fx=c["fx"], fy=c["fy"], cx=c["cx"], cy=c["cy"]))
            w2c.append((rotmat_to_quat(c["viewmat"][:3, :3]), c["viewmat"][:3, 3]))
        rgb8 = np.clip(np.rint((init[4] * C0 + 0.5) * 255.0), 0, 255).astype(np.uint8)
        colmap.write_colmap(tmp, ccams, w2c, init[0], rgb8)
        data = colmap.read_colmap(tmp)
        for c in data.cameras:
            colmap.load_image(c)
        cams = [colmap.render_camera(c) for c in data.cameras]
        images = [torch.from_numpy(c.image).to(DEV) for c in data.cameras]
        init = colmap.init_from_points(data.points_xyz, data.points_rgb, sh_degree=3)
    # schedules scaled to the run: the CLI defaults (--sh-degree-interval 1000, --num-downscales 2,
    # --resolution-schedule 3000) are meant for 30 000 iterations
    sched = dict(sh_degree_interval=1000, resolution_schedule=3000) if a.reference_schedules else \
        dict(sh_degree_interval=max(a.iters // 4, 1), resolution_schedule=max(a.iters // 6, 1))
    T = train.Trainer(*init, DEV, max_steps=a.iters, ssim_weight=0.2, num_cameras=n_train,
                      morton_order=True, num_downscales=a.num_downscales, experimental_graph=a.graph,
                      segmented=not a.no_segments, **sched)
    n_initial = T.N
    sh_interval = T.sh_degree_interval

    def reduced(cam, f):
        """Camera block of Model::forward at scaleFactor f (model.cpp:85-92)."""
        if f == 1:
            return cam
        c = dict(cam)
        c.update(fx=cam["fx"] / f, fy=cam["fy"] / f, cx=cam["cx"] / f, cy=cam["cy"] / f,
                 W=int(cam["W"] / f), H=int(cam["H"] / f))
        fovx, fovy = 2.0 * math.atan(c["W"] / (2.0 * c["fx"])), 2.0 * math.atan(c["H"] / (2.0 * c["fy"]))
        c["projmat"] = (scenes.projection_matrix(0.001, 1000.0, fovx, fovy) @ cam["viewmat"]).astype(np.float32)
        return c

    pyramid, prepared = {}, {}

    def target(ci, f):
        """Camera::getImage(downscaleFactor) (input_data.cpp:96-114): area-averaged pyramid, cached."""
        if f == 1:
            return images[ci]
        if (ci, f) not in pyramid:
            img = images[ci][: (H // f) * f, : (W // f) * f].permute(2, 0, 1)[None]
            pyramid[(ci, f)] = torch.nn.functional.avg_pool2d(img, f)[0].permute(1, 2, 0).contiguous()
        return pyramid[(ci, f)]

    def evaluate():
        vals = []
        for c, img in zip(cams[n_train:], images[n_train:]):
            deg = min(T.step_count // sh_interval, 3)
            vals.append(psnr(T.render(c, bg, deg).clamp(0, 1), img))
        return float(np.mean(vals))

    curve = [{"step": 0, "psnr": evaluate(), "gaussians": T.N}]
    order = np.random.RandomState(1)
    refinements = []
    torch.cuda.synchronize()
    t0 = time.time()
    train_time = 0.0
    for step in range(1, a.iters + 1):
        ci = int(order.randint(0, n_train))
        deg = T.degrees_to_use(step)             # model.cpp:178
        f = T.downscale_factor(step)             # model.cpp:249-251
        if a.graph:     # (the 36-float camera block and the GsCamera of a (camera, scale) pair, made once)
            if (ci, f) not in prepared:
                prepared[(ci, f)] = T.prepare_camera(reduced(cams[ci], f))
            loss = T.train_step(prepared[(ci, f)], target(ci, f), bg, deg)
        else:
            loss = T.train_step(reduced(cams[ci], f), target(ci, f), bg, deg)
        if step % max(a.iters // 6, 1) == 0:
            last_loss = [float(x) for x in loss.cpu()]   # before a refinement reallocates buffers
        c = T.after_train(step)
        if c is not None:
            refinements.append({"step": step, "added": c["added"], "culled": c["culled"], "gaussians": T.N})
        if step % max(a.iters // 6, 1) == 0:
            torch.cuda.synchronize()
            train_time += time.time() - t0
            curve.append({"step": step, "psnr": evaluate(), "gaussians": T.N,
                          "loss": last_loss})
            torch.cuda.synchronize()
            t0 = time.time()
    torch.cuda.synchronize()
    train_time += time.time() - t0

    # save / load round trip through the Inria-compatible PLY (model.cpp:505-562, 640-767)
    params = dict(means=T.means, log_scales=T.log_scales, quats=T.quats,
                  opacity_logits=T.opacity_logits.view(-1, 1), features_dc=T.features_dc,
                  features_rest=T.features_rest)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "splat.ply")
        io.save(path, params, a.iters)
        loaded, step_loaded = io.load_ply(path)
        size = os.path.getsize(path)
        io.save(os.path.join(d, "scene.splat"), params, a.iters)
        splat_size = os.path.getsize(os.path.join(d, "scene.splat"))
    T2 = train.Trainer(loaded["means"], loaded["log_scales"], loaded["quats"], loaded["opacity_logits"],
                       loaded["features_dc"], loaded["features_rest"], DEV)
    same = bool(torch.equal(T2.render(cams[-1], bg, 3), T.render(cams[-1], bg, 3)))

    out = {"workload": f"synthetic capture: {a.gt_gaussians} ground-truth Gaussians, {n_train} training + "
                       f"{n_test} held-out cameras at {W}x{H}, SH degree 3; {a.init_points} initial points; "
                       f"{a.iters} iterations with the reference's densification defaults",
           "input": "COLMAP project on disk (opensplat_amd.colmap)" if a.via_colmap else "in-memory capture",
           "psnr_curve": curve, "refinements": refinements, "final_gaussians": T.N,
           "schedules": "reference CLI defaults" if a.reference_schedules else "scaled to the run",
           "captured_iterations": T.graph_stats if a.graph else None,
           "initial_gaussians": n_initial,
           "train_seconds": train_time, "iterations_per_s": a.iters / train_time,
           "ply_bytes": size, "splat_bytes": splat_size, "ply_round_trip_step": step_loaded,
           "ply_round_trip_renders_identically": same}

    if a.cpu_baseline:
        out["cpu_baseline"] = cpu_iterations(init, cams[0], W, H, a, T)
    print(json.dumps(out))


def cpu_iterations(init, cam, W, H, a, T):
    """The reference's own CPU chain (oracle/_ref: ProjectGaussiansCPU -> SphericalHarmonicsCPU ->
    RasterizeGaussiansCPU forward + backward, then ssim.cpp's loss) on ONE iteration of the initial point set at
    each resolution the schedule visits; the run's CPU time is extrapolated from the iterations spent at each
    (lower bound: the set grows, the optimiser and afterTrain are not counted)."""
    import oracle

    if not oracle.have_reference():
        return {"error": "oracle/_ref is not built"}
    R = oracle.reference()
    means, ls, q, lo, dc, rest = [np.asarray(x, np.float32) for x in init]
    vm = np.asarray(cam["viewmat"], np.float32)
    cam_pos = (-vm[:3, :3].T @ vm[:3, 3]).astype(np.float32)
    dirs = means - cam_pos
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    coeffs = np.concatenate([dc[:, None, :], rest], 1)
    res, total = {}, 0.0
    for f in sorted({T.downscale_factor(s) for s in range(1, a.iters + 1)}, reverse=True):
        Wc, Hc = int(W / f), int(H / f)
        fovx, fovy = 2.0 * math.atan(Wc / (2.0 * cam["fx"] / f)), 2.0 * math.atan(Hc / (2.0 * cam["fy"] / f))
        pm = (scenes.projection_matrix(0.001, 1000.0, fovx, fovy) @ vm).astype(np.float32)
        v = np.random.RandomState(2).uniform(-1e-4, 1e-4, (Hc, Wc, 3)).astype(np.float32)
        r = R.chain_fwd_bwd(means, np.exp(ls), q / np.linalg.norm(q, axis=1, keepdims=True), dirs, coeffs,
                            (1 / (1 + np.exp(-lo))).astype(np.float32), vm, pm, cam["fx"] / f,
                            cam["fy"] / f, cam["cx"] / f, cam["cy"] / f, Hc, Wc, np.zeros(3, np.float32), v,
                            degrees_to_use=3)
        x, y = scenes.loss_images(Wc, Hc, seed=2)
        R.main_loss(x, y, 0.2)
        n_it = sum(1 for s in range(1, a.iters + 1) if T.downscale_factor(s) == f)
        sec = (r["fwd_ms"] + r["bwd_ms"] + R.last_ms) / 1e3
        res["%dx%d" % (Wc, Hc)] = {"iterations": n_it, "render_fwd_bwd_s": (r["fwd_ms"] + r["bwd_ms"]) / 1e3,
                                   "main_loss_s": R.last_ms / 1e3}
        total += n_it * sec
    return {"kind": "reference", "cores": os.cpu_count(), "gaussians": int(means.shape[0]),
            "per_resolution": res, "extrapolated_run_seconds_lower_bound": total,
            "sample": "one iteration (render forward + backward + loss) of the INITIAL point set per resolution"}


if __name__ == "__main__":
    main()
