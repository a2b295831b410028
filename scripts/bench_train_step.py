#!/usr/bin/env python3
"""SURVEY.md §8 row f2 measurement: the fused loss (L1 + SSIM forward+backward) at 1920x1080 and
the six-group Adam step at N = 1 M Gaussians / SH degree 3 (59 M parameters), HIP-event timed on
one MI355X, with
  * the same ops as the reference issues them, run by torch on the SAME GPU (grouped conv2d SSIM +
    autograd; torch::optim::Adam's op sequence per group) — what OpenSplat's GPU build executes;
(the reference's own code on the host CPU is timed by `python bench.py --train-cpu-baselines`, the one
place outside tests/ that may touch oracle/).
Prints one JSON object.  python scripts/bench_train_step.py [--ours-only]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from opensplat_amd import cabi, scenes  # noqa: E402

DEV = torch.device("cuda:0")
W, H, N, K = 1920, 1080, 1_000_000, 16
HBM_PEAK = 8000.0  # GB/s


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def torch_ssim_loss(window):
    """ssim.cpp:7-33 + model.cpp:54-56,780-784 as torch ops (what the reference's GPU build runs)."""
    w2 = torch.outer(window, window)[None, None].expand(3, 1, 11, 11).contiguous()

    def f(rendered, gt, ssim_weight):
        img1 = gt.permute(2, 0, 1)[None]
        img2 = rendered.permute(2, 0, 1)[None]
        mu1 = F.conv2d(img1, w2, padding=5, groups=3)
        mu2 = F.conv2d(img2, w2, padding=5, groups=3)
        mu1Sq, mu2Sq, mu1mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
        s1 = F.conv2d(img1 * img1, w2, padding=5, groups=3) - mu1Sq
        s2 = F.conv2d(img2 * img2, w2, padding=5, groups=3) - mu2Sq
        s12 = F.conv2d(img1 * img2, w2, padding=5, groups=3) - mu1mu2
        C1, C2 = 0.01 ** 2, 0.03 ** 2
        m = ((2.0 * mu1mu2 + C1) * (2.0 * s12 + C2)) / ((mu1Sq + mu2Sq + C1) * (s1 + s2 + C2))
        return (1.0 - ssim_weight) * torch.abs(gt - rendered).mean() + ssim_weight * (1.0 - m.mean())
    return f


def torch_adam_step(p, g, m, v, lr, step, b1=0.9, b2=0.999, eps=1e-8):
    """torch::optim::Adam::step's op sequence (libtorch C++ frontend, one parameter)."""
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def torch_refine_ms(P, M, V, gnorm, vis, m2d, W, H):
    """Model::afterTrain's densification branch (model.cpp:345-458 + addToOptimizer /
    removeFromOptimizer) as the torch ops the reference issues, on the GPU; wall time incl. syncs."""
    import time

    def run():
        means, scales, quats, opac, dc, rest = P
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        avg = (gnorm / vis) * 0.5 * float(max(W, H))
        high = avg > 0.0002
        splits = scales.exp().max(-1)[0] > 0.01
        splits = (splits | (m2d > 0.05)) & high
        n_splits = int(splits.sum().item())
        smp = torch.randn((2 * n_splits, 3), device=means.device)
        scaled = torch.exp(scales[splits].repeat(2, 1)) * smp
        qs = quats[splits] / quats[splits].norm(dim=-1, keepdim=True)
        u = F.normalize(qs.repeat(2, 1), dim=-1)
        w, x, y, z = u.unbind(-1)
        rots = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
                            torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
                            torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], -2)
        split_means = torch.bmm(rots, scaled[..., None]).squeeze(-1) + means[splits].repeat(2, 1)
        split_scales = torch.log(torch.exp(scales[splits]) / 1.6).repeat(2, 1)
        dups = (scales.exp().max(-1)[0] <= 0.01) & high
        new = [torch.cat([means, split_means, means[dups]]), torch.cat([scales, split_scales, scales[dups]]),
               torch.cat([quats, quats[splits].repeat(2, 1), quats[dups]]),
               torch.cat([opac, opac[splits].repeat(2, 1), opac[dups]]),
               torch.cat([dc, dc[splits].repeat(2, 1), dc[dups]]),
               torch.cat([rest, rest[splits].repeat(2, 1, 1), rest[dups]])]
        m2 = torch.cat([m2d, torch.zeros(2 * n_splits + int(dups.sum().item()), device=means.device)])
        si, di = torch.where(splits)[0], torch.where(dups)[0]
        states = []
        for st in (M, V):
            grown = []
            for t in st:
                t = torch.cat([t, torch.zeros_like(t[si]).repeat(2, *([1] * (t.dim() - 1)))])
                t = torch.cat([t, torch.zeros_like(t[di])])
                grown.append(t)
            states.append(grown)
        splits_mask = torch.cat([splits, torch.zeros(2 * n_splits + int(dups.sum().item()), dtype=torch.bool,
                                                     device=means.device)])
        culls = (torch.sigmoid(new[3]) < 0.1).squeeze() | splits_mask
        culls |= (torch.exp(new[1]).max(-1)[0] > 0.5) | (m2 > 0.15)
        if int(culls.sum().item()) > 0:
            new = [t[~culls] for t in new]
            states = [[t[~culls] for t in st] for st in states]
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3
    run()
    return min(run() for _ in range(3))


def main():
    out = {"workload": f"loss at {W}x{H}; Adam over N={N} Gaussians, SH degree 3 (K={K}): "
                       f"{N * (3 * K + 11)} parameters in six groups"}
    rendered_np, gt_np = scenes.loss_images(W, H, seed=1)
    rendered, gt = torch.from_numpy(rendered_np).to(DEV), torch.from_numpy(gt_np).to(DEV)
    ws = torch.empty(cabi.lib().gs_loss_workspace_bytes(W, H), device=DEV, dtype=torch.uint8)
    buf = (torch.empty(3, device=DEV), torch.empty_like(rendered))
    ms = timeit(lambda: cabi.main_loss(rendered, gt, 0.2, 1.0, True, out=buf, workspace=ws))
    P = W * H
    loss_bytes = P * (24 + 108) + P * (108 + 24 + 12)   # k_ssim_maps + k_ssim_grad, DESIGN.md §11
    out["loss"] = {"ms": ms, "algorithmic_bytes": loss_bytes,
                   "achieved_GBs": loss_bytes / ms / 1e6, "frac_of_hbm_peak": loss_bytes / ms / 1e6 / HBM_PEAK,
                   "value": [float(x) for x in buf[0].cpu()]}
    buf0 = (torch.empty(3, device=DEV), torch.empty_like(rendered))
    ms0 = timeit(lambda: cabi.main_loss(rendered, gt, 0.0, 1.0, True, out=buf0, workspace=ws))
    out["loss_l1_only"] = {"ms": ms0}
    ours_only = "--ours-only" in sys.argv   # profiling runs: skip the torch-op comparisons
    if not ours_only:
        # the reference's op sequence on this GPU
        window = torch.tensor(cabi.ssim_window(), device=DEV)
        f = torch_ssim_loss(window)

        def torch_loss():
            r = rendered.detach().requires_grad_(True)
            f(r, gt, 0.2).backward()
            return r.grad
        out["loss_torch_ops_same_gpu"] = {"ms": timeit(torch_loss, reps=5, warm=2)}
        g_t = torch_loss()
        out["loss"]["max_abs_grad_diff_vs_torch_ops"] = float((g_t - buf[1]).abs().max())
        out["loss"]["max_abs_grad"] = float(g_t.abs().max())

    # Adam: Model's six groups (model.cpp:61-66)
    sizes = [N * 3, N * 3, N * 4, N * 3, N * (K - 1) * 3, N]
    lrs = [0.00016, 0.005, 0.001, 0.0025, 0.000125, 0.05]
    total = sum(sizes)
    gen = torch.Generator(device=DEV).manual_seed(0)
    flat = [torch.randn(total, device=DEV, generator=gen) for _ in range(2)] + \
           [torch.zeros(total, device=DEV) for _ in range(2)]
    groups, off = [], 0
    for sz, lr in zip(sizes, lrs):
        groups.append(tuple(t[off:off + sz] for t in flat) + (lr,))
        off += sz
    step = [0]

    def ours():
        step[0] += 1
        cabi.adam_step(groups, step[0])
    ms = timeit(ours)
    adam_bytes = total * 28
    out["adam"] = {"ms": ms, "parameters": total, "algorithmic_bytes": adam_bytes,
                   "achieved_GBs": adam_bytes / ms / 1e6, "frac_of_hbm_peak": adam_bytes / ms / 1e6 / HBM_PEAK}
    tstep = [0]

    def theirs():
        tstep[0] += 1
        for p, g, m, v, lr in groups:
            torch_adam_step(p, g, m, v, lr, tstep[0])
    if not ours_only:
        out["adam_torch_ops_same_gpu"] = {"ms": timeit(theirs, reps=5, warm=2)}

    # ---- the whole iteration (opensplat.cpp:151-170) at C2: render + loss + backward + Adam ------
    from opensplat_amd import train
    s_ = scenes.config_c2()
    raw = scenes.raw_parameters(s_)
    T = train.Trainer(*raw, DEV, max_steps=30000, ssim_weight=0.2)
    cam = dict(viewmat=s_.viewmat, projmat=s_.projmat, fx=s_.fx, fy=s_.fy, cx=s_.cx, cy=s_.cy,
               W=s_.W, H=s_.H)
    it_ms = timeit(lambda: T.train_step(cam, gt, s_.background, s_.degrees_to_use), reps=30, warm=5)
    stages = {}
    T.render(cam, s_.background, s_.degrees_to_use)
    stages["render"] = timeit(lambda: T.render(cam, s_.background, s_.degrees_to_use), reps=10)
    lo = lambda: cabi.main_loss(T.fwd["img_clamped"], gt, 0.2, 1.0, True, out=T.loss_out,
                                workspace=T.loss_ws)
    stages["loss"] = timeit(lo, reps=10)
    stages["backward"] = timeit(lambda: T.backward(T.loss_out[1]), reps=10)
    stages["adam"] = timeit(T.optimizer_step, reps=10)
    out["iteration"] = {"workload": "C2 scene (1 M Gaussians, 1920x1080, SH degree 3), one camera: "
                                    "Model::forward + mainLoss + backward + optimizersStep",
                        "ms": it_ms, "iterations_per_s": 1e3 / it_ms, "stage_ms": stages,
                        "loss_after": [float(x) for x in T.loss_out[0].cpu()]}

    # ---- camera batches through the library (Trainer.train_step_batch: two cameras in flight, ONE Adam step) ----
    from opensplat_amd.scenes import C4_YAWS, yaw_camera
    cams = []
    for y in C4_YAWS[:4]:
        vm, pm = yaw_camera(s_.W, s_.H, y)
        cams.append(dict(viewmat=vm, projmat=pm, fx=s_.fx, fy=s_.fy, cx=s_.cx, cy=s_.cy, W=s_.W, H=s_.H))
    gts = [gt] * 4
    batch = {}
    one = timeit(lambda: T.train_step(cams[0], gt, s_.background, s_.degrees_to_use), reps=30, warm=5)
    batch["c1_train_step"] = {"ms_per_camera": one, "cameras_per_s": 1e3 / one}
    for c in (2, 4):
        for serial in (True, False):
            ms_b = timeit(lambda: T.train_step_batch(cams[:c], gts[:c], s_.background, s_.degrees_to_use,
                                                     serial=serial), reps=20, warm=4)
            batch["c%d_%s" % (c, "serial" if serial else "two_in_flight")] = {
                "ms_per_step": ms_b, "ms_per_camera": ms_b / c, "cameras_per_s": c * 1e3 / ms_b,
                "per_camera_speedup_vs_c1": one / (ms_b / c)}
    batch["note"] = ("C2 Gaussians, the first c C4 cameras (yaw offsets), render + loss + backward per camera, ONE "
                     "optimiser step per batch; serial = the same cameras one after the other on one lane")
    out["camera_batches"] = batch

    # ---- row f4: Model::afterTrain on the device ------------------------------------------------
    N_ = s_.N
    stats = [torch.zeros(N_, device=DEV) for _ in range(3)]
    cabi.densify_stats(T.rgrads["v_xy"], T.proj["radii"], float(max(s_.W, s_.H)), True, *stats)
    st_ms = timeit(lambda: cabi.densify_stats(T.rgrads["v_xy"], T.proj["radii"],
                                              float(max(s_.W, s_.H)), False, *stats))
    prob = scenes.densify_problem(N, K, seed=11)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    P_, M_, V_ = ([d(a) for a in prob[k]] for k in ("params", "exp_avg", "exp_avg_sq"))
    gs_, vc_, m2_ = d(prob["xys_grad_norm"]), d(prob["vis_counts"]), d(prob["max_2d_size"])
    cfg = cabi.densify_config(prob["width"], prob["height"])
    import time as _t

    def refine():
        torch.cuda.synchronize()
        t0 = _t.perf_counter()
        r = cabi.densify(cfg, P_, M_, V_, gs_, vc_, m2_)
        torch.cuda.synchronize()
        return (_t.perf_counter() - t0) * 1e3, r[3]
    refine()
    times = [refine() for _ in range(5)]
    out["densify"] = {"stats_ms_per_iteration": st_ms,
                      "stats_algorithmic_bytes": N_ * (8 + 4 + 6 * 4),
                      "refine_wall_ms": min(t for t, _ in times), "refine_counts": times[0][1],
                      "refine_workload": f"{N} Gaussians, K={K}: parameters + 2 Adam moments "
                                         f"({3 * N * (3 * K + 11) * 4 / 1e6:.0f} MB in), incl. the "
                                         "count read-back and torch.randn"}
    if not ours_only:
        out["densify"]["refine_torch_ops_same_gpu_wall_ms"] = torch_refine_ms(P_, M_, V_, gs_, vc_, m2_,
                                                                              prob["width"], prob["height"])

    print(json.dumps(out))


if __name__ == "__main__":
    main()
