#!/usr/bin/env python3
"""bench.py — full forward+backward rasterizations per second of the MI355X rasterizer.

One "step" = one complete pass of the hot path over one camera (BASELINE.json metric):
    project fwd -> SH fwd (+0.5, clamp_min 0) -> pack / count / scan / scatter / per-tile sort
    -> composite fwd -> composite bwd -> SH bwd (clamp mask) -> project bwd
    [-> RCCL all-reduce if N > 1]
producing the image and the gradients of OpenSplat's six parameter tensors (means, scales, quats,
opacities, featuresDc, featuresRest).  The element-wise glue between the three operators
(model.cpp:114,176-177,192: cat, view directions, +0.5 / clamp_min) runs inside the SH kernels
(row f1 variants gs_sh_forward_fused / gs_sh_backward_fused) — same arithmetic, no extra passes.
The intersection count that sizes the id list is taken from the previous step and validated
while the forward kernel runs (an event wait on the scan kernel, see gs_bin_sort in
include/gsplat_hip.h); a forward whose guess was too small is repeated inside the timed region.  Every stage goes through the C ABI of
libgsplat_hip.so (include/gsplat_hip.h); torch only provides device memory, the stream and, for
N > 1, torch.distributed (nccl == RCCL).

Workload: N = 1 -> BASELINE configs[1] ("C2": 1 M Gaussians, 1920x1080, SH degree 3, 16x16
tiles, seeded synthetic scene of SURVEY.md §8d).  N > 1 -> configs[3] ("C4"): the same number of
shared Gaussians, one camera per rank, gradients all-reduced over xGMI; weak scaling (one camera
per GPU), value = cameras rasterized per second by the whole job.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline      dominant kernel: algorithmic bytes / HIP-event duration vs the 8 TB/s HBM peak
  cpu_baseline  OpenSplat's own gsplat-cpu (oracle/_ref, compiled from /root/reference) timed on
                this host's cores on the same scene — only here is anything under oracle/ used.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--config", default="auto", choices=["auto", "c2", "c3", "c4", "c4-sequence"],
                    help="c4-sequence: ONE GPU cycling through the 8 C4 cameras, a different camera every "
                         "step (what a training loop does): measures the speculative binning on a moving "
                         "camera (misses = forwards repeated because the id list was too small)")
    ap.add_argument("--exchange", choices=["auto", "flat", "factored"], default="auto",
                    help="N > 1: gradient exchange — flat = ONE sum all-reduce of the whole gradient "
                         "buffer (236 B per Gaussian at K = 16); factored = all-reduce of the geometry "
                         "block (44 B) + all-gather of the colour cotangents (12 B per camera), SH "
                         "gradients formed locally (opensplat_amd/dist.py FactoredExchange); auto = "
                         "factored while cameras_per_rank x ranks <= 32")
    ap.add_argument("--cameras-per-rank", type=int, default=1,
                    help="cameras each rank renders per gradient exchange (gradients accumulated in the flat "
                         "buffer, ONE all-reduce per c rasterizations); 1 = BASELINE config 4.  With --gpus 1 "
                         "and c >= 2 the step keeps TWO cameras in flight (HotPath.step_cameras) and the line "
                         "reports the serial loop beside it")
    ap.add_argument("--serial-cameras", action="store_true",
                    help="--gpus 1 --cameras-per-rank c: time the serial camera loop as the headline")
    ap.add_argument("--fast-exp", action="store_true",
                    help="hardware exp instead of the glibc-bit-exact one (not the parity mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--order", default="given", choices=["given", "morton", "tile"],
                    help="EXPERIMENT: permute the scene's Gaussians before the run — 3-D Morton order "
                         "of the means, or by the 16x16 tile of the projected centre (upper bound)")
    ap.add_argument("--train-cpu-baselines", action="store_true",
                    help="time the reference's CPU code for the training-step rows (mainLoss, Adam, one "
                         "iteration's render + loss of scripts/train_synthetic.py's initial set), print "
                         "one JSON object and exit; needs no GPU")
    ap.add_argument("--stage-kernels", action="store_true",
                    help="run projection / SH / pack and their backwards as separate kernels "
                         "(operator granularity) instead of the fused per-Gaussian kernels")
    ap.add_argument("--hot", type=float, default=0.0,
                    help="experiment: fraction of the Gaussians concentrated in a 48x48 px window")
    ap.add_argument("--cpu-gaussians", type=int, default=0,
                    help="Gaussians in the CPU-baseline sample (0 = the whole workload)")
    return ap.parse_args()


# The hot path object (every buffer preallocated) and the camera batch with two cameras in flight live in the
# product: opensplat_amd/pipeline.py.  bench.py only drives and times them.
from opensplat_amd.pipeline import HotPath as Pipeline  # noqa: E402


def reorder_scene(s, how):
    """Permute the Gaussians of scene `s` (any order is a valid input: the result is the same image
    and the same gradients, permuted)."""
    if how == "tile":
        fx, fy = s.fx, s.fy
        p = s.means @ s.viewmat[:3, :3].T + s.viewmat[:3, 3]
        u = np.clip(p[:, 0] / p[:, 2] * fx + s.cx, 0, s.W - 1).astype(np.int64) // 16
        v = np.clip(p[:, 1] / p[:, 2] * fy + s.cy, 0, s.H - 1).astype(np.int64) // 16
        key = v * 100000 + u
    else:
        lo, hi = s.means.min(0), s.means.max(0)
        q = ((s.means - lo) / (hi - lo + 1e-9) * 1023).astype(np.uint64)

        def spread(x):
            x = (x | (x << 16)) & 0x030000FF
            x = (x | (x << 8)) & 0x0300F00F
            x = (x | (x << 4)) & 0x030C30C3
            x = (x | (x << 2)) & 0x09249249
            return x
        key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    perm = np.argsort(key, kind="stable")
    for name in ("means", "scales", "quats", "opacities", "sh_coeffs", "dirs", "colors"):
        a = getattr(s, name, None)
        if a is not None:
            setattr(s, name, np.ascontiguousarray(a[perm]))


def train_cpu_baselines():
    """cpu_baseline leg for rows f2 / f4 / the end-to-end run: the reference's own code (oracle/_ref:
    ssim.cpp, libtorch Adam, the gsplat-cpu operator chain) on this host's cores, bounded samples."""
    import importlib.util

    import torch

    import oracle
    from opensplat_amd import scenes

    if not oracle.have_reference():
        return {"error": "oracle/_ref is not built"}
    R = oracle.reference()
    out = {"kind": "reference", "threads_torch": torch.get_num_threads(), "cores": os.cpu_count()}
    W, H = 1920, 1080
    rendered, gt = scenes.loss_images(W, H, seed=1)
    R.main_loss(rendered[:270], gt[:270], 0.2)                 # warm-up on a quarter frame
    R.main_loss(rendered, gt, 0.2)
    out["main_loss_1080p_ms"] = R.last_ms
    n, total = 4_000_000, 59_000_000
    p0, grads = scenes.adam_problem(n, 3, 1)
    t0 = time.time()
    R.adam_steps(p0, grads, 0.005)
    out["adam_ms_per_59M_params"] = (time.time() - t0) / 3 * 1e3 * total / n
    out["adam_sample"] = "3 libtorch Adam steps over %d parameters, scaled to 59 M" % n
    # one iteration's render + loss on the initial set of scripts/train_synthetic.py
    spec = importlib.util.spec_from_file_location("train_synthetic",
                                                  os.path.join(ROOT, "scripts", "train_synthetic_inputs.py"))
    ts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ts)
    rs = np.random.RandomState(0)
    Wc, Hc, K, n_init = 384, 288, 16, 6000
    cam = ts.make_camera((3.5, 0.0, 0.0), Wc, Hc)
    gt_params = ts.ground_truth(20000, K, rs)
    means, ls, q, lo, dc, rest = ts.sfm_like_init(gt_params, n_init, K, rs)
    vm = cam["viewmat"]
    cam_pos = (-vm[:3, :3].T @ vm[:3, 3]).astype(np.float32)
    dirs = means - cam_pos
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    coeffs = np.concatenate([dc[:, None, :], rest], 1)
    v = np.random.RandomState(2).uniform(-1e-4, 1e-4, (Hc, Wc, 3)).astype(np.float32)
    r = R.chain_fwd_bwd(means, np.exp(ls), q / np.linalg.norm(q, axis=1, keepdims=True), dirs, coeffs,
                        (1 / (1 + np.exp(-lo))).astype(np.float32), vm, cam["projmat"], cam["fx"], cam["fy"],
                        cam["cx"], cam["cy"], Hc, Wc, np.zeros(3, np.float32), v, degrees_to_use=3)
    a, b = scenes.loss_images(Wc, Hc, seed=2)
    R.main_loss(a, b, 0.2)
    out["e2e_iteration"] = {"render_fwd_bwd_s": (r["fwd_ms"] + r["bwd_ms"]) / 1e3, "main_loss_s": R.last_ms / 1e3,
                            "sample": "render + loss of one iteration on the %d-point initial set of "
                                      "scripts/train_synthetic.py at %dx%d (no optimiser, no growth)" % (n_init, Wc, Hc)}
    return out


def algorithmic_bytes(N, K, M, P):
    """compulsory HBM traffic of one fwd+bwd, and the per-kernel shares used for
    roofline.achieved (stated in DESIGN.md §Measurement)."""
    total = N * (244 + 24 * K) + 100 * M + 40 * P
    per_stage = {
        "project_fwd": N * (40 + 44),            # read means/scales/quats, write xy conic depth radii tiles (+cov)
        "sh_fwd": N * (12 + 12 * K + 12),        # dirs + coeffs -> rgb
        "bin_sort": 24 * M + N * 52,             # write + read (key,id) once; pack reads
        "rasterize_fwd": 40 * M + 20 * P,        # id 4 + gather 36 per intersection; rgb/T/idx per pixel
        "rasterize_bwd": 40 * M + 20 * P + 36 * N,  # + 9 accumulated floats per Gaussian
        "sh_bwd": N * (24 + 12 * K),
        "project_bwd": N * (40 + 36 + 40),
    }
    return total, per_stage


def kernel_tables(N, K, M, P, stage_ms, kernel_ms, kernel_live, profiled_name):
    """(kernels, kernels_profiled, roofline_valu) for the JSON line.

    kernels           EVERY kernel of the step as THIS run measured it (instrumented pass: HIP events on the
                      launch stream right around each launch, gs_debug_timeline): mean ms per step, launches
                      per step, and — where SURVEY.md §8d / DESIGN.md §4 define them — the kernel's algorithmic
                      bytes and HBM fraction; followed by the stage rows (events at the stage boundaries)
    kernels_profiled  REPLAYED: every gs:: kernel from the committed rocprofv3 passes of the same command
                      (profiles/kernels*.json, scripts/summarize_profile.py): average duration, counter
                      traffic, VALU issue occupancy
    roofline_valu     REPLAYED: the two compositing kernels are bound by VALU issue, not by HBM.  issue_roofline =
                      wave-instructions x the measured cycles per instruction of a saturated loop in the kernel's
                      own class mix / (1024 SIMDs x 2.4 GHz) / kernel time (profiles/issue_roofline_r06.json);
                      valu_busy_frac = the counter ratio SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES over what that
                      saturated loop reads (an independent derivation of the same thing); live_lane_frac = lanes
                      doing needed work (instrumented build); useful_frac = issue_roofline x live_lane_frac"""
    # algorithmic bytes per kernel (prefix of the kernel's short name -> bytes per launch)
    kalg = [
        ("k_sh_project_pack16", N * (232 + 12) + N * (44 + 68)),
        ("k_gaussian_forward", N * (44 + 12 * K + 112)),
        ("k_gaussian_backward", N * (124 + 44 + 12 * K)),
        ("k_count_tiles", N * 32),
        ("k_scan_tiles", 16 * (P // 256)),
        ("k_scatter", N * 36 + 16 * M),
        ("k_bucket_sort", 22 * M),
        ("k_rasterize_forward", 40 * M + 20 * P),
        ("k_rasterize_backward", 40 * M + 20 * P + 36 * N),
        ("memset(gradientrecords)", 64 * N),
        ("k_project_forward", N * (40 + 44)),
        ("k_project_backward", N * (40 + 36 + 40)),
        ("k_sh_forward", N * (12 + 12 * K + 12)),
        ("k_sh_backward", N * (24 + 12 * K)),
        ("k_pack_splats", N * 52 + N * 52),
        ("k_unpack_grads", N * 100),
    ]
    kernels = []
    for name, e in kernel_live.items():
        nbytes = next((b for pre, b in kalg if name.startswith(pre)), None)
        ms = e["ms"]
        per_launch = ms / max(e["launches_per_step"], 1e-9)
        gbs = nbytes / (per_launch * 1e-3) / 1e9 if (nbytes and per_launch > 0) else None
        kernels.append({"kernel": name, "ms": ms, "launches_per_step": e["launches_per_step"],
                        "algorithmic_bytes": nbytes, "hbm_GBs": gbs,
                        "hbm_frac": None if gbs is None else gbs / HBM_PEAK_GBS,
                        "timed_by": "HIP events around the launch, this run (instrumented pass)"})
    alg = {
        "gaussian_fwd": ("k_sh_project_pack16 / k_gaussian_forward", N * (232 + 12) + N * (44 + 68)
                         if K == 16 else N * (44 + 12 * K + 112)),
        "bin_sort": ("k_count_tiles + k_scan_tiles + k_scatter + k_bucket_sort_*", N * 52 + 24 * M),
        "rasterize_fwd": ("k_rasterize_forward", 40 * M + 20 * P),
        "rasterize_bwd": ("k_rasterize_backward (the records are zeroed by the binning's count pass: gs_bin_speculative_zero)", 40 * M + 20 * P + 36 * N),
        "gaussian_bwd": ("k_gaussian_backward", N * (124 + 44 + 12 * K)),
    }
    for st, (kname, nbytes) in alg.items():
        if st not in stage_ms:
            continue
        ms = stage_ms[st]
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else None
        kernels.append({"stage": st, "kernel": kname, "ms": ms, "algorithmic_bytes": nbytes,
                        "hbm_GBs": gbs, "hbm_frac": None if gbs is None else gbs / HBM_PEAK_GBS,
                        "timed_by": "HIP events at the stage boundaries, this run (instrumented pass)"})
    prof, rv = None, None
    path = os.path.join(ROOT, "profiles", profiled_name) if profiled_name else None
    if path and os.path.exists(path):
        try:
            d = json.load(open(path))
            src = "replayed from profiles/%s (rocprofv3, tag %s) — not measured in this run" % (
                profiled_name, d.get("tag"))
            prof = {"source": src, "kernels": d["kernels"]}
            # the calibrated issue figures of the same profile (scripts/issue_roofline.py; VERDICT r05: the ratio
            # SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES has no fixed ceiling of 8 — a saturated plain-fp32 loop reads
            # 13.2, DPP / compare / SGPR-operand / fp64 loops 7.5, a loop in the forward's class mix 10.8)
            issue = {}
            try:
                issue = json.load(open(os.path.join(ROOT, "profiles", "issue_roofline_r06.json")))["profiles"].get(
                    profiled_name, {})
            except Exception:
                issue = {}
            rv = []
            for k, e in sorted(d["kernels"].items()):
                if k.startswith("k_rasterize_") and "valu_busy_of_8" in e:
                    iss = issue.get(k)
                    live = e.get("live_lane_frac")
                    busy = None if iss is None else iss["counter_frac_of_saturated"]
                    rv.append({"kernel": k, "source": src,
                               # counter reading / the reading of a SATURATED loop of this kernel's class mix
                               "valu_busy_frac": busy,
                               "counter_reading": e["valu_busy_of_8"],
                               "counter_reading_if_saturated": None if iss is None else iss["counter_reading_if_saturated"],
                               # wave-instructions x measured cycles per instruction of that mixed loop / (1024 SIMDs x
                               # 2.4 GHz) / kernel time: the same thing derived from instruction count and time alone
                               "issue_roofline": None if iss is None else iss["issue_roofline"],
                               "issue_time_us": None if iss is None else iss["issue_time_us"],
                               "loop_class_mix": None if iss is None else iss["loop_class_mix"],
                               "live_lane_frac": live,
                               "useful_frac": None if (live is None or iss is None) else iss["issue_roofline"] * live,
                               "salu_per_valu": e.get("salu_per_valu"),
                               "valu_insts_per_launch": e.get("valu_insts_per_launch"),
                               "steps_per_list_entry": e.get("steps_per_list_entry"),
                               "block_entries_per_list_entry": e.get("block_entries_per_list_entry")})
        except Exception:
            prof, rv = None, None
    return kernels, prof, rv


def exchange_probe(pipe, world, dev, iters=5):
    """N > 1: BOTH gradient exchanges timed on their own, back to back in the same run, on the step's real
    buffers — `flat` = north_star's single sum all-reduce of the whole gradient buffer (236 B per Gaussian at
    K = 16), `factored` = all-gather of the colour cotangents + all-reduce of the geometry block + the local SH
    backward over the gathered cameras.  ms = max over ranks of the mean over `iters`
    exchanges, each between barriers.  The first multi-GPU run answers the design question by itself."""
    import torch

    from opensplat_amd import dist

    s = pipe.s
    fx = pipe.fx if (pipe.fx is not None and pipe.fx.cpr == 1) else dist.FactoredExchange(s.N, s.K, 1, dev)
    fx.set_cam_pos(0, pipe.cam_pos)

    def flat():
        dist.wait_all(dist.allreduce_all_async(pipe.grads))

    def factored():
        fx.start(pipe.grads)
        fx.finish(pipe.grads, pipe.means, s.degrees_to_use)

    out = {}
    for name, fn in (("flat", flat), ("factored", factored)):
        fn()   # warm-up (communicator channels, buffers)
        torch.cuda.synchronize()
        torch.distributed.barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters * 1e3
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        out[name + "_ms"] = float(t.item())
    out["flat_bytes_moved_per_rank"] = int(2 * (world - 1) / world * pipe.grads.nbytes)
    out["factored_bytes_moved_per_rank"] = fx.bytes_moved_per_rank
    out["note"] = ("exchange alone (no render), %d iterations each, host-timed between synchronisations, max "
                   "over ranks; the gradient values are whatever the last step left (sums of sums)" % iters)
    return out


def cameras_probe(scene, dev, flags, factored, world, rank, cams, steps=6):
    """N > 1: rasterizations/s with ONE and with TWO cameras per rank and exchange (gradients of a rank's cameras
    accumulated locally, one exchange per step), measured back to back in THIS run on pipelines of their own:
    the first multi-GPU record answers DESIGN.md §7's table (exchange exposed at c = 1, amortised at c >= 2) by
    itself.  Collective: every rank takes part."""
    import torch

    out = {}
    for c in (1, 2):
        pipe = Pipeline(scene, dev, flags, factored=factored, cameras_per_rank=c)

        batch = [cams[(rank * c + j) % 8] for j in range(c)]

        def step():
            if c == 1:
                pipe.set_camera(*batch[0])
                pipe.step()
            else:
                pipe.step_cameras(batch)

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        torch.distributed.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        torch.distributed.barrier()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        out["c%d_rasterizations_per_s" % c] = world * c * steps / float(t.item())
        out["c%d_ms_per_step" % c] = float(t.item()) / steps * 1e3
        del pipe
        torch.cuda.empty_cache()
    out["note"] = "%d steps each after 3 warm-up steps, host-timed between barriers, max over ranks" % steps
    return out


def operator_binning_probe(scene, dev, passes=3):
    """The speculative binning as the C++ OPERATORS run it (RasterizeGaussians::apply through
    torch.ops, torch_ops.cpp): the eight C4 cameras in shuffled order, `passes` times — forwards
    repeated because the id-list capacity (running maximum per device and image size) was too small,
    in the first pass and afterwards.  Outside the timed region."""
    import torch

    from opensplat_amd import ops, scenes

    s = scene
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    means, scales, quats, opac = t(s.means), t(s.scales), t(s.quats), t(s.opacities)
    colors = torch.rand((s.N, 3), device=dev)
    bg = t(np.asarray(s.background, np.float32))
    ops.binning_reset()
    rs = np.random.RandomState(0)
    per_pass = []
    with torch.no_grad():
        for _ in range(passes):
            before = ops.binning_counters()[1]
            for c in rs.permutation(8):
                vm, pm = scenes.yaw_camera(s.W, s.H, scenes.C4_YAWS[c])
                p = ops.project_gaussians(means, scales, 1.0, quats, t(vm), t(pm), s.fx, s.fy, s.cx, s.cy,
                                          s.H, s.W)
                ops.rasterize_gaussians(p[0], p[1], p[2], p[3], p[4], colors, opac, s.H, s.W, bg, p[6])
            per_pass.append(ops.binning_counters()[1] - before)
    torch.cuda.synchronize()
    calls, repeats = ops.binning_counters()
    return {"cameras": 8, "passes": passes, "order": "shuffled", "forwards": calls - repeats,
            "repeated_forwards_per_pass": per_pass,
            "capacity": ops.binning_capacity(dev.index or 0, s.W, s.H),
            "policy": "running maximum of 1.125 M + 1024 per (device, width, height)"}


def moving_camera_probe(pipe, scene, cams, passes=3):
    """The timed scene under a camera that changes EVERY step (the eight C4 yaw cameras in turn, after
    one untimed pass): what a training loop sees — no frame finds its lists, records or gradient
    records warm in the caches from an identical predecessor, and the speculative id-list capacity must
    hold across views.  Rides in the default line so that the driver captures it."""
    import torch

    misses0 = pipe.misses
    for c in cams:                      # one pass to settle the capacity (its misses are reported)
        pipe.set_camera(*c)
        pipe.step()
    torch.cuda.synchronize()
    misses_first_pass = pipe.misses - misses0
    n = passes * len(cams)
    t0 = time.perf_counter()
    for i in range(n):
        pipe.set_camera(*cams[i % len(cams)])
        pipe.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"cameras": len(cams), "steps": n, "value": n / dt, "unit": "rasterizations/s",
           "ms_per_step": dt / n * 1e3, "repeated_forwards_first_pass": misses_first_pass,
           "repeated_forwards_timed": pipe.misses - misses0 - misses_first_pass,
           "note": "yaw -14 .. +14 deg around the timed camera, a different camera every step"}
    pipe.set_camera(scene.viewmat, scene.projmat)
    return out


def cpu_baseline(scene, n_sample):
    """OpenSplat's gsplat-cpu (oracle/_ref) or, if that build is absent, the C restatement."""
    import oracle

    s = scene
    n = s.N if n_sample <= 0 else min(n_sample, s.N)
    sl = slice(0, n)
    cores = os.cpu_count() or 1
    if oracle.have_reference():
        R = oracle.reference()
        t0 = time.time()
        r = R.chain_fwd_bwd(s.means[sl], s.scales[sl], s.quats[sl], s.dirs[sl], s.sh_coeffs[sl],
                            s.opacities[sl], s.viewmat, s.projmat, s.fx, s.fy, s.cx, s.cy, s.H, s.W,
                            s.background, s.v_out, degrees_to_use=s.degrees_to_use)
        wall = time.time() - t0
        sec = (r["fwd_ms"] + r["bwd_ms"]) / 1e3
        return dict(value=1.0 / sec, unit="rasterizations/s", cores=cores,
                    threads_torch=R.num_threads(), kind="reference",
                    sample="1 fwd+bwd of %d of %d Gaussians at %dx%d, SH deg %d, same scene/camera"
                           "/cotangent; compositing loops are single-threaded by construction "
                           "(gsplat_cpu.cpp:188,313), projection/SH use torch intra-op threads"
                           % (n, s.N, s.W, s.H, s.degrees_to_use),
                    fwd_ms=r["fwd_ms"], bwd_ms=r["bwd_ms"], wall_s=wall)
    O = oracle.restated()
    t0 = time.time()
    p = O.project_forward(s.means[sl], s.scales[sl], s.quats[sl], s.viewmat, s.projmat, s.fx, s.fy,
                          s.cx, s.cy, s.H, s.W)
    rgb = np.maximum(O.sh_forward(s.degrees_to_use, s.dirs[sl], s.sh_coeffs[sl]) + 0.5, 0.0)
    f = O.rasterize_forward(s.W, s.H, p["xys"], p["conics"], rgb, s.opacities[sl], s.background,
                            p["cov2d"], p["cam_depths"], want_contributors=False)
    g = O.rasterize_backward(s.W, s.H, p["xys"], p["conics"], rgb, s.opacities[sl], s.background,
                             p["cov2d"], p["cam_depths"], f["final_Ts"], f["state"], s.v_out)
    O.sh_backward(s.degrees_to_use, s.dirs[sl], s.sh_coeffs[sl], g["v_colors"])
    O.project_backward(s.means[sl], s.scales[sl], s.quats[sl], s.viewmat, s.projmat, s.fx, s.fy,
                       s.cx, s.cy, s.H, s.W, g["v_xy"], g["v_conic"])
    sec = time.time() - t0
    return dict(value=1.0 / sec, unit="rasterizations/s", cores=1, kind="port",
                sample="1 fwd+bwd of %d of %d Gaussians at %dx%d (plain-C restatement, 1 thread)"
                       % (n, s.N, s.W, s.H))


def main():
    args = parse_args()
    if args.train_cpu_baselines:
        print(json.dumps({"cpu_baselines": train_cpu_baselines()}))
        return
    import torch

    from opensplat_amd import cabi, dist, scenes

    # one rank per GPU over RCCL ("nccl"); GSPLAT_DIST_BACKEND=gloo lets the multi-rank code path be
    # exercised on a box with fewer GPUs than ranks (ranks then share devices round-robin)
    backend = os.environ.get("GSPLAT_DIST_BACKEND", "nccl")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (what torchrun would do)
        have = torch.cuda.device_count()
        if have < args.gpus and backend == "nccl":
            sys.exit("bench.py: --gpus %d requested but only %d GPU(s) visible (RCCL needs one GPU per "
                     "rank; GSPLAT_DIST_BACKEND=gloo shares devices for functional runs)" % (args.gpus, have))
        import socket

        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    rank, world, local = dist.init_from_env(backend)
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if os.environ.get("GSPLAT_BENCH_DRY_LAUNCH"):   # launcher test: report the rendezvous and stop
        print(json.dumps({"dry_launch": True, "rank": rank, "world": world, "local_rank": local}), flush=True)
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    if world > 1 and backend == "nccl" and torch.cuda.device_count() < world:
        sys.exit("bench.py: %d ranks but only %d GPU(s) visible" % (world, torch.cuda.device_count()))
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cabi.lib()  # fail loudly if the HIP library is missing

    cfg = args.config
    if cfg == "auto":
        cfg = "c2" if (world == 1 and args.cameras_per_rank == 1) else "c4"
    if cfg == "c2":
        scene = scenes.camera_scene(args.gaussians, args.width, args.height, K=16, seed=1,
                                    sigma_px=(0.5, 4.0), name="C2", hot=(args.hot, 48))
        workload = "C2: %d Gaussians, %dx%d, SH degree 3 (K=16), 16x16 tiles, seed 1" % (
            args.gaussians, args.width, args.height)
    elif cfg == "c3":
        scene = scenes.camera_scene(5 * args.gaussians, 2 * args.width, 2 * args.height, K=16, seed=2,
                                    sigma_px=(1.0, 8.0), name="C3")
        workload = "C3: %d Gaussians, %dx%d, SH degree 3" % (scene.N, scene.W, scene.H)
    elif cfg == "c4-sequence":
        assert world == 1, "--config c4-sequence is a one-GPU measurement"
        scene = scenes.config_c4(0, args.gaussians)
        workload = ("C4 sequence: %d Gaussians, the 8 C4 cameras (yaw -14 .. +14 deg) at 1920x1080 in "
                    "turn on ONE GPU, a different camera every step, SH degree 3" % args.gaussians)
    else:
        scene = scenes.config_c4(rank * args.cameras_per_rank, args.gaussians)
        workload = ("C4: %d shared Gaussians, %d cameras at 1920x1080 (%d per rank, yaw offsets), "
                    "SH degree 3, gradients exchanged over RCCL" % (args.gaussians,
                                                                   world * args.cameras_per_rank,
                                                                   args.cameras_per_rank))
    if world == 1:
        workload = workload.replace(", gradients exchanged over RCCL", ", one rank (nothing exchanged)")
    else:
        # (the backend actually in use: RCCL on a multi-GPU node, gloo when the ranks share one GPU in the tests)
        workload = workload.replace("exchanged over RCCL", "exchanged over %s" % ("RCCL" if backend == "nccl" else backend))
    # a line on anything but the plain scene, order, schedule and exponential says so in its workload string
    # (a hot-spot line must not wear the baseline's label)
    variants = []
    if args.hot:
        variants.append("hot spot: %g %% of the Gaussians in a 48-px window" % (100 * args.hot))
    if args.order != "given":
        variants.append("Gaussians re-ordered: %s" % args.order)
    if args.fast_exp:
        variants.append("hardware exp (GS_FLAG_FAST_EXP), not the parity mode")
    if args.stage_kernels:
        variants.append("operator-granular stage kernels")
    if os.environ.get("GSPLAT_BENCH_PIECES", "0") != "0":
        variants.append("backward in pieces of %s entries" % os.environ["GSPLAT_BENCH_PIECES"])
    for env in ("GSPLAT_BIN", "GSPLAT_STRIPS_FUSED", "GSPLAT_FWD_FLAGS", "GSPLAT_BWD_FLAGS", "GSPLAT_BWD_PX",
                "GSPLAT_RECORDS_ZEROED", "GSPLAT_HIP_LIB"):
        if os.environ.get(env):
            variants.append("%s=%s" % (env, os.environ[env]))
    if cfg in ("c2", "c3") and (args.gaussians != 1_000_000 or args.width != 1920 or args.height != 1080):
        variants.append("non-baseline size")
    if variants:
        workload += " + " + "; ".join(variants)
    cpr = args.cameras_per_rank if (world > 1 or cfg == "c4") else 1
    cams = [scenes.yaw_camera(scene.W, scene.H, y) for y in scenes.C4_YAWS]

    if args.order != "given":
        reorder_scene(scene, args.order)
    flags = cabi.GS_FLAG_FAST_EXP if args.fast_exp else 0
    if os.environ.get("GSPLAT_BWD_PX"):      # A/B: pixels per lane of the compositing backward (1, 2, 4)
        flags |= ({"1": 1, "2": 2, "4": 3}[os.environ["GSPLAT_BWD_PX"]]) << 21
    factored = world > 1 and not args.stage_kernels and (
        args.exchange == "factored" or (args.exchange == "auto" and cpr * world <= 32))
    pipe = Pipeline(scene, dev, flags, stage_kernels=args.stage_kernels, factored=factored,
                    cameras_per_rank=cpr)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    sequence = cfg == "c4-sequence"
    step_no = [0]

    # one GPU, several cameras per step: two of them in flight (the serial loop is timed beside it)
    two_in_flight = world == 1 and cpr >= 2 and not args.stage_kernels and not args.serial_cameras
    step_cams = [cams[(rank * cpr + j) % 8] for j in range(cpr)]

    def one_step(ev=None, kev=None, serial=False):
        """One bench step: cpr cameras on this rank (gradients accumulated), then the exchange."""
        if sequence:
            pipe.set_camera(*cams[step_no[0] % 8])
        step_no[0] += 1
        if cpr > 1 and not args.stage_kernels and not args.serial_cameras and not serial and ev is None \
                and kev is None:
            # several cameras per rank: two in flight, on several ranks every camera's all-gather behind its own
            # backward (opensplat_amd.pipeline.HotPath.step_cameras)
            pipe.step_cameras(step_cams)
            return
        for j in range(cpr):
            if cpr > 1:
                pipe.set_camera(*cams[(rank * cpr + j) % 8])
            last = j == cpr - 1
            pipe.step(ev if last else None, kev if last else None, accumulate=j > 0, exchange=last, slot=j)

    if world > 1:
        # who runs where, over what: printed (stderr) BEFORE anything is timed, so that a hang or a
        # mis-mapped device is visible in the log of a multi-GPU run
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            ver = "unknown"
        info = [None] * world
        torch.distributed.all_gather_object(info, {"rank": rank, "local_rank": local, "device": str(dev),
                                                   "name": torch.cuda.get_device_name(dev),
                                                   "pid": os.getpid()})
        if rank == 0:
            print("bench.py: backend=%s RCCL=%s world=%d exchange=%s cameras_per_rank=%d"
                  % (backend, ver, world, "factored" if factored else "flat", cpr), file=sys.stderr)
            for i in info:
                print("bench.py:   rank %(rank)d -> %(device)s (%(name)s), local_rank %(local_rank)d, pid %(pid)d" % i,
                      file=sys.stderr)
            sys.stderr.flush()
    # ---- warm-up (untimed): the W steps asked for, then more until the clocks and caches have settled —
    # a 1080p step is 0.8 ms, five of them are over before the GPU has left its idle clock (the driver's
    # `--steps 20 --warmup W` line of round 3 read 6 % below a 50-step run for that reason alone)
    def timed_block(n, serial=False):
        """n plain steps (no events, no hooks) between barrier + synchronize on both sides; seconds (max
        over ranks)."""
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            one_step(serial=serial)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    warm_t0 = time.perf_counter()
    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    warmup_run = args.warmup
    settle = float(os.environ.get("GSPLAT_BENCH_SETTLE_S", "0.3"))
    while True:
        # (every rank takes the same decision: the exchange inside one_step is collective)
        go = torch.tensor([1.0 if (time.perf_counter() - warm_t0 < settle and warmup_run < 4000) else 0.0],
                          device=dev)
        if world > 1:
            torch.distributed.all_reduce(go, op=torch.distributed.ReduceOp.MAX)
        if float(go.item()) == 0.0:
            break
        for _ in range(16):
            one_step()
        torch.cuda.synchronize()
        warmup_run += 16
    misses_warmup = pipe.misses

    # ---- the timed region: EXACTLY --steps plain steps.  Nothing is recorded inside it: no HIP events, no
    # hooks (round 3 sampled every (steps // 8)-th step with ~11 event records of ~5 us each — with the
    # driver's --steps 20 that was every other step) ----
    elapsed = timed_block(args.steps)
    misses_timed = pipe.misses - misses_warmup

    # ---- one GPU, c >= 2 cameras per step: the serial camera loop timed beside the two-in-flight headline ----
    cameras_ab = None
    if two_in_flight:
        for _ in range(4):
            one_step(serial=True)
        dt_serial = timed_block(args.steps, serial=True)
        dt_pipe2 = timed_block(args.steps)       # (again, after the serial block: same clock state)
        cameras_ab = {"cameras_per_step": cpr,
                      "serial_rasterizations_per_s": cpr * args.steps / dt_serial,
                      "two_in_flight_rasterizations_per_s": cpr * args.steps / dt_pipe2,
                      "speedup": dt_serial / dt_pipe2,
                      "note": "same cameras, same buffers, %d steps each; two_in_flight: camera j+1's per-Gaussian "
                              "forward + binning + compositing forward on a second stream under camera j's "
                              "compositing backward (opensplat_amd.pipeline.HotPath.step_cameras)" % args.steps}

    # ---- a second, longer plain block (>= 200 steps and >= 0.25 s), reported beside the headline: what the
    # path sustains once a run is long enough for the clocks to stop moving ----
    sustained = None
    if world == 1 or os.environ.get("GSPLAT_BENCH_SUSTAINED"):
        n_sus = max(200, int(0.25 / max(elapsed / args.steps, 1e-6)) + 1)
        n_sus = min(n_sus, 20000)
        dt_sus = timed_block(n_sus)
        sustained = {"steps": n_sus, "ms_per_step": dt_sus / n_sus * 1e3,
                     "value": world * cpr * n_sus / dt_sus,
                     "note": "plain steps like the timed region, more of them; not the headline"}

    # ---- the instrumented pass, OUTSIDE the timed region: HIP events at the stage boundaries and — through
    # the library's kernel timeline (gs_debug_timeline) — around EVERY kernel launch of the step ----
    n_inst = 8
    all_events, timelines = [], []
    for _ in range(n_inst):
        ev = []
        cabi.timeline(True)
        one_step(ev)
        timelines.append(cabi.timeline_read())
        cabi.timeline(False)
        all_events.append(ev)
    torch.cuda.synchronize()

    # per-stage durations from HIP events recorded on the launch stream
    stage_ms = {n: 0.0 for n in pipe.stage_names}
    for ev in all_events:
        for k, name in enumerate(pipe.stage_names):
            stage_ms[name] += ev[k].elapsed_time(ev[k + 1])
    stage_ms = {k: v / max(len(all_events), 1) for k, v in stage_ms.items()}
    # every kernel of the step, by name: mean ms per step and launches per step
    kernel_live = {}
    for tl in timelines:
        for name, ms in tl:
            e = kernel_live.setdefault(cabi.kernel_short_name(name), [0.0, 0])
            e[0] += ms
            e[1] += 1
    kernel_live = {k: {"ms": v[0] / n_inst, "launches_per_step": v[1] / n_inst} for k, v in kernel_live.items()}
    # the two compositing kernels alone (whatever template instance ran)
    kernel_ms = {}
    for short in ("k_rasterize_forward", "k_rasterize_backward"):
        kernel_ms[short] = sum(v["ms"] for k, v in kernel_live.items() if k.startswith(short))

    exchange_ab = None
    if world > 1 and not args.stage_kernels:
        try:
            exchange_ab = exchange_probe(pipe, world, dev)     # collective: every rank takes part
        except Exception as e:
            exchange_ab = {"error": repr(e)}

    cameras_c12 = None
    comm = None
    if world > 1 and not args.stage_kernels:
        try:
            cameras_c12 = cameras_probe(scene, dev, flags, factored, world, rank, cams)   # collective
        except Exception as e:
            cameras_c12 = {"error": repr(e)}
        try:   # what the communicator is made of: library version, ranks, devices
            ver = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
        except Exception:
            ver = None
        names = [None] * world
        torch.distributed.all_gather_object(names, "%s:%d" % (torch.cuda.get_device_name(dev), local))
        comm = {"backend": backend, "rccl_version": ver, "ranks": world, "devices": names,
                "NCCL_DEBUG": os.environ.get("NCCL_DEBUG")}

    if rank == 0:
        N, K, M, P = scene.N, scene.K, pipe.num_isects, scene.W * scene.H
        total_bytes, per_stage = algorithmic_bytes(N, K, M, P)
        # dominant kernel = the longer of the two compositing kernels (every other kernel of the
        # path is >= 5x shorter, see stage_ms / profiles/)
        dom = max(kernel_ms, key=kernel_ms.get)
        dom_ms = kernel_ms[dom]
        dom_bytes = per_stage["rasterize_bwd" if dom == "k_rasterize_backward" else "rasterize_fwd"]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        traffic = None
        # the PMC passes are collected on the default command (C2 / C4 per-rank workload, parity exp)
        # and on --config c3; other workloads carry no measured traffic
        plain = (not args.fast_exp and not args.hot and args.order == "given" and not args.stage_kernels
                 and not sequence and cpr == 1)
        tname = None
        if plain and scene.N == 1_000_000 and scene.W == 1920 and scene.H == 1080:
            tname = "traffic.json"
        elif plain and scene.N == 5_000_000 and scene.W == 3840 and scene.H == 2160:
            tname = "traffic_c3.json"
        if tname and os.path.exists(os.path.join(ROOT, "profiles", tname)):
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", tname)))
                # (the file is keyed by kernel name: k_rasterize_backward_q since round 5)
                traffic = tj.get(dom)
                if traffic is None:
                    traffic = next((v for k, v in sorted(tj.items()) if k.startswith(dom)), None)
            except Exception:
                traffic = None
        ms_per_step = elapsed / args.steps * 1e3
        kernels, kernels_profiled, roofline_valu = kernel_tables(
            N, K, M, P, stage_ms, kernel_ms, kernel_live,
            None if not tname else ("kernels.json" if tname == "traffic.json" else "kernels_c3.json"))
        out = {
            "metric": "forward+backward rasterizations/sec at 1M Gaussians 1080p",
            "value": world * cpr * args.steps / elapsed,
            "unit": "rasterizations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "gaussians": N, "width": scene.W, "height": scene.H,
                       "sh_bases": K, "tile": 16, "intersections_M": M,
                       "exp": ("hardware v_exp_f32 — NOT the parity mode: at C2 1 pixel of 2 073 600 changes "
                               "its contributor set (1.4e-3), the others stay within 3.6e-7, gradients "
                               "within 1.1e-3 of max|g| instead of 8e-7 (tests/test_gpu_parity_r03.py, "
                               "profiles/parity_r03.json)") if args.fast_exp
                       else "glibc-bit-exact expf (parity mode)",
                       "per_gaussian_stages": "separate kernels" if args.stage_kernels else
                       "fused (gs_gaussian_forward / gs_gaussian_backward)",
                       "parallelism": "camera-per-rank dp%d" % world,
                       "cameras_per_rank_per_step": cpr},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": None if traffic is None else
                         "replayed from profiles/%s (rocprofv3 PMC pass of the same command)" % tname,
                         "kernel_ms": dom_ms, "algorithmic_bytes": dom_bytes,
                         "note": "achieved: HIP events around the kernel launch itself, THIS run; the compositing kernels "
                                 "are VALU-issue-bound, not HBM-bound (no dense contraction, no MFMA; "
                                 "SURVEY.md §8d, DESIGN.md §4) — the HBM fraction is reported as "
                                 "required; traffic = rocprofv3 FETCH_SIZE x 1.18 (factor calibrated on "
                                 "48-byte record gathers, profiles/calib_r02.json) + WRITE_SIZE per "
                                 "launch from profiles/"},
            "roofline_valu": roofline_valu,
            "kernels": kernels,
            "kernels_profiled": kernels_profiled,
            "kernel_ms": kernel_ms,
            # pixel x Gaussian evaluations per second of the compositing kernels, counted
            # as 256 pixels per (tile, Gaussian) list entry (the upper bound both kernels are sized by)
            "pixel_gaussian_evals_per_s": {k: 256.0 * M / (v * 1e-3) for k, v in kernel_ms.items() if v > 0},
            "warmup_steps_run": warmup_run,
            # said where the number is quoted: what --warmup became, and how short the timed region is
            "timing": {"timed_region_ms": elapsed * 1e3, "warmup_steps_asked": args.warmup,
                       "warmup_steps_run": warmup_run,
                       "note": "--warmup W is extended until the clocks have settled (>= %.1f s of steps, in blocks of "
                               "16): W = %d became %d untimed steps.  The timed region is EXACTLY --steps = %d plain "
                               "steps = %.1f ms of GPU time between barrier + synchronize pairs — shorter than a "
                               "utilisation sampler's period; `sustained` (a longer plain block, same code) is the "
                               "cross-check" % (settle, args.warmup, warmup_run, args.steps, elapsed * 1e3)},
            "sustained": sustained,
            "instrumented_steps": n_inst,
            # which fields were MEASURED IN THIS RUN and which are read back from committed files
            "provenance": {
                "live": ["value", "ms_per_step", "sustained", "stage_ms", "kernel_ms", "kernels",
                         "roofline.achieved", "roofline.kernel_ms", "path_roofline", "speculative_binning",
                         "moving_camera", "cpu_baseline", "exchange_ms"],
                "replayed": ["roofline.traffic", "roofline_valu", "kernels_profiled"],
                "note": "live: timed region = exactly --steps plain steps; stage / kernel durations from an "
                        "instrumented pass of %d steps outside it (HIP events on the launch stream, around "
                        "every kernel launch).  replayed: rocprofv3 PMC / kernel-trace results of the same "
                        "command, committed under profiles/ — NOT measured in this run" % n_inst},
            "path_roofline": {"algorithmic_bytes": total_bytes,
                              "achieved_GBs": total_bytes / (ms_per_step * 1e-3) / 1e9,
                              "frac": total_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "stage_ms": stage_ms,
            # the gradient exchange: what is all-reduced / all-gathered per step and the xGMI bytes a
            # rank sends (ring all-reduce 2 (W-1)/W x S, all-gather (W-1) x message)
            "exchange": ({"mode": "none"} if world == 1 else
                         {"mode": "factored", "allreduce_bytes": 11 * N * 4,
                          "allgather_message_bytes": pipe.fx.chunk * 4,
                          "bytes_moved_per_rank": pipe.fx.bytes_moved_per_rank} if pipe.fx is not None else
                         {"mode": "flat", "allreduce_bytes": pipe.grads.nbytes,
                          "bytes_moved_per_rank": int(2 * (world - 1) / world * pipe.grads.nbytes)}),
            "grad_bytes_allreduced": (0 if world == 1 else 11 * N * 4 if pipe.fx is not None
                                      else pipe.grads.nbytes),
            # both exchanges timed alone in this run (N > 1): flat = north_star's single all-reduce
            "exchange_ab": exchange_ab,
            # --gpus 1 --cameras-per-rank c >= 2: serial camera loop against two cameras in flight, this run
            "cameras_ab": cameras_ab,
            # N > 1: one against two cameras per rank and exchange, this run; the communicator
            "cameras_c1_c2": cameras_c12,
            "communicator": comm,
            # per-link xGMI bytes of the exchange as launched: a ring over W ranks puts bytes_moved_per_rank on
            # each rank's outgoing link (RCCL rings on an 8-GPU MI355X node use one xGMI link per neighbour)
            "per_link_bytes_per_step": (None if world == 1 else
                                        pipe.fx.bytes_moved_per_rank if pipe.fx is not None else
                                        int(2 * (world - 1) / world * pipe.grads.nbytes)),
            "allreduce_ms_rank0": stage_ms.get("allreduce", 0.0),
            # the whole exchange (all-reduce [+ all-gather + SH backward over the gathered cameras]) as
            # timed by HIP events on rank 0, apart from ms_per_step
            "exchange_ms": stage_ms.get("allreduce", 0.0) if world > 1 else 0.0,
            # speculative binning: forwards repeated because the id list (sized from the previous
            # call, +12.5 %) was too small — during warm-up / inside the timed region
            "speculative_binning": {"misses_warmup": misses_warmup,
                                    "misses_timed": misses_timed,
                                    "id_list_capacity": pipe.ws.capacity},
        }
        if world == 1 and plain and not sequence and not args.no_cpu_baseline:   # (profiling passes skip both)
            try:
                out["speculative_binning"]["operator_path"] = operator_binning_probe(scene, dev)
            except Exception as e:
                out["speculative_binning"]["operator_path"] = {"error": repr(e)}
        if world == 1 and plain and cfg == "c2" and not args.no_cpu_baseline:
            try:
                out["moving_camera"] = moving_camera_probe(pipe, scene, cams)
            except Exception as e:
                out["moving_camera"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(scene, args.cpu_gaussians)
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
