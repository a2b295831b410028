"""One optimiser step of OpenSplat's training loop on the MI355X kernels.

Host-side mirror of the reference's iteration (opensplat.cpp:151-170):

    model.optimizersZeroGrad()                      -> nothing to do: every gradient is overwritten
    rgb = model.forward(cam, step)                  -> projection + SH + binning + compositing with the
                                                       Model::forward glue fused in (row f1)
    mainLoss = model.mainLoss(rgb, gt, ssimWeight)  -> gs_main_loss: value AND d loss / d rgb in one call
    mainLoss.backward()                             -> compositing / SH / projection backward
    [multi-camera batch: one camera per rank]       -> sum all-reduce of the flat gradient buffer (RCCL)
    model.optimizersStep()                          -> gs_adam_step: the six groups in one launch
    model.schedulersStep(step)                      -> gs_sched_lr for the means (model.cpp:68,245-247)

Everything goes through the C ABI (opensplat_amd/cabi.py); torch only owns the device buffers and
the process group.  Parameters, gradients and both Adam moments live in four flat buffers with the
same layout (dist.GradBuffer: [features_rest | features_dc | means | scales | quats | opacities]),
so the Adam groups are plain slices and the all-reduce needs no gather copy.

A batch of B cameras = B ranks; the step's loss is the mean of the per-camera losses (each rank
scales its cotangent by 1/B, the all-reduce sums).

Trainer.after_train mirrors Model::afterTrain (model.cpp:311-494, row f4): per-iteration
statistics, and every `refine_every` steps the split / duplicate / cull refinement with the
optimiser-state surgery and the alpha reset, all on the device (include/gsplat_densify.h).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import cabi, dist


def rank_merged_config(cfg: "cabi.GsDensifyConfig", world: int) -> "cabi.GsDensifyConfig":
    """With one camera per rank every rank scales its loss cotangent by 1/world (train_step), so the
    per-camera d loss / d xys that feeds the densification statistics is 1/world of what a
    single-camera iteration sees, while --densify-grad-thresh (0.0002, model.cpp:347) is calibrated
    for unscaled single-camera gradients.  The statistics are linear in the gradient norm and only
    enter the plan through xysGradNorm / visCounts * half_max_side, so scaling half_max_side by
    `world` undoes the batch scaling exactly (gsplat's strategy does the same with
    `grads *= n_cameras`)."""
    cfg.half_max_side = float(cfg.half_max_side) * float(world)
    return cfg


def morton_permutation(means: torch.Tensor) -> torch.Tensor:
    """Indices that sort the rows of `means` [N, 3] along a 3-D Morton (Z-order) curve, 10 bits per
    axis over the bounding box.  Any permutation of the Gaussians renders the same image."""
    lo, hi = means.min(0).values, means.max(0).values
    q = ((means - lo) / (hi - lo).clamp_min(1e-12) * 1023.0).to(torch.int64).clamp_(0, 1023)

    def spread(x):
        x = (x | (x << 16)) & 0x030000FF
        x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3
        x = (x | (x << 2)) & 0x09249249
        return x
    key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return torch.argsort(key, stable=True)


class Trainer:
    # model.cpp:61-66, 68 (means decay to lr / 100 over max_steps)
    LR = dict(means=0.00016, scales=0.005, quats=0.001, features_dc=0.0025,
              features_rest=0.000125, opacities=0.05)
    MEANS_LR_FINAL = 0.0000016

    def __init__(self, means, log_scales, quats, opacity_logits, features_dc, features_rest,
                 device, max_steps: int = 30000, ssim_weight: float = 0.2, refine_every: int = 100,
                 warmup_length: int = 500, reset_alpha_every: int = 30,
                 densify_grad_thresh: float = 0.0002, densify_size_thresh: float = 0.01,
                 stop_screen_size_at: int = 4000, split_screen_size: float = 0.05,
                 num_cameras: int = 1, morton_order: bool = False, num_downscales: int = 2,
                 resolution_schedule: int = 3000, sh_degree_interval: int = 1000,
                 reference_alpha_reset: bool = False, grad_buckets: int = 4, exchange: str = "auto",
                 deterministic: bool = False, segmented: bool = True, experimental_graph: bool = False):
        """Parameters as Model holds them (model.hpp): means [N,3], log-scales [N,3], raw quats
        [N,4], opacity logits [N] or [N,1], featuresDc [N,3], featuresRest [N,K-1,3].
        segmented: on frames of few tiles the compositing backward runs the pieces of a tile's list side by
        side from checkpoints the forward leaves (cabi.Checkpoints; scheduling only).
        experimental_graph: NOT part of the supported surface (round 6: taken out of it).  Replays the iteration
        as one captured HIP graph on a stream of its own; pays only below ~1500 Gaussians; a GPU write fault seen
        in round 4 when replays shared a stream with eager launches was worked around (own stream), never
        explained (profiles/HISTORY.md).  The building blocks it is made of — gs_adam_step_scheduled,
        gs_stage_f32, gs_copy_indirect_f32 — are supported and tested eagerly (tests/test_gpu_train_blocks.py)."""
        graph = experimental_graph
        self.segmented, self._ckpt = segmented, cabi.Checkpoints()
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a,
                                      dtype=torch.float32).to(device)
        N = means.shape[0]
        K = 1 + (int(features_rest.shape[1]) if features_rest is not None else 0)
        # (an index-less 'cuda' would compare unequal to the tensors' 'cuda:0')
        device = torch.empty(0, device=device).device
        self.N, self.K, self.dev = N, K, device
        self.max_steps, self.ssim_weight = max_steps, ssim_weight
        self.params = dist.GradBuffer(N, K, device)   # same layout, holds the parameters
        self.grads = dist.GradBuffer(N, K, device)
        self.exp_avg = dist.GradBuffer(N, K, device)
        self.exp_avg_sq = dist.GradBuffer(N, K, device)
        P = self.params
        P.v_means.copy_(t(means)); P.v_scales.copy_(t(log_scales)); P.v_quats.copy_(t(quats))
        P.v_opacity.copy_(t(opacity_logits).reshape(-1)); P.v_dc.copy_(t(features_dc))
        if K > 1:
            P.v_rest.copy_(t(features_rest))
        # densification schedule: the CLI defaults of opensplat.cpp:37-43, model.hpp:30
        self.refine_every, self.warmup_length = refine_every, warmup_length
        self.reset_alpha_every, self.stop_split_at = reset_alpha_every, max_steps // 2
        self.densify_grad_thresh, self.densify_size_thresh = densify_grad_thresh, densify_size_thresh
        self.stop_screen_size_at, self.split_screen_size = stop_screen_size_at, split_screen_size
        self.num_cameras = num_cameras
        self.num_downscales, self.resolution_schedule = num_downscales, resolution_schedule
        self.sh_degree_interval = sh_degree_interval
        # re-sort the Gaussians along a 3-D Morton curve whenever a refinement rebuilds the tensors:
        # neighbours in space become neighbours in memory (binning scatter and the per-Gaussian
        # kernels gain locality: +1..3 % per iteration at 1 M Gaussians)
        self.morton_order = morton_order
        # Alpha reset (model.cpp:464-479).  The reference builds a zeroed AdamParamState but never
        # installs it, and re-binds `opacities` to a tensor the optimiser does not know: until the
        # next refinement re-registers it (addToOptimizer / removeFromOptimizer, :253-309) neither
        # the opacities nor their moments are updated.  False (default): the evident intent —
        # opacities stay trainable, their moments are zeroed.  True: the reference's actual
        # behaviour step for step (KNOWN PARITY DEVIATION switch).
        self.reference_alpha_reset = reference_alpha_reset
        # gradient exchange of a camera batch (one camera per rank): the flat buffer goes out as
        # `grad_buckets` collectives and the Adam step of a bucket's parameters runs as soon as ITS
        # collective has finished, while the next bucket is still on the wire
        self.grad_buckets = max(1, int(grad_buckets))
        self.bucket_single_rank = False   # tests: take the bucketed path without a process group
        self._pending = None
        self._opacity_frozen = False
        self._opacity_lag = 0        # optimiser steps the opacities' Adam state is behind (reference_alpha_reset)
        self._visible = True
        self._stats = None          # (xysGradNorm, visCounts, max2DSize); None = cleared
        self.step_count = 0
        self.means_lr = self.LR["means"]
        self._shape = None
        self.world = torch.distributed.get_world_size() if (
            torch.distributed.is_available() and torch.distributed.is_initialized()) else 1
        self.rank = torch.distributed.get_rank() if self.world > 1 else 0
        # "factored" (default with several ranks): all-gather the 12-byte colour cotangents and form the
        # SH gradients of all cameras locally, all-reduce the geometry block only; "flat": the whole
        # gradient buffer through bucketed all-reduces (dist.FactoredExchange)
        assert exchange in ("auto", "flat", "factored")
        self.factored = exchange == "factored" or (exchange == "auto" and 1 < self.world <= 32)
        self.fx = None
        # deterministic=True: the compositing backward sums in 64-bit fixed point (GS_FLAG_DETERMINISTIC):
        # bit-reproducible gradients, hence bit-reproducible training (tests compare whole runs)
        self.deterministic = bool(deterministic)
        # graph=True (one rank): train_step replays the whole iteration — per-Gaussian forward, binning,
        # compositing, loss, both backward kernels, Adam — as ONE captured HIP graph per (buffers, SH degree,
        # intrinsics, id-list capacity) on a stream of its own; see _train_step_graph.
        self.graph = bool(graph) and self.world == 1
        if self.graph and reference_alpha_reset:
            raise ValueError("experimental_graph=True keeps one optimiser step count for all groups: not with "
                             "reference_alpha_reset (the opacities' Adam state lags behind there)")
        self._graphs = {}
        self._buf_gen = 0            # bumped whenever a device buffer a captured graph points at is replaced
        self._g_rows = None          # device table of the per-step Adam scalars (gs_adam_step_scheduled)
        self._g_rows_first = 0       # optimiser step of row 0
        self._g_stream = None        # the replays' own stream: nothing else is ever enqueued on it
        self.graph_stats = dict(captures=0, replays=0, eager=0, overflows=0)

    def degrees_to_use(self, step: int) -> int:
        """model.cpp:178: one more SH degree every sh_degree_interval steps."""
        sh_degree = {1: 0, 4: 1, 9: 2, 16: 3, 25: 4}[self.K]
        return min(step // self.sh_degree_interval, sh_degree)

    def downscale_factor(self, step: int) -> int:
        """Model::getDownscaleFactor (model.cpp:249-251): training starts on images reduced by
        2^num_downscales and doubles the resolution every resolution_schedule steps."""
        return 2 ** max(self.num_downscales - step // self.resolution_schedule, 0)

    # the six tensors, as views of the flat parameter buffer
    means = property(lambda s: s.params.v_means)
    log_scales = property(lambda s: s.params.v_scales)
    quats = property(lambda s: s.params.v_quats)
    opacity_logits = property(lambda s: s.params.v_opacity)
    features_dc = property(lambda s: s.params.v_dc)
    features_rest = property(lambda s: s.params.v_rest)

    def _buffers(self, W, H):
        if self._shape == (W, H):
            return
        N, dev = self.N, self.dev
        f = dict(device=dev, dtype=torch.float32)
        i = dict(device=dev, dtype=torch.int32)
        # per-Gaussian outputs of gs_gaussian_forward (the 2-D intermediates stay in registers)
        self.proj = dict(packed=torch.empty((N, cabi.GS_SPLAT_DWORDS), **f), depths=torch.empty((N,), **f),
                         radii=torch.empty((N,), **i), rgb_raw=torch.empty((N, 3), **f), xys=None)
        old_ws = getattr(self, "bin_ws", None)
        self.bin_ws = cabi.BinWorkspace()
        # (the id-list capacity learnt so far is a good first guess for the new buffers too)
        self.bin_ws.capacity = max(old_ws.capacity if old_ws is not None else 0, 1024)
        self.fwd = dict(img=torch.empty((H, W, 3), **f), final_Ts=torch.empty((H, W), **f),
                        final_idx=torch.empty((H, W), **i), img_clamped=torch.empty((H, W, 3), **f))
        # gradient records of the compositing backward, consumed in place by gs_gaussian_backward
        ws_bytes = (cabi.lib().gs_rasterize_backward_workspace_bytes_det(N) if self.deterministic
                    else cabi.lib().gs_rasterize_backward_workspace_bytes(N))
        self.bwd_ws = torch.zeros((ws_bytes + 64,), device=dev, dtype=torch.uint8)
        # (the records are zeroed by the binning's count pass on the way — gs_bin_speculative_zero — instead of by a fill
        # kernel in front of the compositing backward; not in deterministic mode, whose sums live elsewhere)
        self._rec_zero = None if (self.deterministic or ws_bytes % 16 or os.environ.get("GSPLAT_RECORDS_MEMSET") == "1") \
            else self.bwd_ws[:ws_bytes]
        self.v_xy = torch.zeros((N, 2), **f)     # d loss / d xys: the densification statistics' input
        self.rgrads = dict(v_xy=self.v_xy)
        self.gout = dict(v_means=self.grads.v_means, v_scales=self.grads.v_scales,
                         v_quats=self.grads.v_quats, v_opacity=self.grads.v_opacity,
                         v_dc=self.grads.v_dc, v_rest=self.grads.v_rest)
        self.loss_ws = torch.empty(cabi.lib().gs_loss_workspace_bytes(W, H), device=dev,
                                   dtype=torch.uint8)
        self.loss_out = (torch.empty(3, **f), torch.empty((H, W, 3), **f))
        self._shape = (W, H)
        self._buf_gen += 1
        self._graphs.clear()
        if self.graph:
            # what changes from step to step is written by the host into pinned memory and fetched by the
            # graph's own first nodes: [viewmat 16 | projmat 16 | camera centre 3 | -] and the ADDRESS of the
            # ground-truth image
            self._g_cam = torch.zeros(36, **f)
            self._g_cam_host = torch.zeros(36, dtype=torch.float32).pin_memory()
            self._g_cam_np = self._g_cam_host.numpy()
            self._g_gt = torch.empty((H, W, 3), **f)
            self._g_gt_ptr = torch.zeros(1, dtype=torch.int64).pin_memory()
            self._g_gt_ptr_np = self._g_gt_ptr.numpy()
            self._g_mhost_np = self.bin_ws.m_host.numpy()
            self._g_done = torch.cuda.Event()
            if self._g_stream is None:
                self._g_stream = torch.cuda.Stream()

    def render(self, cam: dict, background, degrees_to_use: int):
        """Model::forward (model.cpp:83-225) for one camera -> clamped rgb [H, W, 3]."""
        W, H = cam["W"], cam["H"]
        self._buffers(W, H)
        gcam = cabi.make_camera(cam["viewmat"], cam["projmat"], cam["fx"], cam["fy"], cam["cx"],
                                cam["cy"], W, H, flags=cabi.GS_CAM_LOG_SCALES)
        vm = np.asarray(cam["viewmat"], dtype=np.float32)
        cam_pos = (-vm[:3, :3].T @ vm[:3, 3]).astype(np.float32)   # model.cpp:95
        if self.factored:
            # the camera centre travels in the exchange message: keep a device copy per camera (the
            # upload is a synchronising copy — once per camera, not once per step)
            key = (vm.tobytes(), str(self.dev))
            hit = cam.get("_cam_pos_dev")
            if hit is None or hit[0] != key:
                hit = cam["_cam_pos_dev"] = (key, torch.from_numpy(cam_pos).to(self.dev))
            self._cam_pos_dev = hit[1]
        flags = cabi.GS_FLAG_LOGIT_OPACITY | cabi.GS_FLAG_CLAMP_IMAGE
        # a frame of few tiles with long lists: the forward leaves checkpoints along the lists and the
        # backward runs their pieces side by side (planned from the previous frame's list statistics)
        ck = self._ckpt if self.segmented and self._ckpt.plan(W, H, self.bin_ws.list_stats, self.dev) else None
        while True:
            p = cabi.gaussian_forward(gcam, self.means, self.log_scales, self.quats,
                                      self.opacity_logits, self.features_dc,
                                      self.features_rest if self.K > 1 else None, cam_pos,
                                      degrees_to_use, flags, out=self.proj)
            b = cabi.bin_and_sort(W, H, None, p["depths"], None, None, None, None, None, self.bin_ws,
                                  speculative=True, packed=p["packed"], zero=self._rec_zero)
            f = cabi.rasterize_forward(W, H, b, background, flags, out=self.fwd, checkpoints=ck)
            if cabi.validate_binning(b):   # id-list capacity guess was large enough
                break
        # the plan as THIS forward used it (a later render() re-plans the shared object).  backward() differentiates
        # the LAST render(): the context, the plan and the records are replaced together (self._ctx)
        f["checkpoints"] = ck.frozen() if ck is not None else None
        # no visible Gaussian: Model::forward returns the bare background (model.cpp:173), xys gets
        # no gradient and afterTrain returns at once (model.cpp:315)
        self._visible = b.num_isects > 0
        self._records_clean = getattr(b, "zeroed", False)
        self._ctx = (gcam, cam_pos, p, p["rgb_raw"], b, f, flags, degrees_to_use, background, W, H)
        return f["img_clamped"]

    def backward(self, v_rgb):
        """d loss / d parameters into self.grads (overwritten), from d loss / d (clamped rgb)."""
        gcam, cam_pos, p, rgb_raw, b, f, flags, deg, background, W, H = self._ctx
        keep = cabi.GS_FLAG_KEEP_RECORDS | (cabi.GS_FLAG_DETERMINISTIC if self.deterministic else 0)
        if self._rec_zero is not None and self._records_clean:
            keep |= cabi.GS_FLAG_RECORDS_ZEROED      # (render()'s binning zeroed them; a second backward() of the same
        self._records_clean = False                  #  render lets the library fill them as before)
        cabi.rasterize_backward(W, H, self.N, b, background, f["final_Ts"], f["final_idx"], v_rgb,
                                flags | keep, workspace=self.bwd_ws, img_raw=f["img"],
                                checkpoints=f.get("checkpoints"))
        if self.factored:
            if self.fx is None or self.fx.N != self.N:      # (a refinement changes N)
                self.fx = dist.FactoredExchange(self.N, self.K, 1, self.dev)
            fx = self.fx
            fx.set_cam_pos(0, self._cam_pos_dev)      # device-to-device: no host synchronisation
            cabi.gaussian_backward(gcam, self.means, self.log_scales, self.quats, self.opacity_logits,
                                   cam_pos, self.K, deg, p["radii"], rgb_raw, self.bwd_ws,
                                   dict(self.gout, v_dc=fx.v_color(0), v_rest=None),
                                   flags | cabi.GS_FLAG_EMIT_VCOLOR, v_xy=self.v_xy)
            fx.start(self.grads)
            sh = self.grads.sh_numel
            # (lo, hi, wait-and-prepare): the SH block first — its Adam step overlaps the geometry all-reduce
            self._pending = [(0, sh, lambda: fx.finish_sh(self.grads, self.means, deg)),
                             (sh, self.grads.flat.numel(), fx.finish_geometry)]
            return
        cabi.gaussian_backward(gcam, self.means, self.log_scales, self.quats, self.opacity_logits,
                               cam_pos, self.K, deg, p["radii"], rgb_raw, self.bwd_ws, self.gout, flags,
                               v_xy=self.v_xy)
        # start the exchange; optimizer_step() consumes it bucket by bucket
        self._pending = [(lo, hi, (lambda w=w: dist.wait_all(w)))
                         for lo, hi, w in dist.allreduce_buckets_async(self.grads, self.grad_buckets)] \
            if (self.world > 1 or self.bucket_single_rank) else None

    def adam_groups(self, lo: int = 0, hi: int | None = None):
        """The six Adam groups (model.cpp:61-66) as slices of the flat buffers, restricted to the
        element range [lo, hi) of the flat layout (a gradient bucket)."""
        P, G, M, V = self.params, self.grads, self.exp_avg, self.exp_avg_sq
        hi = P.flat.numel() if hi is None else hi
        lr = dict(self.LR, means=self.means_lr)
        names = [("v_rest", "features_rest"), ("v_dc", "features_dc"), ("v_means", "means"),
                 ("v_scales", "scales"), ("v_quats", "quats"), ("v_opacity", "opacities")]   # flat order
        out, o = [], 0
        for v, n in names:
            cnt = P.views[v].numel()
            a, b = max(lo, o), min(hi, o + cnt)
            if a < b and not (n == "opacities" and self._opacity_frozen):
                out.append((P.flat[a:b], G.flat[a:b], M.flat[a:b], V.flat[a:b], lr[n],
                            self._opacity_lag if n == "opacities" else 0))
            o += cnt
        return out

    def _adam(self, groups):
        """gs_adam_step takes ONE step count per launch: groups are launched by step count (one launch,
        unless the reference's alpha reset left the opacities' optimiser behind: reference_alpha_reset)."""
        for lag in sorted({g[5] for g in groups}):
            cabi.adam_step([g[:5] for g in groups if g[5] == lag], self.step_count - lag)

    def optimizer_step(self):
        """Model::optimizersStep + schedulersStep (model.cpp:236-247)."""
        self.step_count += 1
        if self._opacity_frozen:
            # torch::optim::Adam skips a parameter without gradient: its step count stays behind for
            # good (the re-registration at the next refinement copies it, model.cpp:253-309)
            self._opacity_lag += 1
        if self._pending is None:
            self._adam(self.adam_groups())
        else:
            for lo, hi, ready in self._pending:     # Adam of bucket k overlaps the transfer of k + 1
                ready()
                groups = self.adam_groups(lo, hi)
                if groups:
                    self._adam(groups)
            self._pending = None
        # OptimScheduler::step(step) sets the lr the NEXT optimiser step uses (opensplat.cpp:168-169)
        self.means_lr = cabi.sched_lr(self.LR["means"], self.MEANS_LR_FINAL, self.max_steps,
                                      self.step_count)

    # ---- the iteration as one captured HIP graph (graph=True) ---------------------------------------
    ADAM_ROWS = 2048

    class PreparedCamera:
        """What a replayed iteration needs of a camera, computed once: the 36-float block the graph's first
        node fetches ([viewmat | projmat | centre | -]), the intrinsics (kernel ARGUMENTS, hence part of the
        capture key) and the host-side GsCamera of the launches."""

        def __init__(self, cam: dict):
            vm = np.asarray(cam["viewmat"], dtype=np.float32)
            self.W, self.H = int(cam["W"]), int(cam["H"])
            self.block = np.zeros(36, np.float32)
            self.block[0:16] = vm.reshape(-1)
            self.block[16:32] = np.asarray(cam["projmat"], dtype=np.float32).reshape(-1)
            self.block[32:35] = -vm[:3, :3].T @ vm[:3, 3]                      # model.cpp:95
            self.cam_pos = self.block[32:35].copy()
            self.intr = (float(cam["fx"]), float(cam["fy"]), float(cam["cx"]), float(cam["cy"]), self.W, self.H)
            # (the matrices the struct carries are ignored in favour of the device copies)
            eye = np.eye(4, dtype=np.float32)
            self.gcam = cabi.make_camera(eye, eye, cam["fx"], cam["fy"], cam["cx"], cam["cy"], self.W, self.H,
                                         flags=cabi.GS_CAM_LOG_SCALES)

    def prepare_camera(self, cam) -> "Trainer.PreparedCamera":
        return cam if isinstance(cam, Trainer.PreparedCamera) else Trainer.PreparedCamera(cam)

    def _adam_lrs(self, step: int):
        """Learning rates of optimiser step `step` (1-based), in adam_groups() order: the means' follows
        the schedule — optimizer_step() sets it from the number of steps already taken (model.cpp:245-247)."""
        lr = dict(self.LR, means=cabi.sched_lr(self.LR["means"], self.MEANS_LR_FINAL, self.max_steps, step - 1)
                  if step > 1 else self.LR["means"])
        order = ["features_rest", "features_dc", "means", "scales", "quats", "opacities"]
        if self.K == 1:
            order = order[1:]
        return [lr[n] for n in order]

    def _ensure_adam_rows(self):
        """The device table holds the scalars of steps [first, first + ADAM_ROWS); row index (device) = steps
        taken since `first`.  Refilled — between iterations — when the next step runs off it."""
        nxt = self.step_count + 1
        if self._g_rows is not None and self._g_rows_first <= nxt < self._g_rows_first + self.ADAM_ROWS:
            return
        rows = cabi.adam_schedule_rows([self._adam_lrs(nxt + r) for r in range(self.ADAM_ROWS)], nxt)
        if self._g_rows is None:
            self._g_rows = torch.empty((self.ADAM_ROWS, cabi.GS_ADAM_ROW_FLOATS), device=self.dev,
                                       dtype=torch.float32)
            self._g_row_index = torch.zeros(1, device=self.dev, dtype=torch.int32)
        self._g_rows.copy_(torch.from_numpy(rows))       # (synchronising copy: once per ADAM_ROWS steps)
        self._g_row_index.zero_()
        torch.cuda.synchronize()
        self._g_rows_first = nxt

    def _iteration_launches(self, gcam, deg, background, W, H):
        """Every launch of one training iteration, on the current stream, reading the camera and the target
        through the pinned words: what the graph captures and what a first (or repeated) step runs
        eagerly.  The Adam step is guarded by the intersection count the scan left on the device."""
        flags = cabi.GS_FLAG_LOGIT_OPACITY | cabi.GS_FLAG_CLAMP_IMAGE
        c = self._g_cam
        cabi.stage_f32(c, self._g_cam_host, 36)
        cabi.copy_indirect_f32(self._g_gt, self._g_gt_ptr, H * W * 3)
        vm, pm, pos = c[0:16], c[16:32], c[32:35]
        p = cabi.gaussian_forward(gcam, self.means, self.log_scales, self.quats, self.opacity_logits,
                                  self.features_dc, self.features_rest if self.K > 1 else None, pos, deg,
                                  flags, out=self.proj, viewmat_dev=vm, projmat_dev=pm)
        b = cabi.bin_and_sort(W, H, None, p["depths"], None, None, None, None, None, self.bin_ws,
                              speculative=True, packed=p["packed"])
        f = cabi.rasterize_forward(W, H, b, background, flags, out=self.fwd)
        loss, v_rgb = cabi.main_loss(f["img_clamped"], self._g_gt, self.ssim_weight, 1.0, True,
                                     out=self.loss_out, workspace=self.loss_ws)
        cabi.rasterize_backward(W, H, self.N, b, background, f["final_Ts"], f["final_idx"], v_rgb,
                                flags | cabi.GS_FLAG_KEEP_RECORDS |
                                (cabi.GS_FLAG_DETERMINISTIC if self.deterministic else 0),
                                workspace=self.bwd_ws, img_raw=f["img"])
        cabi.gaussian_backward(gcam, self.means, self.log_scales, self.quats, self.opacity_logits, pos,
                               self.K, deg, p["radii"], p["rgb_raw"], self.bwd_ws, self.gout, flags,
                               v_xy=self.v_xy, viewmat_dev=vm, projmat_dev=pm)
        off = cabi.lib().gs_bin_num_isects_offset(W, H)
        guard = self.bin_ws.bufs["ws"][off:off + 4].view(torch.int32)
        groups = [g[:5] for g in self.adam_groups()]
        cabi.adam_step_scheduled(groups, self._g_rows, self._g_row_index, guard, b.capacity)
        cabi.adam_advance(self._g_row_index, guard, b.capacity)
        return p, b, f, loss, flags

    def _train_step_graph(self, cam, gt, background, degrees_to_use: int):
        pc = self.prepare_camera(cam)
        W, H = pc.W, pc.H
        self._buffers(W, H)
        self._ensure_adam_rows()
        # (the previous iteration has completed — we waited for it below — so the pinned words are free)
        self._g_cam_np[:] = pc.block
        # the captured copy kernel reads H * W * 3 floats through this raw pointer: the tensor must be exactly that
        # (a uint8 or strided target would be read as garbage or out of bounds)
        if not (isinstance(gt, torch.Tensor) and gt.is_cuda and gt.device == torch.device(self.dev)
                and gt.dtype == torch.float32 and gt.is_contiguous() and gt.numel() == H * W * 3):
            raise ValueError("Trainer(experimental_graph=True): gt must be a contiguous float32 [H, W, 3] tensor on %s, got %s"
                             % (self.dev, (tuple(gt.shape), gt.dtype, gt.device) if isinstance(gt, torch.Tensor) else type(gt)))
        self._g_gt_ptr_np[0] = gt.data_ptr()
        self._g_gt_ref = gt          # (alive until the iteration that reads it has completed)
        bgk = background.tobytes() if isinstance(background, np.ndarray) else tuple(float(x) for x in background)
        mh = self._g_mhost_np
        while True:
            key = (self._buf_gen, degrees_to_use, pc.intr, bgk, self.bin_ws.capacity)
            hit = self._graphs.get(key)
            if hit is None:
                # first iteration under this key: run it launch by launch on the caller's stream; the same
                # launches are captured for the following ones (a capture executes nothing)
                p, b, f, loss, flags = self._iteration_launches(pc.gcam, degrees_to_use, background, W, H)
                self.graph_stats["eager"] += 1
                self._g_done.record()
            else:
                self._graphs[key] = self._graphs.pop(key)   # most recently used last
                graph, (p, b, f, loss, flags) = hit
                # The replays run on a stream of their own, behind whatever the caller's stream holds (a
                # render for evaluation, afterTrain's statistics kernel, a refinement) — graphs replayed on
                # the stream that also carried such eager launches ended in GPU memory faults on ROCm 7.0
                # (profiles/HISTORY.md)
                gs = self._g_stream
                gs.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(gs):
                    graph.replay()
                    self._g_done.record()
                self.graph_stats["replays"] += 1
            self._g_done.synchronize()
            M, longest = int(mh[0]), int(mh[1])
            b.num_isects = M
            ls = b.workspace.list_stats
            ls[0], ls[1] = M, longest
            if M > b.capacity:
                # the id list was too small: the guard kept Adam and the row index from moving; grow the
                # list (a new key: the old graph is dropped) and repeat the iteration
                self.graph_stats["overflows"] += 1
                self.bin_ws.capacity = M + M // 8 + 1024
                self._graphs.pop(key, None)
                continue
            break
        if hit is None and self.bin_ws.capacity == b.capacity:
            # (least recently used leaves: a data set with many distinct intrinsics keeps its hot keys)
            while len(self._graphs) >= 8:
                self._graphs.pop(next(iter(self._graphs)))
            g = torch.cuda.CUDAGraph()
            self._g_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.graph(g, stream=self._g_stream):
                objs = self._iteration_launches(pc.gcam, degrees_to_use, background, W, H)
            torch.cuda.current_stream().wait_stream(self._g_stream)
            self._graphs[key] = (g, objs)
            self.graph_stats["captures"] += 1
        self.step_count += 1
        self.means_lr = cabi.sched_lr(self.LR["means"], self.MEANS_LR_FINAL, self.max_steps, self.step_count)
        self._visible = M > 0
        self._ctx = (pc.gcam, pc.cam_pos, p, p["rgb_raw"], b, f, flags, degrees_to_use, background, W, H)
        return loss

    def train_step(self, cam, gt, background, degrees_to_use: int):
        """One iteration of opensplat.cpp:151-170 for this rank's camera of the batch.
        Returns the device tensor {mainLoss, l1, ssim} of THIS camera (no host sync)."""
        if self.graph:
            return self._train_step_graph(cam, gt, background, degrees_to_use)
        rgb = self.render(cam, background, degrees_to_use)
        self._cot_div = self.world
        loss, v_rgb = cabi.main_loss(rgb, gt, self.ssim_weight, 1.0 / self.world, True,
                                     out=self.loss_out, workspace=self.loss_ws)
        self.backward(v_rgb)
        self.optimizer_step()
        return loss


    # ---- a batch of cameras per optimiser step on this rank (two in flight) --------------------------
    def _batch_lanes(self, W, H):
        from .pipeline import CameraLane

        if getattr(self, "_lane_key", None) != (W, H, self.N):
            self._lanes = [CameraLane(self.N, W, H, self.dev, clamped=True, loss=True) for _ in range(2)]
            self._lane_key = (W, H, self.N)
        return self._lanes

    def train_step_batch(self, cams, gts, background, degrees_to_use: int, step: int | None = None,
                         step_optimizer: bool = True, serial: bool = False):
        """One optimiser step over the cameras `cams` (same image size) with ground truths `gts` on THIS rank — on
        several ranks every rank brings its own batch of the same length, and the step's loss is the mean over
        all world x len(cams) cameras.  Generalises the per-image body of opensplat.cpp:151-170: render, loss and
        backward of every camera, ONE gradient exchange, ONE optimiser step.

        Two cameras are in flight (pipeline.two_in_flight): camera j + 1's per-Gaussian forward, binning,
        compositing forward and loss run on a second stream under camera j's compositing backward.  The
        gradients are accumulated in camera order (GS_FLAG_ACCUMULATE_GRADS behind an event): the sums of the
        serial loop, bit-identical with Trainer(deterministic=True) (serial=True runs that loop, for tests).
        Factored exchange: every camera's colour-cotangent all-gather starts behind ITS backward and travels under
        the next camera; one geometry all-reduce closes the batch (dist.FactoredExchange).

        step: the iteration number, when the caller runs after_train(step) afterwards — the densification
        statistics (model.cpp:317-337) are then accumulated here, camera by camera in camera order, exactly as a
        sequence of single-camera iterations would, and after_train(step) only takes the refinement decisions.
        Returns the device tensor [len(cams), 3] of {mainLoss, l1, ssim} per camera (no host sync).  The pieces
        schedule of small frames (segmented) is not used for batches."""
        from .pipeline import two_in_flight

        assert len(cams) == len(gts) and len(cams) >= 1
        assert not self.graph, "a captured iteration holds one camera"
        W, H = cams[0]["W"], cams[0]["H"]
        assert all(c["W"] == W and c["H"] == H for c in cams), "one image size per batch"
        self._buffers(W, H)
        lanes = self._batch_lanes(W, H)
        c, N = len(cams), self.N
        flags = cabi.GS_FLAG_LOGIT_OPACITY | cabi.GS_FLAG_CLAMP_IMAGE
        keep = cabi.GS_FLAG_KEEP_RECORDS | (cabi.GS_FLAG_DETERMINISTIC if self.deterministic else 0)
        scale = 1.0 / (self.world * c)
        losses = torch.empty((c, 3), device=self.dev, dtype=torch.float32)
        rest = self.features_rest if self.K > 1 else None
        host = []
        for cam in cams:
            vm = np.asarray(cam["viewmat"], dtype=np.float32)
            host.append((cabi.make_camera(cam["viewmat"], cam["projmat"], cam["fx"], cam["fy"], cam["cx"], cam["cy"],
                                          W, H, flags=cabi.GS_CAM_LOG_SCALES),
                         (-vm[:3, :3].T @ vm[:3, 3]).astype(np.float32)))        # model.cpp:95
        fx = None
        if self.factored:
            if self.fx is None or self.fx.N != N or self.fx.cpr != c:
                self.fx = dist.FactoredExchange(N, self.K, c, self.dev)
            fx = self.fx
            pos_dev = []
            for cam, (_, cam_pos) in zip(cams, host):      # (one synchronising upload per camera, ever)
                key = (np.asarray(cam["viewmat"], dtype=np.float32).tobytes(), str(self.dev))
                hit = cam.get("_cam_pos_dev")
                if hit is None or hit[0] != key:
                    hit = cam["_cam_pos_dev"] = (key, torch.from_numpy(cam_pos).to(self.dev))
                pos_dev.append(hit[1])
        collect = step is not None and step < self.stop_split_at
        first_stats = [collect and self._stats is None]
        if first_stats[0]:
            self._stats = tuple(torch.zeros(N, device=self.dev, dtype=torch.float32) for _ in range(3))

        def front(L, j):
            gcam, cam_pos = host[j]
            with torch.cuda.stream(L.stream):
                L.g = cabi.gaussian_forward(gcam, self.means, self.log_scales, self.quats, self.opacity_logits,
                                            self.features_dc, rest, cam_pos, degrees_to_use, flags, out=L.gfwd)
                L.b = cabi.bin_and_sort(W, H, None, L.g["depths"], None, None, None, None, None, L.ws,
                                        speculative=True, packed=L.g["packed"],
                                        zero=None if self.deterministic else L.rec_zero)
                L.f = cabi.rasterize_forward(W, H, L.b, background, flags, out=L.fwd)
                L.loss, L.v_rgb = cabi.main_loss(L.f["img_clamped"], gts[j], self.ssim_weight, scale, True,
                                                 out=L.loss_out, workspace=L.loss_ws)

        def back(L, j, prev):
            gcam, cam_pos = host[j]
            with torch.cuda.stream(L.stream):
                losses[j].copy_(L.loss)
                cabi.rasterize_backward(W, H, N, L.b, background, L.f["final_Ts"], L.f["final_idx"], L.v_rgb,
                                        flags | keep | (cabi.GS_FLAG_RECORDS_ZEROED if getattr(L.b, "zeroed", False) else 0),
                                        workspace=L.bwd_ws, img_raw=L.f["img"])
                if prev is not None:
                    L.stream.wait_event(prev.done)      # the flat gradient buffer and the statistics: camera order
                gout, gflags = self.gout, flags | (cabi.GS_FLAG_ACCUMULATE_GRADS if j > 0 else 0)
                if fx is not None:
                    gout = dict(self.gout, v_dc=fx.v_color(j), v_rest=None)
                    fx.set_cam_pos(j, pos_dev[j])
                    gflags |= cabi.GS_FLAG_EMIT_VCOLOR
                cabi.gaussian_backward(gcam, self.means, self.log_scales, self.quats, self.opacity_logits, cam_pos,
                                       self.K, degrees_to_use, L.g["radii"], L.g["rgb_raw"], L.bwd_ws, gout, gflags,
                                       v_xy=L.v_xy)
                if fx is not None:
                    fx.start_camera(j)
                if collect:
                    # (only rank 0's first camera takes the "first iteration" branch: see after_train)
                    cabi.densify_stats(L.v_xy, L.g["radii"], float(max(H, W)),
                                       first_stats[0] and j == 0 and self.rank == 0, *self._stats)
                L.done.record(L.stream)

        seen = [False]

        def validate(L):
            ok = cabi.validate_binning(L.b)
            seen[0] = seen[0] or (ok and L.b.num_isects > 0)
            return ok

        last = two_in_flight(lanes, c, front, validate, back, serial=serial)
        # the exchange, consumed bucket by bucket by optimizer_step() like backward()'s
        if fx is not None:
            fx.start(self.grads)
            sh = self.grads.sh_numel
            self._pending = [(0, sh, lambda: fx.finish_sh(self.grads, self.means, degrees_to_use)),
                             (sh, self.grads.flat.numel(), fx.finish_geometry)]
        else:
            self._pending = [(lo, hi, (lambda w=w: dist.wait_all(w)))
                             for lo, hi, w in dist.allreduce_buckets_async(self.grads, self.grad_buckets)] \
                if (self.world > 1 or self.bucket_single_rank) else None
        self._visible = seen[0]                     # (some camera of the batch saw a Gaussian)
        self._batch_stats_step = step if collect else None
        self._cot_div = self.world * c              # what the cameras' cotangents were divided by (see _refine)
        self._ctx = (host[-1][0], host[-1][1], last.g, last.g["rgb_raw"], last.b, last.f, flags, degrees_to_use,
                     background, W, H)
        if step_optimizer:
            self.optimizer_step()
        return losses

    # ---- Model::afterTrain (model.cpp:311-494) -------------------------------------------------
    def _param_list(self, buf):
        return [buf.v_means, buf.v_scales, buf.v_quats, buf.v_opacity.view(-1, 1), buf.v_dc,
                buf.v_rest if self.K > 1 else None]

    def after_train(self, step: int):
        """Call after train_step(step's camera).  Returns the densification counts dict when this
        step refined the Gaussian set, else None."""
        gcam, cam_pos, p, rgb_raw, b, f, flags, deg, background, W, H = self._ctx
        N, dev = self.N, self.dev
        if not self._visible and self.world == 1:   # model.cpp:315 (with a batch, another rank's
            return None                             # camera may see Gaussians: the step counts)
        if getattr(self, "_batch_stats_step", None) == step:
            self._batch_stats_step = None            # train_step_batch(step=...) has accumulated this step's cameras
        elif step < self.stop_split_at:   # model.cpp:317-337
            first = self._stats is None
            if first:
                self._stats = tuple(torch.zeros(N, device=dev, dtype=torch.float32) for _ in range(3))
            # A batch of `world` cameras is merged (in _refine) into what ONE rank would have
            # accumulated had it seen the cameras one after the other: only rank 0's first camera
            # takes the "first iteration" branch (every Gaussian counted once, visible or not,
            # model.cpp:321-323); the other ranks start from zero accumulators and count visible
            # Gaussians only (:325-326).  Sums / maxima over ranks then equal the sequential result.
            cabi.densify_stats(self.rgrads["v_xy"], p["radii"], float(max(H, W)),
                               first and self.rank == 0, *self._stats)
        counts = None
        if step % self.refine_every == 0 and step > self.warmup_length and self._stats is not None:
            reset_interval = self.reset_alpha_every * self.refine_every
            do_densify = step < self.stop_split_at and \
                step % reset_interval > self.num_cameras + self.refine_every
            if do_densify:
                counts = self._refine(step, W, H)
            if step < self.stop_split_at and step % reset_interval == self.refine_every:
                if self.reference_alpha_reset:
                    # what model.cpp:464-479 really does: clamp, leave the optimiser state alone,
                    # and the re-bound tensor is unknown to the optimiser until the next refinement
                    cabi.reset_opacity(self.params.v_opacity, 0.2, None, None)
                    self._opacity_frozen = True
                else:
                    # the intended behaviour: keep the opacities trainable, zero their moments
                    cabi.reset_opacity(self.params.v_opacity, 0.2, self.exp_avg.v_opacity,
                                       self.exp_avg_sq.v_opacity)
            self._stats = None           # model.cpp:482-484
        return counts

    def _refine(self, step, W, H):
        gn, vc, m2 = self._stats
        if self.world > 1:
            # one camera per rank: merge the ranks' statistics so that every replica takes the
            # same decisions (sum of norms, sum of counts, max size — see after_train)
            torch.distributed.all_reduce(gn)
            torch.distributed.all_reduce(vc)
            torch.distributed.all_reduce(m2, op=torch.distributed.ReduceOp.MAX)
        cfg = cabi.densify_config(W, H, self.densify_grad_thresh, self.densify_size_thresh,
                                  step < self.stop_screen_size_at, self.split_screen_size,
                                  step > self.refine_every * self.reset_alpha_every)
        # (the statistics are linear in the cotangent scale: 1 / world per camera of a train_step, 1 / (world c)
        # per camera of a train_step_batch)
        cfg = rank_merged_config(cfg, getattr(self, "_cot_div", self.world))
        gen = torch.Generator(device=self.dev).manual_seed(1_000_003 * step)  # same on every rank
        samples_fn = lambda n: torch.randn((2 * n, 3), device=self.dev, generator=gen)
        new = {}

        def alloc(new_n):
            for name in ("params", "exp_avg", "exp_avg_sq"):
                new[name] = dist.GradBuffer(new_n, self.K, self.dev)
            return tuple(self._param_list(new[n]) for n in ("params", "exp_avg", "exp_avg_sq"))
        _, _, _, counts = cabi.densify(cfg, self._param_list(self.params),
                                       self._param_list(self.exp_avg),
                                       self._param_list(self.exp_avg_sq), gn, vc, m2, samples_fn,
                                       alloc)
        if counts["new_n"] <= 0:
            raise RuntimeError("densification at step %d culled every Gaussian (new_n = 0): nothing "
                               "left to train" % step)
        self._opacity_frozen = False     # the refinement re-registers the opacities (model.cpp:253-309)
        if self.morton_order and counts["new_n"] > 1:
            perm = morton_permutation(new["params"].v_means)
            for buf in new.values():
                for t in self._param_list(buf):
                    if t is not None and t.numel() > 0:
                        t.copy_(t.index_select(0, perm))
        self.params, self.exp_avg, self.exp_avg_sq = new["params"], new["exp_avg"], new["exp_avg_sq"]
        self.N = counts["new_n"]
        self.grads = dist.GradBuffer(self.N, self.K, self.dev)
        self._shape = None               # per-N render buffers are rebuilt on the next render
        self._graphs.clear()             # (captured graphs point at the replaced buffers)
        return counts
