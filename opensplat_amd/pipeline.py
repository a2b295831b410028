"""The hot path as ONE object per GPU, and a camera batch with two cameras in flight.

HotPath is BASELINE.json's path with every buffer preallocated: per-Gaussian forward (projection + SH + packed
record), binning, compositing forward, compositing backward, per-Gaussian backward, gradient exchange — what
bench.py times and what the BASELINE-size parity tests drive.  It replaces the per-image body of
opensplat.cpp:151-170 (Model::forward + backward) for a caller that holds the parameters as flat device buffers.

step_cameras() generalises that loop to a batch of c cameras per optimiser step on one rank (and per gradient
exchange on several): camera j + 1's front half (gs_gaussian_forward, binning, compositing forward — HBM- and
latency-bound, the vector ALUs mostly idle) is enqueued on a second stream before camera j's back half
(compositing backward — VALU-bound —, gs_gaussian_backward), so the two overlap on the GPU: 1.19-1.26 x the serial
loop at C2 (profiles/bench_r05final_cpr2.json).  Gradients accumulate into the flat buffer in camera order (an
event orders the gs_gaussian_backward launches, which read-modify-write it): the same sums in the same order as the
serial loop, bit-identical under GS_FLAG_DETERMINISTIC (tests/test_gpu_two_in_flight.py).  With several ranks each
camera's colour-cotangent all-gather (dist.FactoredExchange) starts behind ITS backward and runs under the next
camera; ONE geometry all-reduce closes the step.  train.Trainer.train_step_batch runs the same schedule
(two_in_flight below) with the image loss on the lanes and the optimiser behind it.

Only torch (device memory, streams, process group) and the C ABI (cabi) are used: no fallback.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import cabi, dist


def two_in_flight(lanes, n, front, validate, back, serial=False):
    """The schedule of a camera batch on two lanes (objects with .stream and .done, a torch.cuda.Event):
        front(L, j)      enqueue camera j's per-Gaussian forward, binning and compositing forward on L.stream
        validate(L)      True if the id list of L's frame was large enough (else its front runs again)
        back(L, j, prev) enqueue camera j's backward half on L.stream; it must wait for prev.done (the previous
                         camera's lane, None for j = 0) before it touches what the cameras accumulate into, and
                         record L.done behind its last launch
    Camera j + 1's front is enqueued BEFORE camera j's back.  The caller's current stream waits for the last
    camera's `done`.  serial=True: one camera after the other on lanes[0] (the reference order of the tests)."""
    main = torch.cuda.current_stream()
    for L in lanes:
        L.stream.wait_stream(main)
    if serial:
        L, prev = lanes[0], None
        for j in range(n):
            front(L, j)
            while not validate(L):
                front(L, j)
            back(L, j, prev)
            prev = L
        main.wait_event(prev.done)
        return prev
    front(lanes[0], 0)
    prev = None
    for j in range(n):
        L = lanes[j % 2]
        while not validate(L):          # (the id list was too small: this camera's front again)
            front(L, j)
        if j + 1 < n:
            front(lanes[(j + 1) % 2], j + 1)
        back(L, j, prev)
        prev = L
    main.wait_event(prev.done)
    return prev


class HotPath:
    """The hot path on one GPU with every buffer preallocated."""

    def _pieces(self):
        """-> (Checkpoints or None, backward flag bits) for the next forward / backward pair."""
        ls = self.ws.list_stats
        if not self.pieces or ls[1] <= 2 * self.pieces:
            return None, 0
        segs = (int(ls[1]) * 5 // 4 + self.pieces - 1) // self.pieces + 1
        self.ckpt.plan(self.s.W, self.s.H, ls, self.dev, seg_len=self.pieces, max_segments=segs)
        return self.ckpt, 3 << 21

    def __init__(self, scene, device, flags, stage_kernels=False, factored=False, cameras_per_rank=1):
        """scene: opensplat_amd.scenes.Scene (parameters, camera, cotangent v_out); flags: GS_FLAG_* of the
        compositing kernels; stage_kernels: the operator-granular kernels instead of the fused per-Gaussian ones;
        factored / cameras_per_rank: the gradient exchange of a multi-rank run (dist.FactoredExchange)."""
        self.torch, self.cabi, self.dist = torch, cabi, dist
        s = self.s = scene
        self.flags = flags
        dev = self.dev = device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.means, self.scales, self.quats = t(s.means), t(s.scales), t(s.quats)
        self.opac = t(s.opacities.reshape(-1))
        self.features_dc = t(s.sh_coeffs[:, 0, :])      # as OpenSplat stores them (model.hpp)
        self.features_rest = t(s.sh_coeffs[:, 1:, :])
        R, tr = s.viewmat[:3, :3], s.viewmat[:3, 3]
        self.cam_pos = t((-R.T @ tr).astype(np.float32))   # camera centre, model.cpp:95
        self.background = t(np.asarray(s.background, dtype=np.float32))
        self.v_out = t(s.v_out)
        self.vm_dev, self.pm_dev = t(s.viewmat), t(s.projmat)
        self.cam = cabi.make_camera(s.viewmat, s.projmat, s.fx, s.fy, s.cx, s.cy, s.W, s.H)
        N, K, W, H = s.N, s.K, s.W, s.H
        f = dict(device=dev, dtype=torch.float32)
        i = dict(device=dev, dtype=torch.int32)
        self.proj = dict(xys=torch.empty((N, 2), **f), depths=torch.empty((N,), **f),
                         radii=torch.empty((N,), **i), conics=torch.empty((N, 3), **f),
                         num_tiles_hit=torch.empty((N,), **i), cov3d=torch.empty((N, 6), **f),
                         cov2d=torch.empty((N, 3), **f))
        self.sh_out = (torch.empty((N, 3), **f), torch.empty((N, 3), **f))  # colours, raw rgb
        self.ws = cabi.BinWorkspace()
        # --pieces S (measurement): checkpointed forward + the backward in pieces of S list entries, four
        # pixels per lane, planned from the previous step's list statistics like Trainer does on small frames
        self.pieces = int(os.environ.get("GSPLAT_BENCH_PIECES", "0"))
        self.ckpt = cabi.Checkpoints()
        self.fwd = dict(img=torch.empty((H, W, 3), **f), final_Ts=torch.empty((H, W), **f),
                        final_idx=torch.empty((H, W), **i))
        # 2-D gradients (fully written by gs_rasterize_backward) + its record workspace
        ws_bytes = (cabi.lib().gs_rasterize_backward_workspace_bytes_det(N) if flags & cabi.GS_FLAG_DETERMINISTIC
                    else cabi.lib().gs_rasterize_backward_workspace_bytes(N))
        self.bwd_ws = torch.empty((ws_bytes + 64,), device=dev, dtype=torch.uint8)
        # the 64-byte gradient records are zeroed by the binning's count pass on the way (gs_bin_speculative_zero: no fill
        # kernel in front of the compositing backward); GSPLAT_RECORDS_MEMSET=1 keeps the fill (measurement)
        rec_bytes = cabi.lib().gs_rasterize_backward_workspace_bytes(N)
        self.rec_zero = None if (flags & cabi.GS_FLAG_DETERMINISTIC or os.environ.get("GSPLAT_RECORDS_MEMSET") == "1"
                                 or rec_bytes % 16) else self.bwd_ws[:rec_bytes]
        self.g2d = torch.zeros(N * 9, **f)
        self.grads = dist.GradBuffer(N, K, dev)
        # the opacity gradient goes straight into the flat all-reduce buffer
        self.rgrads = dict(v_xy=self.g2d[: 2 * N].view(N, 2), v_conic=self.g2d[2 * N: 5 * N].view(N, 3),
                           v_colors=self.g2d[5 * N: 8 * N].view(N, 3), v_opacity=self.grads.v_opacity)
        self.pb_out = dict(v_means=self.grads.v_means, v_scales=self.grads.v_scales,
                           v_quats=self.grads.v_quats)
        self.num_isects = 0
        self.misses = 0          # forwards repeated because the speculative id list was too small
        self.multi = torch.distributed.is_available() and torch.distributed.is_initialized() and \
            torch.distributed.get_world_size() > 1
        # default: the per-Gaussian stages fused into one kernel per direction (gs_gaussian_*);
        # --stage-kernels runs them as the separate operator-granular kernels instead
        self.stage_kernels = stage_kernels
        self.fx = None
        if stage_kernels:
            self.stage_names = ["project_fwd", "sh_fwd", "bin_sort", "rasterize_fwd", "rasterize_bwd",
                                "sh_bwd", "project_bwd", "allreduce"]
        else:
            self.stage_names = ["gaussian_fwd", "bin_sort", "rasterize_fwd", "rasterize_bwd",
                                "gaussian_bwd", "allreduce"]
            self.gfwd = dict(packed=torch.empty((N, 12), **f), depths=self.proj["depths"],
                             radii=self.proj["radii"], rgb_raw=self.sh_out[1], xys=None)
            self.bwd_ws.zero_()
            self.gout = dict(v_means=self.grads.v_means, v_scales=self.grads.v_scales,
                             v_quats=self.grads.v_quats, v_opacity=self.grads.v_opacity,
                             v_dc=self.grads.v_dc, v_rest=self.grads.v_rest)
        # factored exchange: gs_gaussian_backward hands out the colour cotangent of each local camera
        # (into the all-gather message) instead of SH gradients
        self.fx = dist.FactoredExchange(N, K, cameras_per_rank, dev) if (factored and not stage_kernels) else None

    def set_camera(self, viewmat, projmat):
        """Another camera over the same Gaussians (intrinsics unchanged)."""
        t = lambda a: self.torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        s = self.s
        self.cam = self.cabi.make_camera(viewmat, projmat, s.fx, s.fy, s.cx, s.cy, s.W, s.H)
        self.vm_dev.copy_(t(viewmat))
        self.pm_dev.copy_(t(projmat))
        R, tr = viewmat[:3, :3], viewmat[:3, 3]
        self.cam_pos.copy_(t((-R.T @ tr).astype(np.float32)))

    def step(self, events=None, kernel_events=None, accumulate=False, exchange=True, slot=0):
        """One forward+backward.  events: list that receives the stage-boundary events;
        kernel_events: dict name -> (start, stop) event pairs armed around the two compositing
        kernels alone (gs_debug_time_next_kernel)."""
        if not self.stage_kernels:
            return self.step_fused(events, kernel_events, accumulate, exchange, slot)
        assert not accumulate and exchange, "camera batches per rank run on the fused path"
        torch, cabi, s = self.torch, self.cabi, self.s

        def mark():
            if events is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                events.append(e)

        while True:
            ev_local = []

            def mark():
                if events is not None:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record()
                    ev_local.append(e)

            mark()
            p = cabi.project_forward(self.cam, self.means, self.scales, self.quats, self.vm_dev,
                                     self.pm_dev, out=self.proj)
            mark()
            colors, rgb_raw = cabi.sh_forward_fused(s.degrees_to_use, self.means, self.cam_pos,
                                                    self.features_dc, self.features_rest,
                                                    out=self.sh_out)
            mark()
            b = cabi.bin_and_sort(s.W, s.H, p["xys"], p["depths"], p["radii"], p["conics"], colors,
                                  self.opac, p["cov2d"], self.ws, speculative=True)
            mark()
            if kernel_events is not None:
                cabi.time_next_kernel(*kernel_events["k_rasterize_forward"])
            ck, ckf = self._pieces()
            f = cabi.rasterize_forward(s.W, s.H, b, self.background, self.flags, out=self.fwd, checkpoints=ck)
            mark()
            # was the id list large enough?  Waits for the scan kernel only (long finished while the
            # forward kernel runs): the stream never drains, the host keeps enqueuing.
            if not cabi.validate_binning(b):
                self.misses += 1
                continue
            if kernel_events is not None:
                cabi.time_next_kernel(*kernel_events["k_rasterize_backward"])
            g = cabi.rasterize_backward(s.W, s.H, s.N, b, self.background, f["final_Ts"],
                                        f["final_idx"], self.v_out, self.flags | ckf, out=self.rgrads,
                                        workspace=self.bwd_ws, checkpoints=ck)
            mark()
            cabi.sh_backward_fused(s.degrees_to_use, s.K, self.means, self.cam_pos, rgb_raw,
                                   g["v_colors"], out=(self.grads.v_dc, self.grads.v_rest))
            w1 = self.dist.allreduce_sh_async(self.grads)  # overlaps the projection backward
            mark()
            cabi.project_backward(self.cam, self.means, self.scales, self.quats, p["radii"], g["v_xy"],
                                  g["v_conic"], None, self.vm_dev, self.pm_dev, out=self.pb_out)
            mark()
            break
        self.num_isects = b.num_isects
        if events is not None:
            events.extend(ev_local)

        def mark():
            if events is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                events.append(e)

        w2 = self.dist.allreduce_rest_async(self.grads)
        self.dist.wait_all(w1, w2)
        mark()


    def step_fused(self, events=None, kernel_events=None, accumulate=False, exchange=True, slot=0):
        """Same work with gs_gaussian_forward / gs_gaussian_backward around binning + compositing.
        accumulate: add this camera's gradients to the flat buffer (GS_FLAG_ACCUMULATE_GRADS);
        exchange: exchange the gradients afterwards (the last camera of a rank's batch);
        slot: index of this camera in the rank's batch (factored exchange: its message slot)."""
        torch, cabi, s = self.torch, self.cabi, self.s
        # GSPLAT_RECORDS_ZEROED=1: gs_gaussian_backward zeroes the gradient records behind its read
        # and the per-frame memset is skipped — measured SLOWER at C2 (1.27 vs 1.25 ms): the memset
        # leaves the records in the last-level cache right before the compositing atomics arrive
        ZEROED = cabi.GS_FLAG_RECORDS_ZEROED if os.environ.get("GSPLAT_RECORDS_ZEROED") else 0
        KEEP = cabi.GS_FLAG_KEEP_RECORDS | ZEROED
        # (the 64 MB record memset on a stream of its own, under the forward kernels that leave the HBM idle, was
        # measured in round 5: the backward stage 0.263 -> 0.248 ms, the step 0.640 -> 0.648 ms — the per-Gaussian
        # kernels next to it slow down by more; profiles/HISTORY.md)
        while True:
            ev_local = []

            def mark():
                if events is not None:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record()
                    ev_local.append(e)

            mark()
            g = cabi.gaussian_forward(self.cam, self.means, self.scales, self.quats, self.opac,
                                      self.features_dc, self.features_rest, self.cam_pos,
                                      s.degrees_to_use, 0, out=self.gfwd, viewmat_dev=self.vm_dev,
                                      projmat_dev=self.pm_dev)
            mark()
            b = cabi.bin_and_sort(s.W, s.H, None, g["depths"], None, None, None, None, None, self.ws,
                                  speculative=True, packed=g["packed"], zero=self.rec_zero)
            mark()
            if kernel_events is not None:
                cabi.time_next_kernel(*kernel_events["k_rasterize_forward"])
            ck, ckf = self._pieces()
            f = cabi.rasterize_forward(s.W, s.H, b, self.background, self.flags, out=self.fwd, checkpoints=ck)
            mark()
            if not cabi.validate_binning(b):
                self.misses += 1
                continue
            if kernel_events is not None:
                cabi.time_next_kernel(*kernel_events["k_rasterize_backward"])
            cabi.rasterize_backward(s.W, s.H, s.N, b, self.background, f["final_Ts"], f["final_idx"],
                                    self.v_out, self.flags | KEEP | ckf |
                                    (cabi.GS_FLAG_RECORDS_ZEROED if getattr(b, "zeroed", False) else 0),
                                    workspace=self.bwd_ws, checkpoints=ck)
            mark()
            ACC = cabi.GS_FLAG_ACCUMULATE_GRADS if accumulate else 0
            gout = self.gout
            if self.fx is not None:
                gout = dict(self.gout, v_dc=self.fx.v_color(slot), v_rest=None)
                self.fx.set_cam_pos(slot, self.cam_pos)
                ACC |= cabi.GS_FLAG_EMIT_VCOLOR
            cabi.gaussian_backward(self.cam, self.means, self.scales, self.quats, self.opac,
                                   self.cam_pos, s.K, s.degrees_to_use, g["radii"], g["rgb_raw"],
                                   self.bwd_ws, gout, ZEROED | ACC, viewmat_dev=self.vm_dev,
                                   projmat_dev=self.pm_dev)
            if self.fx is not None:
                # this camera's colour cotangents go on the wire now: the all-gather runs on RCCL's
                # stream while the next local camera is rendered
                self.fx.start_camera(slot)
            mark()
            break
        self.num_isects = b.num_isects
        if events is not None:
            events.extend(ev_local)
        if exchange:
            if self.fx is not None:
                self.fx.start(self.grads)
                self.fx.finish(self.grads, self.means, s.degrees_to_use)
            else:
                self.dist.wait_all(self.dist.allreduce_all_async(self.grads))
        if events is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            events.append(e)

    # ---- camera batches -------------------------------------------------------------------------------
    def camera_block(self, viewmat, projmat):
        """(GsCamera, viewmat, projmat, camera centre) on the device, uploaded once per camera of THIS object
        (intrinsics and image size are the scene's)."""
        cache = self.__dict__.setdefault("_cam_cache", {})
        key = (viewmat.tobytes(), projmat.tobytes())
        hit = cache.get(key)
        if hit is None:
            s = self.s
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
            R, tr = viewmat[:3, :3], viewmat[:3, 3]
            hit = cache[key] = (cabi.make_camera(viewmat, projmat, s.fx, s.fy, s.cx, s.cy, s.W, s.H),
                                t(viewmat), t(projmat), t((-R.T @ tr).astype(np.float32)))
        return hit

    def step_cameras(self, cameras, exchange=True, det=False, serial=False):
        """One step over `cameras` [(viewmat, projmat), ...]: their gradients accumulated into self.grads in camera
        order, then (exchange=True) ONE gradient exchange.  Two cameras are in flight (two_in_flight above);
        serial=True (and the stage-kernel path) runs the plain camera loop on the current stream — the same sums
        in the same order.  With the factored exchange each camera's all-gather starts behind ITS backward."""
        s = self.s
        if serial or self.stage_kernels:
            for j, cam in enumerate(cameras):
                self.set_camera(*cam)
                last = j == len(cameras) - 1
                self.step(accumulate=j > 0, exchange=exchange and last, slot=j)
            return
        if not hasattr(self, "lanes"):
            self.lanes = [CameraLane(s.N, s.W, s.H, self.dev), CameraLane(s.N, s.W, s.H, self.dev)]
        KEEP = cabi.GS_FLAG_KEEP_RECORDS | (cabi.GS_FLAG_DETERMINISTIC if det else 0)
        fx = self.fx
        if fx is not None:
            assert fx.cpr >= len(cameras), "FactoredExchange was sized for fewer cameras per rank"

        def front(L, j):
            L.cam, L.vm_dev, L.pm_dev, L.cam_pos = self.camera_block(*cameras[j])
            with torch.cuda.stream(L.stream):
                L.g = cabi.gaussian_forward(L.cam, self.means, self.scales, self.quats, self.opac, self.features_dc,
                                            self.features_rest, L.cam_pos, s.degrees_to_use, 0, out=L.gfwd,
                                            viewmat_dev=L.vm_dev, projmat_dev=L.pm_dev)
                L.b = cabi.bin_and_sort(s.W, s.H, None, L.g["depths"], None, None, None, None, None, L.ws,
                                        speculative=True, packed=L.g["packed"], zero=None if det else L.rec_zero)
                L.f = cabi.rasterize_forward(s.W, s.H, L.b, self.background, self.flags, out=L.fwd)

        def validate(L):
            ok = cabi.validate_binning(L.b)
            if not ok:
                self.misses += 1
            return ok

        def back(L, j, prev):
            with torch.cuda.stream(L.stream):
                cabi.rasterize_backward(s.W, s.H, s.N, L.b, self.background, L.f["final_Ts"], L.f["final_idx"],
                                        self.v_out, self.flags | KEEP |
                                        (cabi.GS_FLAG_RECORDS_ZEROED if getattr(L.b, "zeroed", False) else 0),
                                        workspace=L.bwd_ws)
                if prev is not None:
                    L.stream.wait_event(prev.done)     # the flat gradient buffer: camera order
                gout, acc = self.gout, (cabi.GS_FLAG_ACCUMULATE_GRADS if j > 0 else 0)
                if fx is not None:
                    gout = dict(self.gout, v_dc=fx.v_color(j), v_rest=None)
                    fx.set_cam_pos(j, L.cam_pos)
                    acc |= cabi.GS_FLAG_EMIT_VCOLOR
                cabi.gaussian_backward(L.cam, self.means, self.scales, self.quats, self.opac, L.cam_pos, s.K,
                                       s.degrees_to_use, L.g["radii"], L.g["rgb_raw"], L.bwd_ws, gout, acc,
                                       viewmat_dev=L.vm_dev, projmat_dev=L.pm_dev)
                if fx is not None:
                    fx.start_camera(j)      # on the wire now, under the next camera's rendering
                L.done.record(L.stream)

        last = two_in_flight(self.lanes, len(cameras), front, validate, back)
        self.num_isects = last.b.num_isects
        if exchange:
            if fx is not None:
                fx.start(self.grads)
                fx.finish(self.grads, self.means, s.degrees_to_use)
            else:
                dist.wait_all(dist.allreduce_all_async(self.grads))


class CameraLane:
    """What ONE camera in flight owns: per-Gaussian forward outputs, binning workspace, images, gradient
    records, its camera block — and a HIP stream.  Two lanes let camera j + 1's per-Gaussian forward and
    binning (HBM- / latency-bound, VALU mostly idle) run under camera j's compositing (VALU-bound)."""

    def __init__(self, N, W, H, dev, clamped=False, loss=False):
        f = dict(device=dev, dtype=torch.float32)
        i = dict(device=dev, dtype=torch.int32)
        self.gfwd = dict(packed=torch.empty((N, cabi.GS_SPLAT_DWORDS), **f), depths=torch.empty((N,), **f),
                         radii=torch.empty((N,), **i), rgb_raw=torch.empty((N, 3), **f), xys=None)
        self.ws = cabi.BinWorkspace()
        self.fwd = dict(img=torch.empty((H, W, 3), **f), final_Ts=torch.empty((H, W), **f),
                        final_idx=torch.empty((H, W), **i))
        if clamped:      # Model::forward's clamp_max(1) image (GS_FLAG_CLAMP_IMAGE)
            self.fwd["img_clamped"] = torch.empty((H, W, 3), **f)
        self.bwd_ws = torch.zeros((cabi.lib().gs_rasterize_backward_workspace_bytes_det(N) + 64,),
                                  device=dev, dtype=torch.uint8)
        rec_bytes = cabi.lib().gs_rasterize_backward_workspace_bytes(N)
        self.rec_zero = None if (os.environ.get("GSPLAT_RECORDS_MEMSET") == "1" or rec_bytes % 16) \
            else self.bwd_ws[:rec_bytes]    # (zeroed by the binning's count pass: HotPath.__init__)
        if loss:         # the image loss of the lane's camera (train.Trainer.train_step_batch)
            self.loss_ws = torch.empty(cabi.lib().gs_loss_workspace_bytes(W, H), device=dev, dtype=torch.uint8)
            self.loss_out = (torch.empty(3, **f), torch.empty((H, W, 3), **f))
            self.v_xy = torch.zeros((N, 2), **f)
        self.stream = torch.cuda.Stream(device=dev)
        self.done = torch.cuda.Event()     # this lane's last gs_gaussian_backward
        self.cam = self.vm_dev = self.pm_dev = self.cam_pos = None
        self.g = self.b = self.f = None
