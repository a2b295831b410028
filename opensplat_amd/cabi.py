"""ctypes binding of the C ABI (include/gsplat_hip.h) operating on torch GPU tensors.

This is the thinnest possible host: it allocates outputs with torch, passes raw data pointers and
the current HIP stream to libgsplat_hip.so, and returns tensors.  The parity tests and bench.py
drive the kernels through it; the C++/libtorch operators (opensplat_amd/ops.py) call exactly the
same entry points.  No fallback: the library must exist and inputs must be GPU tensors.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _build

GS_TILE = 16
GS_SPLAT_DWORDS = 12
GS_ABI_VERSION = 404   # include/gsplat_hip.h
GS_FLAG_FAST_EXP = 1
GS_FLAG_LOGIT_OPACITY = 2
GS_FLAG_CLAMP_IMAGE = 4
GS_FLAG_KEEP_RECORDS = 8
GS_FLAG_RECORDS_ZEROED = 16
GS_FLAG_ACCUMULATE_GRADS = 32
GS_FLAG_DETERMINISTIC = 64
GS_FLAG_EMIT_VCOLOR = 128
GS_CAM_LOG_SCALES = 1

# every symbol include/gsplat_hip.h declares (tests check they are all exported)
SYMBOLS = [
    "gs_strerror", "gs_last_hip_error", "gs_version", "gs_project_forward", "gs_project_backward",
    "gs_sh_forward", "gs_sh_backward", "gs_sh_forward_fused", "gs_sh_backward_fused", "gs_pack_splats", "gs_bin_workspace_bytes", "gs_bin_scan", "gs_bin_num_isects_offset",
    "gs_bin_sort", "gs_bin_strips", "gs_bin_speculative", "gs_bin_speculative_zero", "gs_bin_and_sort", "gs_block_masks", "gs_rasterize_forward", "gs_rasterize_backward", "gs_rasterize_checkpoint_plan", "gs_rasterize_forward_ckpt",
    "gs_rasterize_backward_ckpt", "gs_rasterize_backward_workspace_bytes", "gs_rasterize_backward_workspace_bytes_det", "gs_debug_expf",
    "gs_debug_timeline", "gs_debug_timeline_read", "gs_debug_row_reduce9", "gs_debug_group_reduce9", "gs_debug_backward_uses_mfma", "gs_debug_time_next_kernel", "gs_gaussian_forward", "gs_gaussian_backward",
    "gs_sh_backward_cameras",
]
# every symbol include/gsplat_train.h declares
TRAIN_SYMBOLS = ["gs_ssim_window", "gs_loss_workspace_bytes", "gs_main_loss", "gs_adam_step",
                 "gs_adam_schedule_row", "gs_adam_step_scheduled", "gs_adam_advance", "gs_stage_f32",
                 "gs_copy_indirect_f32", "gs_sched_lr"]
# every symbol include/gsplat_densify.h declares
DENSIFY_SYMBOLS = ["gs_densify_stats", "gs_densify_workspace_bytes", "gs_densify_plan",
                   "gs_densify_apply", "gs_reset_opacity"]


class GsCamera(C.Structure):
    _fields_ = [("viewmat", C.c_float * 16), ("projmat", C.c_float * 16), ("fx", C.c_float),
                ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("img_width", C.c_int32), ("img_height", C.c_int32), ("clip_thresh", C.c_float),
                ("glob_scale", C.c_float), ("flags", C.c_uint32)]


_lib = None


def lib() -> C.CDLL:
    """Load libgsplat_hip.so (after torch, so both share torch's HIP runtime)."""
    global _lib
    if _lib is None:
        path = os.environ.get("GSPLAT_HIP_LIB", _build.HIP_LIB)  # e.g. an instrumented build
        if not os.path.exists(path):
            raise ImportError("libgsplat_hip.so is not built: run `python -m opensplat_amd._build` "
                              "(no CPU fallback exists)")
        l = C.CDLL(path)
        # the binding below is written against include/gsplat_hip.h at GS_ABI_VERSION: a library built from
        # another header would be called through shifted arguments
        if l.gs_version() != GS_ABI_VERSION:
            raise ImportError("%s has ABI version %d, this binding needs %d: rebuild with "
                              "`python -m opensplat_amd._build`" % (path, l.gs_version(), GS_ABI_VERSION))
        l.gs_strerror.restype = C.c_char_p
        l.gs_last_hip_error.restype = C.c_char_p
        l.gs_bin_workspace_bytes.restype = C.c_size_t
        l.gs_bin_num_isects_offset.restype = C.c_size_t
        l.gs_bin_workspace_bytes.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int]
        l.gs_rasterize_backward_workspace_bytes.restype = C.c_size_t
        l.gs_rasterize_backward_workspace_bytes.argtypes = [C.c_int]
        l.gs_rasterize_backward_workspace_bytes_det.restype = C.c_size_t
        l.gs_rasterize_backward_workspace_bytes_det.argtypes = [C.c_int]
        l.gs_loss_workspace_bytes.restype = C.c_size_t
        l.gs_loss_workspace_bytes.argtypes = [C.c_int, C.c_int]
        l.gs_densify_workspace_bytes.restype = C.c_size_t
        l.gs_densify_workspace_bytes.argtypes = [C.c_int]
        l.gs_sched_lr.restype = C.c_float
        l.gs_sched_lr.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int]
        _lib = l
    return _lib


# every symbol include/gsplat_dist.h declares (libgsplat_dist.so: the gradient exchange on RCCL)
DIST_SYMBOLS = ["gs_dist_unique_id", "gs_dist_init", "gs_dist_allreduce_sum", "gs_dist_allreduce_sum_buckets",
                "gs_dist_allgather", "gs_dist_world_size", "gs_dist_rank", "gs_dist_destroy", "gs_dist_last_error"]
_dist_lib = None


def dist_lib() -> C.CDLL:
    """libgsplat_dist.so (loaded after torch: the process then holds torch's RCCL only)."""
    global _dist_lib
    if _dist_lib is None:
        if not os.path.exists(_build.DIST_LIB):
            raise ImportError("libgsplat_dist.so is not built: run `python -m opensplat_amd._build`")
        l = C.CDLL(_build.DIST_LIB)
        l.gs_dist_last_error.restype = C.c_char_p
        l.gs_dist_allreduce_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        l.gs_dist_allreduce_sum_buckets.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p,
                                                    C.c_void_p]
        l.gs_dist_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        _dist_lib = l
    return _dist_lib


class GsError(RuntimeError):
    pass


def _check(rc: int, what: str) -> None:
    if rc != 0:
        l = lib()
        msg = l.gs_strerror(rc).decode()
        if rc == -4:
            msg += " — " + l.gs_last_hip_error().decode()
        raise GsError("%s failed: %s" % (what, msg))


def _p(t):
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), "GPU contiguous tensor required"
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_camera(viewmat, projmat, fx, fy, cx, cy, W, H, clip=0.01, glob_scale=1.0,
                flags=0) -> GsCamera:
    cam = GsCamera()
    cam.flags = int(flags)
    vm = torch.as_tensor(viewmat, dtype=torch.float32).cpu().reshape(-1).tolist()
    pm = torch.as_tensor(projmat, dtype=torch.float32).cpu().reshape(-1).tolist()
    for i in range(16):
        cam.viewmat[i] = vm[i]
        cam.projmat[i] = pm[i]
    cam.fx, cam.fy, cam.cx, cam.cy = float(fx), float(fy), float(cx), float(cy)
    cam.img_width, cam.img_height = int(W), int(H)
    cam.clip_thresh, cam.glob_scale = float(clip), float(glob_scale)
    return cam


def project_forward(cam: GsCamera, means, scales, quats, viewmat_dev=None, projmat_dev=None,
                    out=None):
    N = means.shape[0]
    f = dict(device=means.device, dtype=torch.float32)
    i = dict(device=means.device, dtype=torch.int32)
    out = out or dict(xys=torch.empty((N, 2), **f), depths=torch.empty((N,), **f),
               radii=torch.empty((N,), **i), conics=torch.empty((N, 3), **f),
               num_tiles_hit=torch.empty((N,), **i), cov3d=torch.empty((N, 6), **f),
               cov2d=torch.empty((N, 3), **f))
    _check(lib().gs_project_forward(C.byref(cam), _p(viewmat_dev), _p(projmat_dev), C.c_int(N),
                                    _p(means), _p(scales), _p(quats), _p(out["xys"]),
                                    _p(out["depths"]), _p(out["radii"]), _p(out["conics"]),
                                    _p(out["num_tiles_hit"]), _p(out["cov3d"]), _p(out["cov2d"]),
                                    _stream()), "gs_project_forward")
    return out


def project_backward(cam: GsCamera, means, scales, quats, radii, v_xy, v_conic, v_depth=None,
                     viewmat_dev=None, projmat_dev=None, out=None):
    N = means.shape[0]
    f = dict(device=means.device, dtype=torch.float32)
    if out is None:
        out = dict(v_means=torch.empty((N, 3), **f), v_scales=torch.empty((N, 3), **f),
                   v_quats=torch.empty((N, 4), **f))
    _check(lib().gs_project_backward(C.byref(cam), _p(viewmat_dev), _p(projmat_dev), C.c_int(N),
                                     _p(means), _p(scales), _p(quats), _p(radii), _p(v_xy),
                                     _p(v_depth), _p(v_conic), _p(out["v_means"]),
                                     _p(out["v_scales"]), _p(out["v_quats"]), _stream()),
           "gs_project_backward")
    return out


def sh_forward(degrees_to_use, dirs, coeffs, out=None):
    N, K = coeffs.shape[0], coeffs.shape[1]
    colors = out if out is not None else torch.empty((N, 3), device=coeffs.device,
                                                     dtype=torch.float32)
    _check(lib().gs_sh_forward(C.c_int(N), C.c_int(K), C.c_int(degrees_to_use), _p(dirs),
                               _p(coeffs), _p(colors), _stream()), "gs_sh_forward")
    return colors


def sh_backward(degrees_to_use, K, dirs, v_colors, out=None):
    N = dirs.shape[0]
    v_coeffs = out if out is not None else torch.empty((N, K, 3), device=dirs.device,
                                                       dtype=torch.float32)
    _check(lib().gs_sh_backward(C.c_int(N), C.c_int(K), C.c_int(degrees_to_use), _p(dirs),
                                _p(v_colors), _p(v_coeffs), _stream()), "gs_sh_backward")
    return v_coeffs


@dataclass
class Binned:
    packed: torch.Tensor          # [N, 12] f32
    tiles_hit: torch.Tensor       # [N] i32
    num_isects: int
    gaussian_ids_sorted: torch.Tensor   # [M] i32, per-tile depth-ordered lists
    tile_bins: torch.Tensor       # [tiles, 2] i32
    block_masks: torch.Tensor = None    # [M] i16: per list entry, the 4x4-pixel blocks of its tile it reaches


class BinWorkspace:
    """Reusable buffers for pack/scan/sort so that a steady-state step allocates nothing."""

    def __init__(self):
        self.ws = None
        self.capacity = 0   # entries the id buffer / sort workspace currently hold
        self.scan_done = None
        # pinned {M, longest tile list}: written by the scan kernel, read by the host
        self.m_host = torch.zeros(2, dtype=torch.int32).pin_memory() if torch.cuda.is_available() else None
        self.list_stats = (C.c_int32 * 2)(0, 0)   # last validated values: scheduling hint
        self.bufs = {}

    def get(self, name, shape, dtype, device):
        n = 1
        for s in shape:
            n *= s
        t = self.bufs.get(name)
        if t is None or t.numel() < n or t.dtype != dtype or t.device != device:
            t = torch.empty(max(n, 1), dtype=dtype, device=device)
            self.bufs[name] = t
        return t[:n].view(*shape)


GS_ERR_CAPACITY = -5
# measurement switch: GSPLAT_BIN=tiles takes the tile-level binning of rounds 1-5 (gs_bin_scan + gs_bin_sort)
# for the speculative path instead of the strip binning of round 6 (gs_bin_strips)
_BIN_MODE = os.environ.get("GSPLAT_BIN", "tiles")


def bin_and_sort(W, H, xys, depths, radii, conics, colors, opacities, cov2d=None,
                 workspace: BinWorkspace | None = None, flags=0, speculative=False,
                 packed=None, zero=None) -> Binned:
    """pack -> count + scan -> scatter + per-tile sort.

    Default: gs_bin_and_sort (one stream sync to read the intersection count M, like the
    reference).  speculative=True: the id buffer keeps the capacity of earlier calls and NOTHING
    synchronises; the caller enqueues the forward kernel and then calls validate_binning(), which
    drains the stream and tells whether M fitted (if not: call again — the buffers have grown).
    packed: records already built by gaussian_forward (xys .. cov2d are then ignored).
    zero (speculative only): a uint8 device tensor zeroed on the way by the count pass (gs_bin_speculative_zero) — the
    record workspace of the rasterize_backward that follows, which then takes GS_FLAG_RECORDS_ZEROED."""
    l = lib()
    N = depths.shape[0]
    dev = depths.device
    w = workspace or BinWorkspace()
    have_packed = packed is not None
    if not have_packed:
        packed = w.get("packed", (N, GS_SPLAT_DWORDS), torch.float32, dev)
    tiles_hit = w.get("tiles_hit", (N,), torch.int32, dev)
    tiles = ((W + GS_TILE - 1) // GS_TILE) * ((H + GS_TILE - 1) // GS_TILE)
    tile_bins = w.get("tile_bins", (tiles, 2), torch.int32, dev)
    # tiles by descending list length: the compositing launches start with the long lists
    tile_order = w.get("tile_order", (tiles,), torch.int32, dev)
    if not have_packed:
        _check(l.gs_pack_splats(C.c_int(W), C.c_int(H), C.c_int(N), _p(xys), _p(radii), _p(conics),
                                _p(colors), _p(opacities), _p(cov2d), _p(packed), _p(tiles_hit),
                                C.c_uint32(flags), _stream()), "gs_pack_splats")
    m_host = w.m_host if w.m_host is not None else torch.zeros(2, dtype=torch.int32).pin_memory()
    while True:
        cap = max(w.capacity, 1024)
        ws_bytes = l.gs_bin_workspace_bytes(N, cap, W, H)
        ws = w.get("ws", (ws_bytes,), torch.uint8, dev)
        ids = w.get("ids_sorted", (cap,), torch.int32, dev)
        masks = w.get("block_masks", (cap,), torch.int16, dev)
        zeroed = zero is not None     # (the fallback modes below fill it themselves)
        if speculative:
            # one call, nothing synchronises: gs_bin_speculative (tile-level kernels, the scan folded into the scatter)
            # or — GSPLAT_BIN=strips — the two-level partition of round 6 (gs_bin_strips); GSPLAT_BIN=scan_sort: the
            # two calls of rounds 1 - 5
            if _BIN_MODE == "scan_sort":
                if zero is not None:
                    zero.zero_()
                _check(l.gs_bin_scan(C.c_int(W), C.c_int(H), C.c_int(N), _p(packed), _p(tile_bins),
                                     _p(tile_order), C.c_void_p(m_host.data_ptr()), _p(ws),
                                     C.c_size_t(ws_bytes), _stream()), "gs_bin_scan")
                _check(l.gs_bin_sort(C.c_int(W), C.c_int(H), C.c_int(N), C.c_int32(cap), _p(packed),
                                     _p(depths), _p(tile_bins), _p(ids), _p(masks), w.list_stats, _p(ws),
                                     C.c_size_t(ws_bytes), _stream()), "gs_bin_sort")
            else:
                args = (C.c_int(W), C.c_int(H), C.c_int(N), C.c_int32(cap), _p(packed), _p(depths),
                        _p(tile_bins), _p(ids), _p(masks), _p(tile_order), C.c_void_p(m_host.data_ptr()),
                        w.list_stats, _p(ws), C.c_size_t(ws_bytes))
                if _BIN_MODE == "strips":
                    if zero is not None:
                        zero.zero_()
                    _check(l.gs_bin_strips(*args, _stream()), "gs_bin_strips")
                elif zero is not None:
                    zb = zero.numel() * zero.element_size()
                    rc = l.gs_bin_speculative_zero(*args, _p(zero), C.c_size_t(zb), _stream())
                    zeroed = rc == 0         # 1 = GS_OK_NOT_ZEROED: too large to ride along, the backward fills it
                    _check(0 if rc == 1 else rc, "gs_bin_speculative_zero")
                else:
                    _check(l.gs_bin_speculative(*args, _stream()), "gs_bin_speculative")
            if w.scan_done is None:
                w.scan_done = torch.cuda.Event()
            # validate_binning waits for THIS, not for the whole stream: {M, longest list} are in pinned memory once
            # the binning's launches have run.  (Not while the stream is being captured: an event recorded inside a
            # graph cannot be waited for outside it, and replaying a graph that holds the record node after the event
            # has been recorded eagerly again ended in GPU memory faults on ROCm 7.0; a captured iteration is validated
            # from the pinned count once the replay has completed, train.py.)
            if not torch.cuda.is_current_stream_capturing():
                w.scan_done.record()
            b = Binned(packed, tiles_hit, -1, ids, tile_bins, masks)
            b.tile_order = tile_order
            b.m_host, b.capacity, b.workspace = m_host, cap, w
            b.zeroed = zeroed             # `zero` is zero once this call's launches have run
            b.list_stats = w.list_stats   # from the previous validated frame
            return b
        rc = l.gs_bin_and_sort(C.c_int(W), C.c_int(H), C.c_int(N), C.c_int32(cap), _p(packed),
                               _p(depths), _p(tile_bins), _p(ids), _p(masks), _p(tile_order),
                               C.c_void_p(m_host.data_ptr()), _p(ws), C.c_size_t(ws_bytes), _stream())
        M = int(m_host[0])
        if rc == GS_ERR_CAPACITY:
            w.capacity = M + M // 8 + 1024
            continue
        _check(rc, "gs_bin_and_sort")
        break
    w.list_stats[0], w.list_stats[1] = int(m_host[0]), int(m_host[1])
    b = Binned(packed, tiles_hit, M, ids[:M], tile_bins, masks[:M])
    b.list_stats = w.list_stats
    b.tile_order = tile_order
    return b


def _wait_event(ev) -> None:
    """Host wait for an event that is a fraction of a millisecond away: poll it.  Event.synchronize() sleeps
    once the wait is longer than the runtime's spin phase, and the wake-up then comes with the host's timer tick:
    at BASELINE config 3 (the scan is 0.8 ms into the step) every step of a timed loop took 4.000 ms — 250 Hz —
    whatever the kernels took (round 6).  GSPLAT_EVENT_WAIT=block restores the blocking wait."""
    if os.environ.get("GSPLAT_EVENT_WAIT") == "block":
        ev.synchronize()
        return
    while not ev.query():
        pass


def validate_binning(b: Binned) -> bool:
    """After a speculative bin_and_sort (typically once the forward kernel has been enqueued behind
    it): wait until the scan kernel has stored the intersection count — an event wait, the stream
    keeps running — and compare it with the capacity the id list was given."""
    _wait_event(b.workspace.scan_done)
    M = int(b.m_host[0])
    b.num_isects = M
    b.workspace.list_stats[0], b.workspace.list_stats[1] = M, int(b.m_host[1])
    if M > b.capacity:
        b.workspace.capacity = M + M // 8 + 1024
        return False
    b.gaussian_ids_sorted = b.gaussian_ids_sorted[:M]
    b.block_masks = b.block_masks[:M]
    return True


class Checkpoints:
    """The compositing forward's per-pixel state every `seg_len` entries of a tile's list, from which the
    backward runs every piece of a list as a wave of its own (frames of few tiles; gsplat_hip.h:
    gs_rasterize_checkpoint_plan).  `plan` sizes the buffer from the previous frame's {M, longest list} and
    says whether the frame is worth it; the same object goes to rasterize_forward and rasterize_backward."""

    def __init__(self):
        self.buf = None
        self.seg_len = 0
        self.max_segments = 0
        self.bytes = 0

    def plan(self, W, H, list_stats, device, seg_len=None, max_segments=None):
        """True if the next forward / backward pair should be given this object.  seg_len / max_segments
        override the library's choice (tests, measurements)."""
        sl, ms, nb = C.c_int32(0), C.c_int32(0), C.c_size_t(0)
        _check(lib().gs_rasterize_checkpoint_plan(C.c_int(W), C.c_int(H), list_stats, C.byref(sl), C.byref(ms),
                                                  C.byref(nb)), "gs_rasterize_checkpoint_plan")
        env = os.environ.get("GSPLAT_SEG_LEN")   # measurements: another piece length than the library's
        if env and seg_len is None and list_stats is not None and list_stats[1] > 0 and \
                (nb.value or os.environ.get("GSPLAT_SEG_FORCE")):
            seg_len = int(env)
            max_segments = (int(list_stats[1]) * 5 // 4 + seg_len - 1) // seg_len + 1
        if seg_len is not None:
            tiles = ((W + 15) // 16) * ((H + 15) // 16)
            sl.value, ms.value = seg_len, max_segments
            nb.value = tiles * max_segments * 4096
        self.seg_len, self.max_segments, self.bytes = sl.value, ms.value, nb.value
        if self.bytes == 0:
            return False
        if self.buf is None or self.buf.numel() < self.bytes or self.buf.device != torch.device(device):
            self.buf = torch.empty((self.bytes + self.bytes // 4,), device=device, dtype=torch.uint8)
        return True

    def args(self):
        return _p(self.buf), C.c_size_t(self.buf.numel()), C.c_int32(self.seg_len), C.c_int32(self.max_segments)

    def frozen(self):
        """This plan as the forward used it (buffer reference, piece length, record count): what a frame keeps
        for ITS backward — a later plan() on the shared object (another render in between, e.g. an evaluation
        view) then changes neither.  The records themselves are only safe until the next forward
        writes the shared buffer: a render() / backward() pair must not be interleaved with another render()
        that plans pieces (Trainer.backward always differentiates the LAST render())."""
        f = Checkpoints()
        f.buf, f.seg_len, f.max_segments, f.bytes = self.buf, self.seg_len, self.max_segments, self.bytes
        return f


_NO_CHECKPOINTS = (None, C.c_size_t(0), C.c_int32(0), C.c_int32(0))
_FWD_FLAGS_ENV = int(os.environ.get("GSPLAT_FWD_FLAGS", "0"), 0)   # measurements (e.g. bits 23..24: entries per step)
_BWD_FLAGS_ENV = int(os.environ.get("GSPLAT_BWD_FLAGS", "0"), 0)   # measurements (bits 21..22: pixels per lane)


def rasterize_forward(W, H, binned: Binned, background, flags=0, out=None, checkpoints=None):
    dev = binned.packed.device
    if out is None:
        out = dict(img=torch.empty((H, W, 3), device=dev, dtype=torch.float32),
                   final_Ts=torch.empty((H, W), device=dev, dtype=torch.float32),
                   final_idx=torch.empty((H, W), device=dev, dtype=torch.int32))
    bg = _vec3(background)
    ck = checkpoints.args() if checkpoints is not None else _NO_CHECKPOINTS
    flags |= _FWD_FLAGS_ENV
    _check(lib().gs_rasterize_forward_ckpt(C.c_int(W), C.c_int(H), _p(binned.gaussian_ids_sorted),
                                           _p(binned.block_masks), _p(binned.tile_bins),
                                           _p(binned.packed), bg, _p(out["img"]),
                                           _p(out["final_Ts"]), _p(out["final_idx"]),
                                           _p(out.get("img_clamped")), getattr(binned, "list_stats", None),
                                           _p(getattr(binned, "tile_order", None)), C.c_uint32(flags),
                                           *ck, _stream()), "gs_rasterize_forward")
    return out


def rasterize_backward(W, H, N, binned: Binned, background, final_Ts, final_idx, v_out, flags=0,
                       v_out_alpha=None, out=None, workspace=None, img_raw=None, checkpoints=None):
    dev = binned.packed.device
    if flags & GS_FLAG_KEEP_RECORDS:   # gradients stay in the workspace records (gaussian_backward)
        out = dict(v_xy=None, v_conic=None, v_colors=None, v_opacity=None)
        assert workspace is not None
    elif out is None:
        out = dict(v_xy=torch.empty((N, 2), device=dev), v_conic=torch.empty((N, 3), device=dev),
                   v_colors=torch.empty((N, 3), device=dev), v_opacity=torch.empty((N,), device=dev))
    ws_bytes = (lib().gs_rasterize_backward_workspace_bytes_det(N) if flags & GS_FLAG_DETERMINISTIC
                else lib().gs_rasterize_backward_workspace_bytes(N))
    if workspace is None or workspace.numel() < ws_bytes:
        workspace = torch.empty((max(ws_bytes, 64),), device=dev, dtype=torch.uint8)
    bg = _vec3(background)
    ck = checkpoints.args() if checkpoints is not None else _NO_CHECKPOINTS
    flags |= _BWD_FLAGS_ENV
    _check(lib().gs_rasterize_backward_ckpt(C.c_int(W), C.c_int(H), C.c_int(N),
                                       _p(binned.gaussian_ids_sorted), _p(binned.block_masks),
                                       _p(binned.tile_bins),
                                       _p(binned.packed), bg, _p(final_Ts), _p(final_idx), _p(v_out),
                                       _p(v_out_alpha), _p(img_raw), _p(out["v_xy"]), _p(out["v_conic"]),
                                       _p(out["v_colors"]), _p(out["v_opacity"]), _p(workspace),
                                       C.c_size_t(workspace.numel()),
                                       getattr(binned, "list_stats", None),
                                       _p(getattr(binned, "tile_order", None)), C.c_uint32(flags),
                                       *ck, _stream()),
           "gs_rasterize_backward")
    return out


def debug_expf(x, flags=0):
    y = torch.empty_like(x)
    _check(lib().gs_debug_expf(C.c_int64(x.numel()), _p(x), _p(y), C.c_uint32(flags), _stream()),
           "gs_debug_expf")
    return y


def debug_row_reduce9(x):
    """x [blocks, 9, 64] -> [blocks, 4, 9]: the backward kernel's per-row (16-lane) reduction."""
    blocks = x.shape[0]
    y = torch.empty((blocks, 4, 9), device=x.device, dtype=torch.float32)
    _check(lib().gs_debug_row_reduce9(C.c_int(blocks), _p(x), _p(y), _stream()), "gs_debug_row_reduce9")
    return y


def debug_group_reduce9(x, mfma=True):
    """x [blocks, 9, 64] -> [blocks, 4, 9]: the backward kernel's nine-value group reduction; mfma: on the
    matrix pipe (group g = lanes {4 g + q + 16 k}), else the DPP butterfly (group = 16-lane row)."""
    blocks = x.shape[0]
    y = torch.empty((blocks, 4, 9), device=x.device, dtype=torch.float32)
    _check(lib().gs_debug_group_reduce9(C.c_int(blocks), _p(x), _p(y), C.c_int(1 if mfma else 0), _stream()),
           "gs_debug_group_reduce9")
    return y


def timeline(enable: bool) -> None:
    """Arm / disarm the calling thread's kernel timeline (gs_debug_timeline)."""
    _check(lib().gs_debug_timeline(C.c_int(1 if enable else 0)), "gs_debug_timeline")


def timeline_read(capacity: int = 4096):
    """-> [(kernel name, ms)] in launch order since the timeline was armed (waits for the events)."""
    nb = 96
    names = C.create_string_buffer(capacity * nb)
    ms = (C.c_float * capacity)()
    count = C.c_int(0)
    _check(lib().gs_debug_timeline_read(C.c_int(capacity), names, C.c_int(nb), ms, C.byref(count)),
           "gs_debug_timeline_read")
    out = []
    for i in range(min(count.value, capacity)):
        raw = names.raw[i * nb:(i + 1) * nb].split(b"\0", 1)[0].decode()
        out.append((raw, float(ms[i])))
    return out


def kernel_short_name(expr: str) -> str:
    """'(gs::k_bucket_sort_wave<8, 512>)' -> 'k_bucket_sort_wave<8,512>'."""
    e = expr.strip()
    while e.startswith("(") and e.endswith(")"):
        e = e[1:-1].strip()
    return e.replace("gs::", "").replace(" ", "")


def time_next_kernel(ev_start, ev_stop):
    """Arm the measurement hook with two torch.cuda.Event(enable_timing=True) objects that have been
    recorded at least once (so that their HIP handles exist)."""
    _check(lib().gs_debug_time_next_kernel(C.c_void_p(ev_start.cuda_event),
                                           C.c_void_p(ev_stop.cuda_event)),
           "gs_debug_time_next_kernel")


def _vec3(v):
    """float[3] argument: a device tensor is passed as a device pointer, anything else by value."""
    if isinstance(v, torch.Tensor) and v.is_cuda:
        return _p(v)
    return (C.c_float * 3)(*[float(x) for x in v])


def sh_forward_fused(degrees_to_use, means, cam_pos, features_dc, features_rest, out=None):
    """-> (colors = max(SH + 0.5, 0), rgb_raw); features_rest [N, K-1, 3] or None for K = 1."""
    N = means.shape[0]
    K = 1 + (features_rest.shape[1] if features_rest is not None else 0)
    if out is None:
        out = (torch.empty((N, 3), device=means.device, dtype=torch.float32),
               torch.empty((N, 3), device=means.device, dtype=torch.float32))
    colors, raw = out
    cp = _vec3(cam_pos)
    _check(lib().gs_sh_forward_fused(C.c_int(N), C.c_int(K), C.c_int(degrees_to_use), _p(means), cp,
                                     _p(features_dc), _p(features_rest), _p(colors), _p(raw),
                                     _stream()), "gs_sh_forward_fused")
    return colors, raw


def sh_backward_fused(degrees_to_use, K, means, cam_pos, rgb_raw, v_colors, out=None):
    N = means.shape[0]
    if out is None:
        out = (torch.empty((N, 3), device=means.device, dtype=torch.float32),
               torch.empty((N, K - 1, 3), device=means.device, dtype=torch.float32) if K > 1 else None)
    v_dc, v_rest = out
    cp = _vec3(cam_pos)
    _check(lib().gs_sh_backward_fused(C.c_int(N), C.c_int(K), C.c_int(degrees_to_use), _p(means), cp,
                                      _p(rgb_raw), _p(v_colors), _p(v_dc), _p(v_rest), _stream()),
           "gs_sh_backward_fused")
    return v_dc, v_rest


# ---------------------------------------------------------------------------------------------
# loss + optimiser (include/gsplat_train.h)

class GsAdamGroup(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p),
                ("exp_avg_sq", C.c_void_p), ("n", C.c_int64), ("lr", C.c_double)]


def ssim_window():
    """The reference's 1-D SSIM window (ssim.cpp:39-45) as a list of 11 floats."""
    g = (C.c_float * 11)()
    _check(lib().gs_ssim_window(g), "gs_ssim_window")
    return list(g)


def main_loss(rendered, gt, ssim_weight=0.2, grad_scale=1.0, want_grad=True, out=None,
              workspace=None):
    """Model::mainLoss + backward.  rendered, gt: [H, W, 3] GPU tensors.
    -> (loss[3] = {mainLoss, l1, ssim} device tensor, v_rendered [H, W, 3] or None)."""
    H, W = rendered.shape[0], rendered.shape[1]
    assert rendered.shape == (H, W, 3) and gt.shape == (H, W, 3)
    need = lib().gs_loss_workspace_bytes(W, H)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=rendered.device, dtype=torch.uint8)
    if out is None:
        out = (torch.empty(3, device=rendered.device, dtype=torch.float32),
               torch.empty_like(rendered) if want_grad else None)
    loss, v = out
    _check(lib().gs_main_loss(C.c_int(W), C.c_int(H), _p(rendered), _p(gt), C.c_float(ssim_weight),
                              C.c_float(grad_scale), _p(loss), _p(v) if want_grad else C.c_void_p(0),
                              _p(workspace), C.c_size_t(workspace.numel()), _stream()),
           "gs_main_loss")
    return loss, (v if want_grad else None)


def adam_step(groups, step, beta1=0.9, beta2=0.999, eps=1e-8):
    """groups: list of (param, grad, exp_avg, exp_avg_sq, lr) flat GPU tensors, updated in place
    (Model::optimizersStep, model.cpp:236-243; lr per group as in model.cpp:61-66)."""
    arr = _adam_group_array(groups)
    _check(lib().gs_adam_step(C.c_int(len(groups)), arr, C.c_int64(step), C.c_double(beta1),
                              C.c_double(beta2), C.c_double(eps), _stream()), "gs_adam_step")


GS_ADAM_ROW_FLOATS = 9   # include/gsplat_train.h: 1 + GS_ADAM_MAX_GROUPS


def _adam_group_array(groups):
    arr = (GsAdamGroup * len(groups))()
    for i, (p, g, m, v, lr) in enumerate(groups):
        n = p.numel()
        assert g.numel() == n and m.numel() == n and v.numel() == n
        arr[i].param, arr[i].grad = p.data_ptr(), g.data_ptr()
        arr[i].exp_avg, arr[i].exp_avg_sq = m.data_ptr(), v.data_ptr()
        for t in (p, g, m, v):
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
        arr[i].n, arr[i].lr = n, float(lr)
    return arr


def adam_schedule_rows(lrs_per_step, first_step, beta1=0.9, beta2=0.999):
    """Host: rows [len(lrs_per_step), GS_ADAM_ROW_FLOATS] of the scalars gs_adam_step(step) hands its kernel,
    for steps first_step, first_step + 1, ... with that step's learning rates (one list per step)."""
    rows = np.zeros((len(lrs_per_step), GS_ADAM_ROW_FLOATS), np.float32)
    f32p = C.POINTER(C.c_float)
    for r, lrs in enumerate(lrs_per_step):
        arr = (C.c_double * len(lrs))(*[float(x) for x in lrs])
        _check(lib().gs_adam_schedule_row(C.c_int(len(lrs)), arr, C.c_int64(first_step + r),
                                          C.c_double(beta1), C.c_double(beta2),
                                          rows[r].ctypes.data_as(f32p)), "gs_adam_schedule_row")
    return rows


def adam_step_scheduled(groups, rows_dev, row_index_dev, guard_dev=None, guard_max=0, beta1=0.9,
                        beta2=0.999, eps=1e-8):
    """gs_adam_step with the per-step scalars taken from row *row_index_dev of rows_dev [R, 9] (device); does
    nothing when guard_dev[0] > guard_max.  groups as in adam_step (lr ignored).  Graph-capturable."""
    arr = _adam_group_array(groups)
    assert rows_dev.is_cuda and rows_dev.dtype == torch.float32 and rows_dev.is_contiguous()
    assert rows_dev.shape[-1] == GS_ADAM_ROW_FLOATS and row_index_dev.dtype == torch.int32
    _check(lib().gs_adam_step_scheduled(C.c_int(len(groups)), arr, _p(rows_dev), _p(row_index_dev),
                                        C.c_int32(rows_dev.shape[0]), _p(guard_dev), C.c_int32(guard_max),
                                        C.c_double(beta1), C.c_double(beta2), C.c_double(eps), _stream()),
           "gs_adam_step_scheduled")


def adam_advance(row_index_dev, guard_dev=None, guard_max=0):
    _check(lib().gs_adam_advance(_p(row_index_dev), _p(guard_dev), C.c_int32(guard_max), _stream()),
           "gs_adam_advance")


def stage_f32(dst_dev, src_pinned, count):
    """dst_dev[:count] <- src_pinned[:count] by a kernel (graph-capturable); src_pinned: a pinned host tensor."""
    assert dst_dev.is_cuda and src_pinned.is_pinned() and src_pinned.dtype == torch.float32
    _check(lib().gs_stage_f32(_p(dst_dev), C.c_void_p(src_pinned.data_ptr()), C.c_int(count), _stream()),
           "gs_stage_f32")


def copy_indirect_f32(dst_dev, src_ptr_pinned, count):
    """dst_dev[:count] <- the float array whose DEVICE address sits in src_ptr_pinned[0] (pinned int64 tensor)."""
    assert dst_dev.is_cuda and src_ptr_pinned.is_pinned() and src_ptr_pinned.dtype == torch.int64
    _check(lib().gs_copy_indirect_f32(_p(dst_dev), C.c_void_p(src_ptr_pinned.data_ptr()), C.c_int64(count),
                                      _stream()), "gs_copy_indirect_f32")


def sched_lr(lr_init, lr_final, max_steps, step):
    return float(lib().gs_sched_lr(lr_init, lr_final, max_steps, step))


# ---------------------------------------------------------------------------------------------
# densification / culling (include/gsplat_densify.h)

class GsDensifyConfig(C.Structure):
    _fields_ = [("half_max_side", C.c_float), ("densify_grad_thresh", C.c_float),
                ("densify_size_thresh", C.c_float), ("split_screen_size", C.c_float),
                ("check_screen_size", C.c_int32), ("cull_alpha_thresh", C.c_float),
                ("cull_huge", C.c_int32), ("cull_scale_thresh", C.c_float),
                ("cull_screen_size", C.c_float)]


class GsGaussianSet(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("means", "log_scales", "quats", "opacity_logits",
                                          "features_dc", "features_rest")]


COUNT_NAMES = ["n_splits", "n_dups", "kept_orig", "kept_split", "kept_dup", "new_n", "added", "culled"]


def densify_stats(xys_grad, radii, max_side, first, xys_grad_norm, vis_counts, max_2d_size):
    """Model::afterTrain's per-iteration statistics (model.cpp:317-337), in place."""
    N = radii.shape[0]
    _check(lib().gs_densify_stats(C.c_int(N), _p(xys_grad), _p(radii), C.c_float(max_side),
                                  C.c_int(int(first)), _p(xys_grad_norm), _p(vis_counts),
                                  _p(max_2d_size), _stream()), "gs_densify_stats")


def densify_config(width, height, grad_thresh=0.0002, size_thresh=0.01, check_screen=True,
                   split_screen=0.05, cull_huge=True, cull_alpha=0.1, cull_scale=0.5,
                   cull_screen=0.15) -> GsDensifyConfig:
    return GsDensifyConfig(0.5 * float(max(width, height)), grad_thresh, size_thresh, split_screen,
                           int(bool(check_screen)), cull_alpha, int(bool(cull_huge)), cull_scale,
                           cull_screen)


def _set(tensors):
    s = GsGaussianSet()
    for name, t in zip([f[0] for f in GsGaussianSet._fields_], tensors):
        if t is not None and t.numel() > 0:
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
            setattr(s, name, t.data_ptr())
    return s


def densify(cfg: GsDensifyConfig, params, exp_avg, exp_avg_sq, xys_grad_norm, vis_counts,
            max_2d_size, samples_fn=None, alloc_fn=None):
    """One refinement (model.cpp:345-458): params / exp_avg / exp_avg_sq are lists of the six
    tensors [means, log_scales, quats, opacity_logits, features_dc, features_rest] (the moment
    lists may be None).  samples_fn(n_splits) -> [2 n_splits, 3] normal samples (default
    torch.randn on the device, like the reference).  Returns (new_params, new_exp_avg,
    new_exp_avg_sq, counts dict); one host sync, to read the counts.  alloc_fn(new_n) -> (params,
    exp_avg, exp_avg_sq) lists lets the caller place the new set (e.g. views of flat buffers)."""
    means = params[0]
    N, dev = means.shape[0], means.device
    K = 1 + (params[5].shape[1] if params[5] is not None and params[5].numel() > 0 else 0)
    need = lib().gs_densify_workspace_bytes(N)
    ws = torch.empty(need, device=dev, dtype=torch.uint8)
    counts = torch.zeros(8, dtype=torch.int32).pin_memory()
    _check(lib().gs_densify_plan(C.c_int(N), C.byref(cfg), _p(xys_grad_norm), _p(vis_counts),
                                 _p(max_2d_size), _p(params[1]), _p(params[3]),
                                 C.c_void_p(counts.data_ptr()), _p(ws), C.c_size_t(need), _stream()),
           "gs_densify_plan")
    torch.cuda.current_stream().synchronize()
    c = dict(zip(COUNT_NAMES, [int(x) for x in counts]))
    n_splits, new_n = c["n_splits"], c["new_n"]
    if samples_fn is None:
        samples_fn = lambda n: torch.randn((2 * n, 3), device=dev)   # model.cpp:360
    samples = samples_fn(n_splits) if n_splits > 0 else None
    if samples is not None:
        samples = samples.to(dev).contiguous()
        assert samples.shape == (2 * n_splits, 3)

    def alloc(like):
        return [torch.empty((new_n,) + tuple(t.shape[1:]), device=dev, dtype=torch.float32)
                if t is not None else None for t in like]
    if alloc_fn is not None:
        new_p, new_m, new_v = alloc_fn(new_n)
    else:
        new_p = alloc(params)
        new_m = alloc(params) if exp_avg is not None else None
        new_v = alloc(params) if exp_avg_sq is not None else None
    src = (GsGaussianSet * 3)(_set(params), _set(exp_avg or [None] * 6), _set(exp_avg_sq or [None] * 6))
    dst = (GsGaussianSet * 3)(_set(new_p), _set(new_m or [None] * 6), _set(new_v or [None] * 6))
    _check(lib().gs_densify_apply(C.c_int(N), C.c_int(K), C.c_int(new_n), _p(samples), src, dst,
                                  _p(ws), C.c_size_t(need), _stream()), "gs_densify_apply")
    return new_p, new_m, new_v, c


def reset_opacity(opacity_logits, reset_value=0.2, exp_avg=None, exp_avg_sq=None):
    """The alpha reset, model.cpp:464-479 (in place)."""
    _check(lib().gs_reset_opacity(C.c_int(opacity_logits.numel()), C.c_float(reset_value),
                                  _p(opacity_logits), _p(exp_avg), _p(exp_avg_sq), _stream()),
           "gs_reset_opacity")


# ---------------------------------------------------------------------------------------------
# Fused per-Gaussian stages (include/gsplat_hip.h): projection + SH + pack / their backwards

def gaussian_forward(cam: GsCamera, means, scales, quats, opacities, features_dc, features_rest,
                     cam_pos, degrees_to_use, flags=0, out=None, want_xys=False, viewmat_dev=None,
                     projmat_dev=None):
    """-> dict(packed [N,12], depths, radii, rgb_raw, xys or None)."""
    N, dev = means.shape[0], means.device
    K = 1 + (features_rest.shape[1] if features_rest is not None and features_rest.numel() > 0 else 0)
    f = dict(device=dev, dtype=torch.float32)
    if out is None:
        out = dict(packed=torch.empty((N, GS_SPLAT_DWORDS), **f), depths=torch.empty((N,), **f),
                   radii=torch.empty((N,), device=dev, dtype=torch.int32),
                   rgb_raw=torch.empty((N, 3), **f), xys=torch.empty((N, 2), **f) if want_xys else None)
    _check(lib().gs_gaussian_forward(C.byref(cam), _p(viewmat_dev), _p(projmat_dev), C.c_int(N),
                                     C.c_int(K), C.c_int(degrees_to_use), _p(means), _p(scales),
                                     _p(quats), _p(opacities), _p(features_dc),
                                     _p(features_rest) if K > 1 else C.c_void_p(0), _vec3(cam_pos),
                                     _p(out["packed"]), _p(out["depths"]), _p(out["radii"]),
                                     _p(out["rgb_raw"]), _p(out.get("xys")), C.c_uint32(flags),
                                     _stream()), "gs_gaussian_forward")
    return out


def gaussian_backward(cam: GsCamera, means, scales, quats, opacities, cam_pos, K, degrees_to_use,
                      radii, rgb_raw, records, out, flags=0, v_xy=None, viewmat_dev=None,
                      projmat_dev=None):
    """records: the uint8 workspace rasterize_backward filled under GS_FLAG_KEEP_RECORDS (left
    zeroed).  out: dict(v_means, v_scales, v_quats, v_opacity, v_dc, v_rest)."""
    N = means.shape[0]
    _check(lib().gs_gaussian_backward(C.byref(cam), _p(viewmat_dev), _p(projmat_dev), C.c_int(N),
                                      C.c_int(K), C.c_int(degrees_to_use), _p(means), _p(scales),
                                      _p(quats), _p(opacities), _vec3(cam_pos), _p(radii), _p(rgb_raw),
                                      _p(records), C.c_size_t(records.numel() * records.element_size()),
                                      _p(out["v_means"]), _p(out["v_scales"]), _p(out["v_quats"]),
                                      _p(out["v_opacity"]), _p(out["v_dc"]),
                                      _p(out["v_rest"]) if (K > 1 and out.get("v_rest") is not None)
                                      else C.c_void_p(0), _p(v_xy),
                                      C.c_uint32(flags), _stream()), "gs_gaussian_backward")
    return out


def sh_backward_cameras(K, degrees_to_use, means, cam_pos, v_colors, v_dc, v_rest, flags=0,
                        cam_pos_stride=None, v_colors_stride=None, n_cams=None):
    """SH gradients from the colour cotangents of several cameras (gs_sh_backward_cameras).
    cam_pos: device tensor whose camera c starts cam_pos_stride floats after camera c - 1 (default:
    a [n_cams, >=3] tensor); v_colors: [n_cams, N, 3] (or a flat tensor with v_colors_stride)."""
    N = means.shape[0]
    if n_cams is None:
        n_cams = cam_pos.shape[0] if cam_pos_stride is None else (cam_pos.numel() // cam_pos_stride)
    cs = cam_pos.shape[1] if cam_pos_stride is None else cam_pos_stride
    vs = N * 3 if v_colors_stride is None else v_colors_stride
    _check(lib().gs_sh_backward_cameras(C.c_int(N), C.c_int(K), C.c_int(degrees_to_use), C.c_int(n_cams),
                                        _p(means), _p(cam_pos), C.c_size_t(cs), _p(v_colors),
                                        C.c_size_t(vs), _p(v_dc),
                                        _p(v_rest) if K > 1 else C.c_void_p(0), C.c_uint32(flags),
                                        _stream()), "gs_sh_backward_cameras")
    return v_dc, v_rest
