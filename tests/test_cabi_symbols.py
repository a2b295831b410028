"""CPU: libgsplat_hip.so loads and exports every function include/gsplat_hip.h declares
(no compute calls — there is no GPU here)."""
import ctypes
import os
import re

from opensplat_amd import _build, cabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "gsplat_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_is_built_in_tree():
    assert os.path.exists(_build.HIP_LIB), "run python -m opensplat_amd._build"
    assert os.path.dirname(_build.HIP_LIB).startswith(ROOT)


def test_every_declared_symbol_is_exported():
    names = declared_functions()
    assert len(names) >= 13
    l = ctypes.CDLL(_build.HIP_LIB)
    missing = [n for n in names if not hasattr(l, n)]
    assert not missing, missing
    assert sorted(cabi.SYMBOLS) == names, "cabi.SYMBOLS out of sync with the header"


def test_status_strings_and_version():
    l = cabi.lib()
    assert l.gs_version() >= 100
    assert l.gs_strerror(0) == b"ok"
    for code in (-1, -2, -3, -4):
        assert len(l.gs_strerror(code)) > 3
    assert l.gs_strerror(-99) == b"unknown status"


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device."""
    l = cabi.lib()
    cam = cabi.GsCamera()
    cam.img_width, cam.img_height = 70000, 16
    null = ctypes.c_void_p(0)
    one = ctypes.c_void_p(16)
    # negative N
    assert l.gs_sh_forward(-1, 16, 3, null, null, null, null) == -1
    # K not a valid number of SH bases / degree too high for K
    assert l.gs_sh_forward(10, 5, 0, one, one, one, null) == -1
    assert l.gs_sh_forward(10, 4, 2, one, one, one, null) == -1
    # unaligned coefficient pointer
    assert l.gs_sh_forward(10, 4, 1, one, ctypes.c_void_p(20), one, null) == -1
    # null pointers
    assert l.gs_project_forward(ctypes.byref(cam), null, null, 5, null, null, null, null, null,
                                null, null, null, null, null, null) == -1
    # image side > 65535 -> unsupported
    assert l.gs_project_forward(ctypes.byref(cam), null, null, 5, one, one, one, one, one, one,
                                one, one, one, one, null) == -2
    assert l.gs_pack_splats(70000, 16, 5, one, one, one, one, one, null, one, one, 0, null) == -2
    # N == 0 is a no-op success
    assert l.gs_sh_forward(0, 16, 3, null, null, null, null) == 0
    assert l.gs_project_forward(ctypes.byref(cam), null, null, 0, null, null, null, null, null,
                                null, null, null, null, null, null) == 0


def test_torch_operator_library_registers():
    import torch

    from opensplat_amd import ops  # noqa: F401

    for name in ["project_gaussians", "rasterize_gaussians", "spherical_harmonics", "set_fast_exp"]:
        assert hasattr(torch.ops.opensplat_amd, name)


def test_operators_refuse_cpu_tensors():
    """No CPU fallback: the operators raise on host tensors (reference: CHECK_INPUT, bindings.h:14-19)."""
    import pytest
    import torch

    from opensplat_amd import ops

    with pytest.raises(RuntimeError, match="GPU"):
        ops.spherical_harmonics(0, torch.zeros(4, 3), torch.zeros(4, 1, 3))
    with pytest.raises(RuntimeError, match="GPU"):
        ops.project_gaussians(torch.zeros(4, 3), torch.ones(4, 3), 1.0, torch.ones(4, 4),
                              torch.eye(4), torch.eye(4), 10.0, 10.0, 8.0, 8.0, 16, 16)


def test_fused_stage_argument_validation_without_gpu():
    """gs_gaussian_forward / gs_gaussian_backward reject bad arguments before touching the device."""
    l = cabi.lib()
    cam = cabi.GsCamera()
    cam.img_width, cam.img_height = 640, 480
    null, one, rec = ctypes.c_void_p(0), ctypes.c_void_p(256), ctypes.c_void_p(4096)
    i = ctypes.c_int
    u = ctypes.c_uint32
    fwd = lambda N, K, deg, rest=one, packed=one: l.gs_gaussian_forward(
        ctypes.byref(cam), null, null, i(N), i(K), i(deg), one, one, one, one, one, rest, one, packed,
        one, one, one, null, u(0), null)
    assert fwd(0, 16, 3) == 0                      # nothing to do
    assert fwd(-1, 16, 3) == -1
    assert fwd(10, 5, 0) == -1                     # not a valid number of SH bases
    assert fwd(10, 4, 2) == -1                     # degree too high for K
    assert fwd(10, 16, 3, rest=null) == -1         # K > 1 needs features_rest
    assert fwd(10, 16, 3, packed=ctypes.c_void_p(260)) == -1   # packed records are aligned float4s
    cam.img_width = 70000
    assert fwd(10, 16, 3) == -2
    cam.img_width = 640
    bwd = lambda N, K, deg, rbytes, records=rec: l.gs_gaussian_backward(
        ctypes.byref(cam), null, null, i(N), i(K), i(deg), one, one, one, one, one, one, one, records,
        ctypes.c_size_t(rbytes), one, one, one, one, one, one, null, u(0), null)
    assert bwd(0, 16, 3, 0) == 0
    assert bwd(10, 16, 3, 10 * 64 - 1) == -3       # record workspace too small
    assert bwd(10, 16, 3, 640, records=ctypes.c_void_p(4100)) == -1   # 64-byte aligned records
    assert bwd(10, 7, 0, 640) == -1


def test_every_header_under_include_is_exported():
    """All four C-ABI headers (path, training step, densification, launcher-level compatibility):
    every declared function is exported by libgsplat_hip.so."""
    l = ctypes.CDLL(_build.HIP_LIB)
    for header in ("gsplat_hip.h", "gsplat_train.h", "gsplat_densify.h", "gsplat_compat.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names = sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", text)))
        assert names, header
        missing = [n for n in names if not hasattr(l, n)]
        assert not missing, (header, missing)
    # argument validation of the compatibility launchers (no device work)
    null, one = ctypes.c_void_p(0), ctypes.c_void_p(64)
    assert l.gs_compat_tiles_hit(-1, one, one, 4, 4, one, null) == -1
    assert l.gs_compat_tiles_hit(0, null, null, 4, 4, null, null) == 0
    assert l.gs_compat_map_intersects(5, one, one, one, null, 4, 4, one, one, null) == -1
    assert l.gs_compat_tile_bin_edges(ctypes.c_int64(0), null, null, ctypes.c_int64(0), null) == 0


def test_launcher_level_functions_are_registered():
    import torch

    from opensplat_amd import ops  # noqa: F401

    for name in ["launcher_project_gaussians_forward", "launcher_project_gaussians_backward",
                 "launcher_compute_sh_forward", "launcher_compute_sh_backward",
                 "launcher_map_gaussian_to_intersects", "launcher_get_tile_bin_edges",
                 "launcher_rasterize_forward", "launcher_rasterize_backward"]:
        assert hasattr(torch.ops.opensplat_amd, name), name


def test_dist_library_exports_its_header():
    """include/gsplat_dist.h (the gradient exchange on RCCL) <-> libgsplat_dist.so; argument checks
    that need no device."""
    import torch  # noqa: F401  (loads torch's RCCL first)

    text = open(os.path.join(ROOT, "include", "gsplat_dist.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(gs_dist_[a-z0-9_]+)\s*\(", text)))
    assert sorted(cabi.DIST_SYMBOLS) == names
    l = cabi.dist_lib()
    assert not [n for n in names if not hasattr(l, n)]
    null = ctypes.c_void_p(0)
    assert l.gs_dist_unique_id(null) == -1
    comm = ctypes.c_void_p(0)
    ident = (ctypes.c_uint8 * 128)()
    assert l.gs_dist_init(ctypes.byref(comm), 0, 0, ident, 0) == -1       # world < 1
    assert l.gs_dist_init(ctypes.byref(comm), 2, 2, ident, 0) == -1       # rank out of range
    assert l.gs_dist_allreduce_sum(null, null, 0, null) == -1
    assert l.gs_dist_world_size(null) == 0 and l.gs_dist_rank(null) == -1 and l.gs_dist_destroy(null) == 0


def test_image_library_exports_its_header():
    from opensplat_amd import colmap

    text = open(os.path.join(ROOT, "include", "gsplat_image.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", text)))
    assert names == ["gs_image_strerror", "gs_jpeg_decode_rgb", "gs_jpeg_info"]
    l = colmap.image_lib()
    assert not [n for n in names if not hasattr(l, n)]
    assert l.gs_image_strerror(0) == b"ok" and b"unsupported" in l.gs_image_strerror(-2)


def test_cpp_example_maps_one_hip_runtime_and_one_rccl():
    """A program linked against the in-tree libraries and libtorch must end up with ONE HIP runtime
    and ONE RCCL (two of either corrupt the heap at exit and break stream ordering): the link order
    in _build.build_torch/build_example puts libtorch's bundled ROCm first."""
    import re
    import subprocess

    exe = _build.build_example()
    out = subprocess.run(["ldd", exe], capture_output=True, text=True, timeout=120).stdout
    for stem in ("libamdhip64", "librccl", "libhsa-runtime64"):
        found = sorted({m.group(1) for m in re.finditer(r"=> (\S*%s\.so\S*)" % re.escape(stem), out)})
        assert len(found) == 1, (stem, found)


def test_checkpoint_plan_rules_without_gpu():
    """gs_rasterize_checkpoint_plan is host code: which frames get the backward in pieces, the
    piece length, the record count and the buffer size; the checkpointed entry points refuse a buffer that
    does not fit before touching the device."""
    l = cabi.lib()

    def plan(W, H, stats):
        sl, ms, nb = ctypes.c_int32(-1), ctypes.c_int32(-1), ctypes.c_size_t(1)
        arr = (ctypes.c_int32 * 2)(*stats) if stats is not None else None
        assert l.gs_rasterize_checkpoint_plan(W, H, arr, ctypes.byref(sl), ctypes.byref(ms), ctypes.byref(nb)) == 0
        return sl.value, ms.value, nb.value

    tiles = lambda W, H: ((W + 15) // 16) * ((H + 15) // 16)
    # far from filling the chip: always, pieces of one chunk, records for the longest list + a quarter
    sl, ms, nb = plan(384, 288, (60000, 900))
    assert sl == 64 and ms * sl >= 900 * 5 // 4 and nb == tiles(384, 288) * ms * 4096
    assert plan(96, 72, (20000, 2000))[2] > 0
    # nothing worth cutting / no statistics yet / too many tiles
    assert plan(384, 288, (5000, 100)) == (0, 0, 0)
    assert plan(384, 288, None) == (0, 0, 0)
    assert plan(384, 288, (0, 0)) == (0, 0, 0)
    assert plan(1920, 1080, (2000000, 5000)) == (0, 0, 0)
    # beyond 960 tiles only with a tail: a list four times the mean
    assert plan(1008, 756, (600000, 2000))[2] > 0          # mean 199
    assert plan(1008, 756, (1500000, 1200)) == (0, 0, 0)   # mean 497
    assert plan(1504, 1000, (2300000, 1800))[2] > 0        # 5922 tiles, mean 389
    # very long lists on many tiles: longer pieces keep tiles x records below 2^18 workgroups
    sl, ms, nb = plan(1504, 1000, (3000000, 20000))
    assert sl > 64 and sl & (sl - 1) == 0 and tiles(1504, 1000) * ms <= (1 << 18)
    # ... and the records below 512 MiB
    assert nb <= (512 << 20)
    sl, ms, nb = plan(1008, 756, (600000, 4000))   # 64-entry pieces would take 1 GB
    assert sl > 64 and 0 < nb <= (512 << 20)
    # invalid arguments
    assert l.gs_rasterize_checkpoint_plan(0, 10, None, None, None, None) == -1
    # a buffer that is too small / a piece length that is not a power of two: refused before any launch
    null, one = ctypes.c_void_p(0), ctypes.c_void_p(16)
    bg = (ctypes.c_float * 3)(0.0, 0.0, 0.0)
    fwd = (64, 64, one, one, one, one, bg, one, one, one, null, null, null, 0)
    assert l.gs_rasterize_forward_ckpt(*fwd, one, ctypes.c_size_t(4096), 64, 4, null) == -1
    assert l.gs_rasterize_forward_ckpt(*fwd, one, ctypes.c_size_t(16 * 4 * 4096), 96, 4, null) == -1
    bwd = (64, 64, 10, one, one, one, one, bg, one, one, one, null, null, one, one, one, one, ctypes.c_void_p(64),
           ctypes.c_size_t(1 << 20), null, null, 0)
    assert l.gs_rasterize_backward_ckpt(*bwd, one, ctypes.c_size_t(4096), 64, 4, null) == -1
    assert l.gs_rasterize_backward_ckpt(*bwd, ctypes.c_void_p(24), ctypes.c_size_t(1 << 20), 64, 4, null) == -1   # unaligned
