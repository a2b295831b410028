"""-m gpu: several cameras per step with TWO of them in flight on one GPU (opensplat_amd.pipeline.HotPath.step_cameras: camera j + 1's
per-Gaussian forward, binning and compositing forward on a second stream under camera j's compositing backward).
The accumulated gradients must be the serial camera loop's — bit for bit under GS_FLAG_DETERMINISTIC (same
per-camera sums, accumulated in the same camera order) — for two, three and five cameras (lanes re-used)."""
import numpy as np
import pytest

from opensplat_amd import scenes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ncam", [2, 3, 5])
def test_two_cameras_in_flight_accumulate_like_the_serial_loop(ncam):
    import torch

    from opensplat_amd import cabi
    from opensplat_amd.pipeline import HotPath

    s = scenes.camera_scene(30000, 640, 360, K=16, seed=5, znear=0.01, zfar=100.0)
    dev = torch.device("cuda:0")
    pipe = HotPath(s, dev, cabi.GS_FLAG_DETERMINISTIC)
    cams = [scenes.yaw_camera(s.W, s.H, y) for y in (-9.0, -3.0, 2.0, 6.0, 11.0)][:ncam]

    def serial():
        for j, c in enumerate(cams):
            pipe.set_camera(*c)
            pipe.step_fused(accumulate=j > 0, exchange=False, slot=j)
        torch.cuda.synchronize()
        return pipe.grads.flat.clone()

    ref = serial()
    assert torch.isfinite(ref).all() and float(ref.abs().max()) > 0
    assert torch.equal(serial(), ref)                      # the serial loop itself is reproducible
    for _ in range(3):
        pipe.step_cameras(cams, exchange=False)
        torch.cuda.synchronize()
        assert torch.equal(pipe.grads.flat, ref)
    pipe.step_cameras(cams, exchange=False, serial=True)      # the library's own serial loop: the same sums
    torch.cuda.synchronize()
    assert torch.equal(pipe.grads.flat, ref)
    # every camera contributed: one camera alone gives other sums
    pipe.set_camera(*cams[0])
    pipe.step_fused(accumulate=False, exchange=False)
    torch.cuda.synchronize()
    assert not torch.equal(pipe.grads.flat, ref)
