"""-m gpu: the C++ face of the camera batch (gsplat_ops.hpp: CameraBatch, next to GradExchange) through its test
operator torch.ops.opensplat_amd.camera_batch_step — c cameras over the same raw parameters with two of them in
flight on two HIP streams, gradients accumulated in camera order.  Generalises opensplat.cpp:151-170.

  * deterministic: two in flight == the serial loop, bit for bit (gradients and images), lanes re-used across calls;
  * the batch gradients equal the sum over the cameras of SplatRender's autograd gradients (the one-camera node of
    row f1, oracle-checked in tests/test_gpu_fused.py) to the atomics' summation order."""
import numpy as np
import pytest

from opensplat_amd import scenes
from tests.util import to_dev

pytestmark = pytest.mark.gpu


def _problem(c, N=15000, W=320, H=200):
    import torch

    s = scenes.camera_scene(N, W, H, K=16, seed=9, sigma_px=(0.6, 5.0), znear=1.0, zfar=100.0)
    raw = [to_dev(a) for a in scenes.raw_parameters(s)]
    cams = [scenes.yaw_camera(W, H, y, 1.0, 100.0) for y in (-7.0, -2.0, 3.0, 8.0)[:c]]
    vms = to_dev(np.stack([vm for vm, _ in cams]))
    pms = to_dev(np.stack([pm for _, pm in cams]))
    pos = to_dev(np.stack([(-vm[:3, :3].T @ vm[:3, 3]).astype(np.float32) for vm, _ in cams]))
    v_out = to_dev(np.random.RandomState(3).uniform(-1, 1, (c, H, W, 3)).astype(np.float32))
    bg = torch.tensor([0.2, 0.4, 0.1], device="cuda")
    return s, raw, vms, pms, pos, v_out, bg


@pytest.mark.parametrize("c", [2, 4])
def test_cpp_camera_batch_two_in_flight_equals_serial_and_the_one_camera_node(c):
    import torch

    from opensplat_amd import ops

    s, raw, vms, pms, pos, v_out, bg = _problem(c)
    args = (*raw, vms, pms, pos, s.fx, s.fy, s.cx, s.cy, s.H, s.W, 3, bg, v_out)
    ref = ops.camera_batch_step(*args, deterministic=True, serial=True)
    torch.cuda.synchronize()
    assert all(torch.isfinite(t).all() for t in ref) and float(ref[0].abs().max()) > 0
    for _ in range(3):
        got = ops.camera_batch_step(*args, deterministic=True, serial=False)
        torch.cuda.synchronize()
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
    # against the one-camera autograd node, camera by camera
    params = [r.clone().requires_grad_(True) for r in raw]
    for j in range(c):
        rgb = ops.splat_render(*params, vms[j], pms[j], pos[j], s.fx, s.fy, s.cx, s.cy, s.H, s.W, 3, bg)[0]
        assert float((rgb.detach() - ref[6][j]).abs().max()) == 0.0
        (rgb * v_out[j]).sum().backward()
    torch.cuda.synchronize()
    for name, g, p in zip(("means", "log_scales", "quats", "opacity", "dc", "rest"), ref[:6], params):
        want = p.grad.reshape(g.shape).double()
        assert float((g.double() - want).abs().max()) <= 2e-5 * float(want.abs().max()), name
