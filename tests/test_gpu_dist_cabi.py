"""-m gpu: include/gsplat_dist.h (libgsplat_dist.so) — the gradient exchange on RCCL for a C++ caller —
through ctypes and through the libtorch class GradExchange.  The test box has ONE GPU and RCCL does
not admit two ranks on one device, so what runs here is the whole stack on a one-rank communicator
(unique id, ncclCommInitRank, the enqueue path, bucket events, destroy); the multi-rank data path is
the same ncclAllReduce call that torch.distributed's "nccl" backend issues in bench.py."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_through_the_c_abi():
    import torch

    from opensplat_amd import cabi

    l = cabi.dist_lib()
    ident = (C.c_uint8 * 128)()
    assert l.gs_dist_unique_id(ident) == 0 and any(ident)
    comm = C.c_void_p(0)
    rc = l.gs_dist_init(C.byref(comm), 1, 0, ident, 0)
    assert rc == 0, l.gs_dist_last_error()
    assert l.gs_dist_world_size(comm) == 1 and l.gs_dist_rank(comm) == 0
    x = torch.randn(1_000_003, device="cuda")
    ref = x.clone()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert l.gs_dist_allreduce_sum(comm, C.c_void_p(x.data_ptr()), x.numel(), s) == 0
    evs = [torch.cuda.Event() for _ in range(4)]
    for e in evs:
        e.record()
    arr = (C.c_void_p * 4)(*[e.cuda_event for e in evs])
    assert l.gs_dist_allreduce_sum_buckets(comm, C.c_void_p(x.data_ptr()), x.numel(), 4, arr, s) == 0
    for e in evs:
        e.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(x, ref)                      # sum over one rank
    assert l.gs_dist_allreduce_sum_buckets(comm, C.c_void_p(x.data_ptr()), x.numel(), 0, None, s) == -1
    # all-gather over one rank: a device copy (or nothing in place)
    y = torch.zeros_like(x)
    assert l.gs_dist_allgather(comm, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), x.numel(), s) == 0
    assert l.gs_dist_allgather(comm, C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), x.numel(), s) == 0
    torch.cuda.synchronize()
    assert torch.equal(y, ref) and torch.equal(x, ref)
    assert l.gs_dist_allgather(comm, None, C.c_void_p(y.data_ptr()), 4, s) == -1
    assert l.gs_dist_destroy(comm) == 0


def test_rccl_collectives_really_execute_on_a_one_rank_communicator(monkeypatch):
    """GSPLAT_DIST_FORCE_COLLECTIVES=1: the world == 1 shortcuts are off, ncclAllReduce and ncclAllGather
    are enqueued on the stream and run on the GPU (inside this repo RCCL had never executed
    a collective).  Over one rank the sum is the identity and the gather a copy — checked, along with
    in-place gathering and the bucketed form; the libtorch class takes the same path."""
    import torch

    from opensplat_amd import cabi, ops  # noqa: F401

    monkeypatch.setenv("GSPLAT_DIST_FORCE_COLLECTIVES", "1")
    l = cabi.dist_lib()
    ident = (C.c_uint8 * 128)()
    assert l.gs_dist_unique_id(ident) == 0
    comm = C.c_void_p(0)
    assert l.gs_dist_init(C.byref(comm), 1, 0, ident, 0) == 0, l.gs_dist_last_error()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = torch.randn(11 * 1_000_000, device="cuda")          # the geometry block of C2: 44 MB
    ref = x.clone()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    assert l.gs_dist_allreduce_sum(comm, C.c_void_p(x.data_ptr()), x.numel(), s) == 0, l.gs_dist_last_error()
    t1.record()
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    assert t0.elapsed_time(t1) > 0.005          # a kernel ran: the shortcut returns in ~0 ms of stream time
    assert l.gs_dist_allreduce_sum_buckets(comm, C.c_void_p(x.data_ptr()), x.numel(), 4, None, s) == 0
    msg = torch.randn(4 + 3 * 100_000, device="cuda")
    out = torch.zeros_like(msg)
    assert l.gs_dist_allgather(comm, C.c_void_p(msg.data_ptr()), C.c_void_p(out.data_ptr()), msg.numel(), s) == 0
    assert l.gs_dist_allgather(comm, C.c_void_p(msg.data_ptr()), C.c_void_p(msg.data_ptr()), msg.numel(), s) == 0
    torch.cuda.synchronize()
    assert torch.equal(x, ref) and torch.equal(out, msg)
    assert l.gs_dist_destroy(comm) == 0
    # the libtorch class (GradExchange) creates its own communicator: the variable applies there too
    y = torch.randn(59 * 1001, device="cuda")
    yr = y.clone()
    z = torch.ops.opensplat_amd.grad_exchange_selftest(y, 3)
    torch.cuda.synchronize()
    assert torch.equal(z, yr)


def test_grad_exchange_class_of_the_libtorch_surface():
    import torch

    from opensplat_amd import ops  # noqa: F401

    x = torch.randn(59 * 1001, device="cuda")
    ref = x.clone()
    y = torch.ops.opensplat_amd.grad_exchange_selftest(x, 1)
    z = torch.ops.opensplat_amd.grad_exchange_selftest(x, 5)
    torch.cuda.synchronize()
    assert torch.equal(y, ref) and torch.equal(z, ref)
    with pytest.raises(RuntimeError):
        torch.ops.opensplat_amd.grad_exchange_selftest(torch.zeros(8), 1)      # CPU tensor


def test_factored_exchange_of_the_libtorch_surface():
    """GradExchange::exchangeFactored on a one-rank communicator: geometry all-reduce (identity),
    message all-gather (copy) and gs_sh_backward_cameras == the SH backward of the one camera."""
    import torch

    from opensplat_amd import cabi, ops  # noqa: F401
    from tests.util import np_, to_dev

    rs = np.random.RandomState(5)
    N, K, deg = 2500, 16, 3
    means = to_dev(rs.uniform(-2, 2, (N, 3)).astype(np.float32))
    cam = rs.uniform(-5, 5, 3).astype(np.float32)
    vcol = to_dev(rs.normal(size=(N, 3)).astype(np.float32))
    geometry = torch.randn(11 * N, device="cuda")
    gref = geometry.clone()
    message = torch.zeros(4 + 3 * N, device="cuda")
    message[:3] = to_dev(cam)
    message[4:] = vcol.reshape(-1)
    v_dc, v_rest, gathered = torch.ops.opensplat_amd.grad_exchange_factored_selftest(geometry, message, means, K, deg)
    want_dc, want_rest = cabi.sh_backward_fused(deg, K, means, to_dev(cam), torch.ones((N, 3), device="cuda"), vcol)
    torch.cuda.synchronize()
    assert torch.equal(geometry, gref) and torch.equal(gathered, message)
    assert np.abs(np_(v_dc) - np_(want_dc)).max() <= 1e-6 * np.abs(np_(want_dc)).max()
    assert np.abs(np_(v_rest) - np_(want_rest)).max() <= 1e-6 * np.abs(np_(want_rest)).max()
