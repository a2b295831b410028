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
    assert l.gs_dist_destroy(comm) == 0


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
