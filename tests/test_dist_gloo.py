"""CPU, world_size 2 and 8, gloo: the multi-GPU exchange step (flat gradient buffer, SH block first,
two async all-reduces; the factored exchange) — the same code bench.py runs over RCCL.  World 8 is BASELINE
config 4's rank count: every rank's contribution must arrive in the sum, and the bytes a rank moves must be
the figures of DESIGN.md §7's table."""
import os

import numpy as np
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, N, K, q, one_collective=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from opensplat_amd import dist as gdist

    r, w, _ = gdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    buf = gdist.GradBuffer(N, K, torch.device("cpu"))
    gen = torch.Generator().manual_seed(100 + rank)
    buf.flat.copy_(torch.randn(buf.flat.numel(), generator=gen))
    # each rank only "sees" part of the Gaussians: zero rows elsewhere, like a camera would
    lo, hi = rank * N // world, (rank + 1) * N // world
    buf.v_means[:lo] = 0
    buf.v_means[hi:] = 0
    if one_collective:   # the fused per-Gaussian backward delivers everything at once
        gdist.wait_all(gdist.allreduce_all_async(buf))
    else:
        w1 = gdist.allreduce_sh_async(buf)
        # ... projection backward would run here, overlapping w1 ...
        w2 = gdist.allreduce_rest_async(buf)
        gdist.wait_all(w1, w2)
    q.put((rank, buf.flat.numpy().tobytes()))  # bytes: no shared-memory handles to outlive us
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("one_collective", [False, True], ids=["two_blocks", "one_collective"])
def test_allreduce_of_flat_grad_buffer(one_collective, world):
    N, K = 257, 16
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, K, q, one_collective))
             for r in range(world)]
    for p in procs:
        p.start()
    import numpy as np

    raw = dict(q.get(timeout=120) for _ in range(world))
    got = {r: torch.from_numpy(np.frombuffer(b, dtype=np.float32).copy()) for r, b in raw.items()}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # expected: sum of what each rank held
    from opensplat_amd.dist import GradBuffer

    exp = torch.zeros_like(got[0])
    for rank in range(world):
        b = GradBuffer(N, K, torch.device("cpu"))
        gen = torch.Generator().manual_seed(100 + rank)
        b.flat.copy_(torch.randn(b.flat.numel(), generator=gen))
        lo, hi = rank * N // world, (rank + 1) * N // world
        b.v_means[:lo] = 0
        b.v_means[hi:] = 0
        exp += b.flat
    for rank in range(1, world):
        assert torch.equal(got[0], got[rank]), "ranks disagree after all-reduce"
    assert torch.allclose(got[0], exp, atol=1e-5)
    # what one flat all-reduce moves per rank over a ring: 2 (W - 1) / W x 4 N (11 + 3 K) bytes
    assert b.nbytes == 4 * N * (11 + 3 * K)


def test_single_process_is_noop():
    from opensplat_amd import dist as gdist

    buf = gdist.GradBuffer(5, 4, torch.device("cpu"))
    buf.flat.fill_(3.0)
    assert gdist.allreduce_sh_async(buf) is None and gdist.allreduce_all_async(buf) is None
    gdist.allreduce_grads(buf)
    assert (buf.flat == 3.0).all()


@pytest.mark.parametrize("world", [2, 8])
def test_bench_launches_its_own_ranks(world):
    """`python bench.py --gpus N` (the driver's command form, no torchrun around it) must start N ranks by
    itself — two, and the eight of BASELINE config 4; GSPLAT_BENCH_DRY_LAUNCH stops each rank after the
    rendezvous."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GSPLAT_DIST_BACKEND="gloo", GSPLAT_BENCH_DRY_LAUNCH="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    import re

    # (the ranks share stdout: their lines may run together)
    lines = [json.loads(m) for m in re.findall(r'\{"dry_launch".*?\}', r.stdout)]
    assert sorted(l["rank"] for l in lines) == list(range(world)) and all(l["world"] == world for l in lines)


def test_bench_refuses_a_world_that_does_not_match():
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GSPLAT_BENCH_DRY_LAUNCH="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "rank" in (r.stderr + r.stdout)


def test_bucket_bounds_tile_the_buffer():
    from opensplat_amd.dist import bucket_bounds

    for numel, nb in ((59_000_000, 4), (1000, 4), (1025, 3), (5, 8), (4096, 1)):
        b = bucket_bounds(numel, nb)
        assert b[0][0] == 0 and b[-1][1] == numel and len(b) <= nb
        assert all(x[1] == y[0] for x, y in zip(b, b[1:])) and all(lo < hi for lo, hi in b)
        assert all(lo % 1024 == 0 for lo, _ in b)


def _fx_worker(rank, world, port, N, K, cpr, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch

    from opensplat_amd import dist as gdist

    gdist.init_from_env("gloo")
    dev = torch.device("cpu")
    grads = gdist.GradBuffer(N, K, dev)
    grads.flat.copy_(torch.arange(grads.flat.numel(), dtype=torch.float32) * (rank + 1))
    fx = gdist.FactoredExchange(N, K, cpr, dev)
    for j in range(cpr):
        fx.set_cam_pos(j, torch.tensor([rank, j, 7.0]))
        fx.v_color(j).fill_(10.0 * rank + j)
    fx.start_camera(0)                # (what the pipeline does behind the first camera's backward)
    fx.start(grads)
    gdist.wait_all(*fx._w_gather)     # (finish_sh would launch the HIP kernel: GPU test)
    fx.finish_geometry()
    q.put((rank, grads.flat.numpy().copy(), fx.recv.numpy().copy(), fx.bytes_moved_per_rank))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,cpr", [(2, 2), (8, 1), (8, 2)])
def test_factored_exchange_messages(world, cpr):
    """dist.FactoredExchange on two / eight gloo ranks: the geometry block is summed, the SH block is left
    alone, and every rank holds, per local-camera slot, every rank's message [camera centre | colour
    cotangent] in rank order; the bytes moved are DESIGN §7's."""
    N, K = 37, 16
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fx_worker, args=(r, world, port, N, K, cpr, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict((r, (f, m, b)) for r, f, m, b in (q.get(timeout=120) for _ in range(world)))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    total = N * (3 * K + 11)
    sh = N * 3 * K
    base = np.arange(total, dtype=np.float32)
    for rank in range(world):
        flat, msg, moved = got[rank]
        assert np.array_equal(flat[:sh], base[:sh] * (rank + 1))          # SH block: not exchanged
        # geometry: summed over the ranks (factors 1 .. W; exact in fp32: integers below 2^24)
        assert np.array_equal(flat[sh:], base[sh:] * (world * (world + 1) // 2))
        one = 4 + N * 3                                  # one camera's message
        assert msg.shape == (cpr, world * one)            # slot j: the world's cameras j, in rank order
        for j in range(cpr):
            for r in range(world):
                m = msg[j, r * one:(r + 1) * one]
                assert list(m[0:3]) == [r, j, 7.0]
                assert np.all(m[4:] == 10.0 * r + j)
        # ring all-reduce of the 11 geometry floats per Gaussian + all-gather of cpr messages from W - 1 peers
        assert moved == int(2 * (world - 1) / world * 11 * N * 4 + (world - 1) * cpr * one * 4)
        assert moved < int(2 * (world - 1) / world * 4 * N * (11 + 3 * K))   # ... less than the flat exchange
