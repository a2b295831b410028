"""Worker of tests/test_gpu_dist_pipeline.py (not a test module): one rank of a 2-rank job.  Renders
C4-style camera `rank` over the shared Gaussians through opensplat_amd.pipeline.HotPath (the timed path, incl. the
gradient exchange — flat all-reduce or the factored one), then stores the exchanged gradient buffer.

    python -m torch.distributed.run --nproc-per-node 2 ... tests/dist_pipeline_worker.py OUT_PREFIX
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def small_c4(rank, N=20000, W=400, H=240):
    from opensplat_amd import scenes

    return scenes.camera_scene(N, W, H, K=16, seed=3, sigma_px=(0.6, 5.0), znear=1.0, zfar=100.0,
                               yaw_deg=scenes.C4_YAWS[rank % 8], name="C4small_cam%d" % rank)


def main():
    import torch

    from opensplat_amd import dist
    from opensplat_amd.pipeline import HotPath

    rank, world, local = dist.init_from_env(os.environ.get("GSPLAT_DIST_BACKEND", "gloo"))
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    flags = int(os.environ.get("GSPLAT_TEST_FLAGS", "0"))
    # GSPLAT_TEST_EXCHANGE: "flat" (one all-reduce of the whole buffer) or "factored" (geometry
    # all-reduce + colour-cotangent all-gather + local SH backward); GSPLAT_TEST_CPR cameras per rank
    factored = os.environ.get("GSPLAT_TEST_EXCHANGE", "flat") == "factored"
    cpr = int(os.environ.get("GSPLAT_TEST_CPR", "1"))
    pipe = HotPath(small_c4(rank * cpr), dev, flags, factored=factored, cameras_per_rank=cpr)
    # GSPLAT_TEST_SERIAL=1: the plain camera loop; default: the library's camera batch (two in flight, every
    # camera's all-gather behind its own backward)
    serial = os.environ.get("GSPLAT_TEST_SERIAL", "0") == "1"
    batch = [(sc.viewmat, sc.projmat) for sc in (small_c4(rank * cpr + j) for j in range(cpr))]
    for _ in range(2):
        if cpr == 1:
            pipe.step()
        else:
            pipe.step_cameras(batch, serial=serial)
    torch.cuda.synchronize()
    np.save(sys.argv[1] + "_rank%d.npy" % rank, pipe.grads.flat.cpu().numpy())
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
