"""Worker of tests/test_gpu_dist_pipeline.py (not a test module): one rank of a 2-rank TRAINING job.
Each rank owns one camera of the batch (opensplat_amd.train.Trainer, one camera per rank); stores the
exchanged gradients of the first iteration and the parameters after three.

    GSPLAT_TEST_EXCHANGE=flat|factored python -m torch.distributed.run --nproc-per-node 2 ... \
        tests/dist_train_worker.py OUT_PREFIX
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def camera_dict(s):
    return dict(viewmat=s.viewmat, projmat=s.projmat, fx=s.fx, fy=s.fy, cx=s.cx, cy=s.cy, W=s.W, H=s.H)


def main():
    import torch

    from opensplat_amd import cabi, dist, scenes
    from opensplat_amd.train import Trainer
    from tests.dist_pipeline_worker import small_c4

    rank, world, local = dist.init_from_env(os.environ.get("GSPLAT_DIST_BACKEND", "gloo"))
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    s = small_c4(rank, N=6000, W=208, H=128)
    T = Trainer(*scenes.raw_parameters(s), device=dev, exchange=os.environ.get("GSPLAT_TEST_EXCHANGE", "auto"),
                sh_degree_interval=1)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    gt = torch.from_numpy(np.random.RandomState(100 + rank).uniform(0, 1, (s.H, s.W, 3)).astype(np.float32)).to(dev)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    cpr = int(os.environ.get("GSPLAT_TEST_CPR", "1"))
    if cpr > 1:
        # a batch of cpr cameras per rank and optimiser step (Trainer.train_step_batch, two in flight)
        cams = [camera_dict(small_c4(rank * cpr + j, N=6000, W=208, H=128)) for j in range(cpr)]
        gts = [torch.from_numpy(np.random.RandomState(100 + rank * cpr + j).uniform(0, 1, (s.H, s.W, 3))
                                .astype(np.float32)).to(dev) for j in range(cpr)]
        T.train_step_batch(cams, gts, bg, 3, step_optimizer=False)
        for lo, hi, ready in (T._pending or []):
            ready()
        T._pending = None
        torch.cuda.synchronize()
        np.save(sys.argv[1] + "_grads_rank%d.npy" % rank, T.grads.flat.cpu().numpy())
        for it in range(3):
            T.train_step_batch(cams, gts, bg, 3)
        torch.cuda.synchronize()
        np.save(sys.argv[1] + "_params_rank%d.npy" % rank, T.params.flat.cpu().numpy())
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        return
    cam = camera_dict(s)
    # iteration 1 by hand: backward + the exchange, gradients kept before Adam touches anything
    rgb = T.render(cam, bg, 3)
    loss, v_rgb = cabi.main_loss(rgb, gt, T.ssim_weight, 1.0 / T.world, True, out=T.loss_out, workspace=T.loss_ws)
    T.backward(v_rgb)
    for lo, hi, ready in (T._pending or []):
        ready()
    T._pending = None
    torch.cuda.synchronize()
    np.save(sys.argv[1] + "_grads_rank%d.npy" % rank, T.grads.flat.cpu().numpy())
    for it in range(3):
        T.train_step(cam, gt, bg, 3)
    torch.cuda.synchronize()
    np.save(sys.argv[1] + "_params_rank%d.npy" % rank, T.params.flat.cpu().numpy())
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
