#!/usr/bin/env python3
"""Regenerates tests/golden/train_*.npz from the reference's OWN objects:
SSIM (ssim.cpp, compiled in place into oracle/_ref), the five restated lines of Model::mainLoss /
l1 (oracle/ref_train_shim.cpp), libtorch's torch::optim::Adam and OptimScheduler.  Runs only in
the build container (needs /root/reference and `make -C oracle ref`).

  train_loss.npz   window, {mainLoss, l1, ssim} and d mainLoss / d rendered for two seeded image
                   pairs (ragged 75x53 and 96x64) at ssim weights 0.2 (the default,
                   opensplat.cpp:36), 0 and 1
  train_adam.npz   parameter and both moments after 1, 2 and 6 Adam steps on a seeded problem,
                   lr 0.005 (the scales' learning rate, model.cpp:62); OptimScheduler values

Inputs are regenerated from the seeds by opensplat_amd/scenes.py; a digest of them is stored.
Usage: python tests/golden/make_golden_train.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from opensplat_amd import scenes  # noqa: E402

LOSS_CASES = [("ragged", 75, 53, 11), ("even", 96, 64, 12)]
WEIGHTS = [0.2, 0.0, 1.0]
ADAM_N, ADAM_STEPS, ADAM_SEED, ADAM_LR = 4099, (1, 2, 6), 21, 0.005
SCHED = (0.00016, 0.0000016, 30000)  # model.cpp:61,68 with the default --num-iters
SCHED_STEPS = [0, 1, 2, 100, 7000, 15000, 29999, 30000, 31000]


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    R = oracle.reference()
    out = {}
    g, w2 = R.ssim_window()
    out["window_1d"], out["window_2d"] = g, w2
    for name, W, H, seed in LOSS_CASES:
        rendered, gt = scenes.loss_images(W, H, seed)
        out[f"{name}_digest"] = np.array(digest(rendered, gt))
        for w in WEIGHTS:
            loss, v = R.main_loss(rendered, gt, w)
            out[f"{name}_w{w}_loss"] = loss
            out[f"{name}_w{w}_grad"] = v
    np.savez_compressed(os.path.join(HERE, "train_loss.npz"), **out)

    out = {}
    p0, grads = scenes.adam_problem(ADAM_N, max(ADAM_STEPS), ADAM_SEED)
    out["digest"] = np.array(digest(p0, *grads))
    for k in ADAM_STEPS:
        p, m, v = R.adam_steps(p0, grads[:k], ADAM_LR)
        out[f"p{k}"], out[f"m{k}"], out[f"v{k}"] = p, m, v
    out["sched"] = np.array([R.sched_lr(*SCHED, s) for s in SCHED_STEPS], np.float32)
    np.savez_compressed(os.path.join(HERE, "train_adam.npz"), **out)
    print("wrote train_loss.npz, train_adam.npz")


if __name__ == "__main__":
    main()
