#!/usr/bin/env python3
"""Per-kernel durations of ONE training iteration on a small frame (where the iteration is bound by the
dispatch of ~20 dependent kernels, not by their work): python scripts/timeline_small.py [N W H]"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def main():
    import numpy as np
    import torch

    from opensplat_amd import cabi, train
    from train_synthetic_inputs import ground_truth, make_camera, sfm_like_init

    n_init, W, H = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (6000, 384, 288)
    K = 16
    rs = np.random.RandomState(0)
    dev = torch.device("cuda", 0)
    cams = [make_camera((3.5 * math.cos(t), 0.4 * math.sin(2 * t), 3.5 * math.sin(t)), W, H)
            for t in np.linspace(0.0, 2.0 * math.pi, 8, endpoint=False)]
    gt = ground_truth(max(n_init, 8000), K, rs)
    G = train.Trainer(*gt, dev)
    bg = np.zeros(3, np.float32)
    images = [G.render(c, bg, 3).clone() for c in cams]
    segmented = os.environ.get("GSPLAT_SEGMENTED", "1") != "0"
    T = train.Trainer(*sfm_like_init(gt, n_init, K, rs), dev, max_steps=10000, segmented=segmented)
    for s in range(1, 40):
        T.train_step(cams[s % 8], images[s % 8], bg, 3)
        T.after_train(s)
    torch.cuda.synchronize()
    agg = {}
    reps = 20
    import time
    for s in range(40, 40 + reps):
        cabi.timeline(True)
        T.train_step(cams[s % 8], images[s % 8], bg, 3)
        T.after_train(s)
        tl = cabi.timeline_read()
        cabi.timeline(False)
        for name, ms in tl:
            e = agg.setdefault(cabi.kernel_short_name(name), [0.0, 0])
            e[0] += ms; e[1] += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(100, 400):
        T.train_step(cams[s % 8], images[s % 8], bg, 3)
        T.after_train(s)
    host_it = (time.perf_counter() - t0) / 300 * 1e6    # the last launch has been queued
    torch.cuda.synchronize()
    per_it = (time.perf_counter() - t0) / 300 * 1e6
    rows = [(k, v[0] / reps * 1e3, v[1] / reps) for k, v in agg.items()]
    out = {"gaussians": n_init, "width": W, "height": H, "segmented_backward": segmented,
           "us_per_iteration_plain_loop": per_it, "us_per_iteration_host_side": host_it,
           "sum_of_kernel_us": sum(r[1] for r in rows), "launches_per_iteration": sum(r[2] for r in rows),
           "kernels": [{"kernel": k, "us": round(us, 2), "launches": n} for k, us, n in rows]}
    print(json.dumps(out))
    for k, us, n in rows:
        print("%-40s %7.2f us x %.1f" % (k, us, n), file=sys.stderr)
    print("sum %.1f us, %d launches; loop %.1f us/iteration (host side %.1f)" % (out["sum_of_kernel_us"], out["launches_per_iteration"], per_it, host_it), file=sys.stderr)


if __name__ == "__main__":
    main()
