#!/usr/bin/env python3
"""Training iterations per second, launch by launch against Trainer(experimental_graph=True) (one captured HIP graph per
iteration), on frames of growing size: where the iteration is launch-bound and where it is not.

    python scripts/bench_train_graph.py [--iters 600] > profiles/train_graph_r04.json
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=600)
    a = ap.parse_args()
    import numpy as np
    import torch

    from opensplat_amd import train
    from train_synthetic_inputs import ground_truth, make_camera, sfm_like_init

    dev = torch.device("cuda", 0)
    out = {"iterations": a.iters, "cases": []}
    for n_init, W, H in ((1500, 160, 96), (6000, 96, 72), (6000, 384, 288), (20000, 384, 288), (100000, 960, 540)):
        K = 16
        rs = np.random.RandomState(0)
        cams = [make_camera((3.5 * math.cos(t), 0.4 * math.sin(2 * t), 3.5 * math.sin(t)), W, H)
                for t in np.linspace(0.0, 2.0 * math.pi, 8, endpoint=False)]
        gt = ground_truth(max(n_init, 8000), K, rs)
        G = train.Trainer(*gt, dev)
        bg = np.zeros(3, np.float32)
        images = [G.render(c, bg, 3).clone() for c in cams]
        init = sfm_like_init(gt, n_init, K, rs)
        row = {"gaussians": n_init, "width": W, "height": H}
        for mode in ("launches", "graph"):
            T = train.Trainer(*init, dev, max_steps=10 * a.iters, experimental_graph=(mode == "graph"), refine_every=10 ** 9)
            pcs = [T.prepare_camera(c) for c in cams] if mode == "graph" else cams
            for step in range(1, 21):                      # warm-up (capacity, capture)
                T.train_step(pcs[step % 8], images[step % 8], bg, 3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for step in range(21, 21 + a.iters):
                T.train_step(pcs[step % 8], images[step % 8], bg, 3)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            row[mode + "_it_per_s"] = a.iters / dt
            row[mode + "_us_per_it"] = dt / a.iters * 1e6
            if mode == "graph":
                row["graph_stats"] = dict(T.graph_stats)
        out["cases"].append(row)
        print(row, file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
