#!/usr/bin/env python3
"""Compositing kernel durations of a small-frame training iteration against the list length: the same scene
family as scripts/timeline_small.py at several Gaussian counts, with the binning's {M, longest list}.
Separates what a frame costs whatever its lists (launch, table load, first gather) from what an entry of
the longest list costs.   python scripts/timeline_sweep.py [W H [N,N,...]]"""
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

    W, H = (int(x) for x in sys.argv[1:3]) if len(sys.argv) > 2 else (384, 288)
    K = 16
    dev = torch.device("cuda", 0)
    bg = np.zeros(3, np.float32)
    rows = []
    counts = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else (50, 400, 1500, 6000, 20000)
    for n_init in counts:
        rs = np.random.RandomState(0)
        cams = [make_camera((3.5 * math.cos(t), 0.4 * math.sin(2 * t), 3.5 * math.sin(t)), W, H)
                for t in np.linspace(0.0, 2.0 * math.pi, 8, endpoint=False)]
        gt = ground_truth(max(n_init, 8000), K, rs)
        G = train.Trainer(*gt, dev)
        images = [G.render(c, bg, 3).clone() for c in cams]
        T = train.Trainer(*sfm_like_init(gt, n_init, K, rs), dev, max_steps=10000,
                          segmented=os.environ.get("GSPLAT_SEGMENTED", "1") != "0")
        for s in range(1, 30):
            T.train_step(cams[s % 8], images[s % 8], bg, 3)
        torch.cuda.synchronize()
        agg, reps = {}, 16
        stats = [0, 0]
        for s in range(30, 30 + reps):
            cabi.timeline(True)
            T.train_step(cams[s % 8], images[s % 8], bg, 3)
            tl = cabi.timeline_read()
            cabi.timeline(False)
            stats[0] += int(T.bin_ws.list_stats[0]); stats[1] += int(T.bin_ws.list_stats[1])
            for name, ms in tl:
                e = agg.setdefault(cabi.kernel_short_name(name), [0.0, 0])
                e[0] += ms; e[1] += 1
        row = {"gaussians": n_init, "M": stats[0] / reps, "longest_list": stats[1] / reps}
        for k, v in agg.items():
            if k.startswith("k_rasterize") or k.startswith("memset"):
                row[k] = round(v[0] / reps * 1e3, 2)
        rows.append(row)
        print(row, file=sys.stderr)
    print(json.dumps({"width": W, "height": H, "rows": rows,
                      "note": "us per launch incl. ~5 us of event bracketing (see memset rows)"}))


if __name__ == "__main__":
    main()
