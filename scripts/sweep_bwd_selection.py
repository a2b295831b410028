#!/usr/bin/env python3
"""where the sixteen-group backward (k_rasterize_backward_q) stops paying — list entries per
Gaussian M / N swept at 1080p by the splats' pixel size, both kernels forced on the same lists (flag bits 25..26:
1 = sixteen groups, 2 = the four-group kernels), and what the FIRST frame costs (no statistics yet: the library takes
the four-group kernels).  -> JSON on stdout (profiles/r06/bwd_selection_sweep.json)

    python scripts/sweep_bwd_selection.py [N]
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opensplat_amd import cabi, scenes  # noqa: E402
from opensplat_amd.pipeline import HotPath  # noqa: E402


def time_backward(pipe, flags, reps=8):
    s = pipe.s
    KEEP = cabi.GS_FLAG_KEEP_RECORDS
    g = cabi.gaussian_forward(pipe.cam, pipe.means, pipe.scales, pipe.quats, pipe.opac, pipe.features_dc,
                              pipe.features_rest, pipe.cam_pos, s.degrees_to_use, 0, out=pipe.gfwd,
                              viewmat_dev=pipe.vm_dev, projmat_dev=pipe.pm_dev)
    b = cabi.bin_and_sort(s.W, s.H, None, g["depths"], None, None, None, None, None, pipe.ws, speculative=True,
                          packed=g["packed"])
    f = cabi.rasterize_forward(s.W, s.H, b, pipe.background, 0, out=pipe.fwd)
    assert cabi.validate_binning(b)
    ms = []
    for _ in range(reps):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for x in e:
            x.record()
        cabi.time_next_kernel(e[0], e[1])
        cabi.rasterize_backward(s.W, s.H, s.N, b, pipe.background, f["final_Ts"], f["final_idx"], pipe.v_out,
                                KEEP | flags, workspace=pipe.bwd_ws)
        torch.cuda.synchronize()
        ms.append(e[0].elapsed_time(e[1]))
    return float(np.median(ms[2:])) * 1e3, int(b.num_isects), int(b.list_stats[1])


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
    dev = torch.device("cuda:0")
    rows = []
    for scale in (1.0, 1.4, 1.8, 2.2, 2.6, 3.0, 3.5, 4.0, 5.0):
        s = scenes.camera_scene(N, 1920, 1080, K=16, seed=1, sigma_px=(0.5 * scale, 4.0 * scale), name="sweep")
        pipe = HotPath(s, dev, 0)
        for _ in range(3):
            pipe.step()
        torch.cuda.synchronize()
        q, M, longest = time_backward(pipe, 1 << 25)
        four, _, _ = time_backward(pipe, 2 << 25)
        auto, _, _ = time_backward(pipe, 0)
        rows.append({"sigma_scale": scale, "entries_per_gaussian": M / N, "longest_list": longest, "q_us": q,
                     "four_group_us": four, "library_choice_us": auto, "q_over_four": q / four})
        print(rows[-1], file=sys.stderr)
        del pipe
        torch.cuda.empty_cache()
    print(json.dumps({"N": N, "width": 1920, "height": 1080, "rows": rows}))


if __name__ == "__main__":
    main()
