#!/usr/bin/env python3
"""How the compositing kernels' time follows the number of tiles at C2's density: 1920 pixels wide, the height
swept in tile rows, N proportional to the height (C2 = 67.5 rows).  A wave of the sixteen-group backward is a tile
and the chip holds 19 per CU = 4864 at a time: 8160 tiles are 1.68 "rounds".  If us per 1000 tiles depends on the
fractional round, the kernel pays for the tail of the launch, not for issue slots.  -> JSON on stdout

    python scripts/exp_rounds.py [rows ...]
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opensplat_amd import cabi, scenes  # noqa: E402
from opensplat_amd.pipeline import HotPath  # noqa: E402


def time_kernels(pipe, bwd_flags, reps=10):
    s = pipe.s
    KEEP = cabi.GS_FLAG_KEEP_RECORDS
    g = cabi.gaussian_forward(pipe.cam, pipe.means, pipe.scales, pipe.quats, pipe.opac, pipe.features_dc,
                              pipe.features_rest, pipe.cam_pos, s.degrees_to_use, 0, out=pipe.gfwd,
                              viewmat_dev=pipe.vm_dev, projmat_dev=pipe.pm_dev)
    b = cabi.bin_and_sort(s.W, s.H, None, g["depths"], None, None, None, None, None, pipe.ws, speculative=True,
                          packed=g["packed"])
    fw, bw = [], []
    for _ in range(reps):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for x in e:
            x.record()
        cabi.time_next_kernel(e[0], e[1])
        f = cabi.rasterize_forward(s.W, s.H, b, pipe.background, 0, out=pipe.fwd)
        assert cabi.validate_binning(b)
        cabi.time_next_kernel(e[2], e[3])
        cabi.rasterize_backward(s.W, s.H, s.N, b, pipe.background, f["final_Ts"], f["final_idx"], pipe.v_out,
                                KEEP | bwd_flags, workspace=pipe.bwd_ws)
        torch.cuda.synchronize()
        fw.append(e[0].elapsed_time(e[1]))
        bw.append(e[2].elapsed_time(e[3]))
    return float(np.median(fw[2:])) * 1e3, float(np.median(bw[2:])) * 1e3, int(b.num_isects), int(b.list_stats[1])


def main():
    rows_list = [int(a) for a in sys.argv[1:]] or [20, 30, 38, 40, 42, 50, 60, 68, 76, 80, 84, 100, 120, 124, 160]
    dev = torch.device("cuda:0")
    out = []
    for rows in rows_list:
        H = 16 * rows
        N = int(round(1_000_000 * H / 1080.0))
        s = scenes.camera_scene(N, 1920, H, K=16, seed=1, sigma_px=(0.5, 4.0), name="rounds")
        pipe = HotPath(s, dev, 0)
        for _ in range(3):
            pipe.step()
        torch.cuda.synchronize()
        fwd, q, M, longest = time_kernels(pipe, 1 << 25)
        _, four, _, _ = time_kernels(pipe, 2 << 25)
        tiles = 120 * rows
        r = {"rows": rows, "tiles": tiles, "N": N, "M": M, "longest": longest, "fwd_us": fwd, "bwd_q_us": q,
             "bwd_four_us": four, "rounds_q": tiles / 4864.0, "fwd_ns_per_entry": 1e3 * fwd / M,
             "q_ns_per_entry": 1e3 * q / M, "four_ns_per_entry": 1e3 * four / M}
        out.append(r)
        print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}, file=sys.stderr, flush=True)
        del pipe
        torch.cuda.empty_cache()
    print(json.dumps({"width": 1920, "rows": out}))


if __name__ == "__main__":
    main()
