#!/usr/bin/env python3
"""A/B timing of compositing-kernel variants selected by experimental flag bits, same process,
same inputs (C2 scene): python scripts/ab_raster.py [flagsA flagsB ...] (hex)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from opensplat_amd.pipeline import HotPath as Pipeline  # noqa: E402
from opensplat_amd import cabi, scenes  # noqa: E402

cfg = "c2"
argv = sys.argv[1:]
if argv and argv[0] in ("c2", "c3"):
    cfg, argv = argv[0], argv[1:]
flag_sets = [int(a, 16) for a in argv] or [0x0, 0x100]
s = scenes.config_c3() if cfg == "c3" else scenes.config_c2()
pipe = Pipeline(s, torch.device("cuda:0"), 0, stage_kernels=True)
pipe.step()
torch.cuda.synchronize()
p = pipe.proj
colors, _ = cabi.sh_forward_fused(s.degrees_to_use, pipe.means, pipe.cam_pos, pipe.features_dc,
                                  pipe.features_rest)
b = cabi.bin_and_sort(s.W, s.H, p["xys"], p["depths"], p["radii"], p["conics"], colors, pipe.opac,
                      p["cov2d"], pipe.ws)


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


ref = None
for rnd in range(2):
    for fl in flag_sets:
        f = cabi.rasterize_forward(s.W, s.H, b, s.background, fl, out=pipe.fwd)
        img = f["img"].clone()
        if ref is None:
            ref = img
        same = bool(torch.equal(img, ref))
        tf = timeit(lambda: cabi.rasterize_forward(s.W, s.H, b, s.background, fl, out=pipe.fwd))
        tb = timeit(lambda: cabi.rasterize_backward(s.W, s.H, s.N, b, s.background, f["final_Ts"],
                                                    f["final_idx"], pipe.v_out, fl, out=pipe.rgrads,
                                                    workspace=pipe.bwd_ws))
        print("flags 0x%03x  forward %7.1f us  backward(+memset,unpack) %7.1f us  image==first: %s" %
              (fl, tf, tb, same))
