#!/usr/bin/env python3
"""Which backward is closer to the CPU reference on very long effective lists: the one-pass walk (T unwound by
hardware reciprocals from T_final) or the pieces (T from the forward's checkpoints)?  6000 Gaussians of random
opacity on 64x48: lists of thousands of entries most of which contribute."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle  # noqa: E402
from opensplat_amd import cabi, scenes  # noqa: E402
from tests.util import np_, oracle_raster, rel_err, to_dev  # noqa: E402
from tests.test_gpu_segmented import _checkpoints, _front  # noqa: E402

rows = []
for name, kw in [("random opacity", {}), ("opacity 0.05", dict(op=0.05)), ("opacity 0.35", dict(op=0.35))]:
    s = scenes.camera_scene(6000, 64, 48, K=0, seed=19, sigma_px=(3.0, 10.0), znear=1.0, zfar=100.0)
    if "op" in kw:
        s.opacities[:] = kw["op"]
    p, b = _front(s)
    v_out = to_dev(s.v_out)
    f1 = cabi.rasterize_forward(s.W, s.H, b, s.background, 0)
    g1 = cabi.rasterize_backward(s.W, s.H, s.N, b, s.background, f1["final_Ts"], f1["final_idx"], v_out, 0)
    ck = _checkpoints(s, 64, 128)
    f2 = cabi.rasterize_forward(s.W, s.H, b, s.background, 0, checkpoints=ck)
    g2 = cabi.rasterize_backward(s.W, s.H, s.N, b, s.background, f2["final_Ts"], f2["final_idx"], v_out, 0,
                                 checkpoints=ck)
    torch.cuda.synchronize()
    fo, go = oracle_raster(oracle.restated(), s, np_(p["xys"]), np_(p["conics"]), s.colors, np_(p["cov2d"]),
                           np_(p["depths"]), s.v_out)
    bins = np_(b.tile_bins)
    idx = np_(f1["final_idx"])
    row = {"scene": name, "longest_list": int((bins[:, 1] - bins[:, 0]).max())}
    for k in ("v_xy", "v_conic", "v_colors", "v_opacity"):
        row[k] = {"one_pass_vs_oracle": rel_err(np_(g1[k]), go[k]), "pieces_vs_oracle": rel_err(np_(g2[k]), go[k]),
                  "pieces_vs_one_pass": rel_err(np_(g2[k]), np_(g1[k]))}
    rows.append(row)
    print(row, file=sys.stderr)
print(json.dumps(rows))
