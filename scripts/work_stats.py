#!/usr/bin/env python3
"""Work counters of the compositing kernels on the bench scene (instrumented -DGS_STATS build).

  cd opensplat_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
      -ffp-contract=off -fno-slp-vectorize -DGS_STATS gs_*.hip -o libgsplat_hip_stats.so
  GSPLAT_HIP_LIB=opensplat_amd/csrc/libgsplat_hip_stats.so python scripts/work_stats.py
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from opensplat_amd.pipeline import HotPath as Pipeline  # noqa: E402
from opensplat_amd import cabi, scenes  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
s = {"C1": scenes.config_c1, "C2": scenes.config_c2, "C3": scenes.config_c3}[cfg]()
if s.v_out is None:
    import numpy as np
    s.v_out = np.random.RandomState(3).uniform(-1, 1, (s.H, s.W, 3)).astype(np.float32)
pipe = Pipeline(s, torch.device("cuda:0"), 0)
l = cabi.lib()
buf = (C.c_ulonglong * 16)()
pipe.step()
l.gs_debug_stats(buf, 1)
pipe.step()
l.gs_debug_stats(buf, 1)
v = list(buf)
fwd = dict(zip(["steps", "steps_with_a_needing_lane", "needing_lanes", "block_entries", "wave_chunks"], v[0:5]))
bwd = dict(zip(["steps", "steps_with_a_needing_lane", "needing_lanes", "block_entries", "wave_chunks"], v[8:13]))
M = pipe.num_isects
# the full-frame backward has sixteen four-lane groups per wave since round 5; flag bits 25..26 of
# GSPLAT_BWD_FLAGS = 2 select the four-group kernels of rounds 2 - 4
bwd_groups = 4 if (int(os.environ.get("GSPLAT_BWD_FLAGS", "0"), 0) >> 25) & 3 == 2 else 16
for d in (fwd, bwd):
    d["steps_per_list_entry"] = d["steps"] / max(M, 1)
    d["groups_per_wave"] = bwd_groups if d is bwd else 4
    d["ideal_steps_if_groups_balanced"] = d["block_entries"] / float(d["groups_per_wave"])
    d["live_lanes_per_needing_step"] = d["needing_lanes"] / max(d["steps_with_a_needing_lane"], 1)
print(json.dumps({"M": M, "forward": fwd, "backward": bwd}))
