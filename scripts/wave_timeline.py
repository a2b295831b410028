#!/usr/bin/env python3
"""Per-wave timeline of the two full-frame compositing kernels (instrumented -DGS_WAVELOG build: every workgroup
logs start / end on the 100 MHz counter, its XCC / CU / SIMD and its tile's list length).  What it answers: how many
waves are resident over the launch (the tail), how a wave's speed depends on how many share its SIMD, how evenly
the dispatcher spreads waves over XCDs / CUs.

    scripts/build_variant.sh wavelog -DGS_WAVELOG
    GSPLAT_HIP_LIB=opensplat_amd/csrc/libgsplat_hip_wavelog.so python scripts/wave_timeline.py [C2|C3] > out.json
"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from opensplat_amd import cabi, scenes  # noqa: E402
from opensplat_amd.pipeline import HotPath  # noqa: E402


def read_log(n):
    buf = (C.c_ulonglong * (4 * n))()
    rc = cabi.lib().gs_debug_wavelog(buf, C.c_int(n))
    assert rc == 0, rc
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).copy()
    return a


def analyse(a, name, slots_per_cu):
    t0, t1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
    ok = t1 > 0
    a, t0, t1 = a[ok], t0[ok], t1[ok]
    base = t0.min()
    s, e = (t0 - base) * 0.01, (t1 - base) * 0.01          # us
    hw = (a[:, 2] & 0xFFFFFFFF).astype(np.int64)
    xcc = (a[:, 2] >> 32).astype(np.int64) & 0xF
    # HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    length = (a[:, 3] >> 32).astype(np.int64)
    dur = e - s
    total = e.max()
    # resident waves over time
    grid = np.linspace(0, total, 101)
    resident = [(int(((s <= t) & (e > t)).sum())) for t in grid]
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    simdid = cuid * 4 + simd
    per_cu = np.bincount(cuid, minlength=1)
    per_cu = per_cu[per_cu > 0]
    # speed of a wave against the average number of waves on its SIMD while it ran (sampled at its midpoint)
    mid = 0.5 * (s + e)
    order = np.argsort(simdid, kind="stable")
    share = np.zeros(len(s))
    for sid in np.unique(simdid):
        idx = np.nonzero(simdid == sid)[0]
        for i in idx:
            share[i] = ((s[idx] <= mid[i]) & (e[idx] > mid[i])).sum()
    rate = length / np.maximum(dur, 1e-3)     # list entries per us
    by_share = {}
    for k in sorted(set(share.astype(int))):
        m = share.astype(int) == k
        if m.sum() >= 20:
            by_share[int(k)] = {"waves": int(m.sum()), "entries_per_us_per_wave": float(np.median(rate[m])),
                                "simd_entries_per_us": float(k * np.median(rate[m]))}
    work_us = float(dur.sum())
    return {
        "kernel": name, "waves": int(len(s)), "span_us": float(total),
        "wave_us_median": float(np.median(dur)), "wave_us_p5_p95": [float(np.percentile(dur, 5)), float(np.percentile(dur, 95))],
        "sum_wave_us": work_us, "mean_resident": work_us / float(total),
        "mean_resident_per_cu": work_us / float(total) / max(len(per_cu), 1),
        "cus_seen": int(len(per_cu)), "simds_seen": int(len(np.unique(simdid))),
        "waves_per_cu_min_max": [int(per_cu.min()), int(per_cu.max())],
        "slots_per_cu_assumed": slots_per_cu,
        "resident_at_percent_of_span": resident,
        "time_when_last_wave_started_us": float(s.max()),
        "fraction_of_span_after_last_start": float(1.0 - s.max() / total),
        "rate_by_waves_sharing_the_simd": by_share,
        "start_us_percentiles": [float(np.percentile(s, q)) for q in (0, 25, 50, 60, 75, 90, 100)],
    }


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
    s = {"C2": scenes.config_c2, "C3": scenes.config_c3}[cfg]()
    pipe = HotPath(s, torch.device("cuda:0"), 0)
    for _ in range(20):
        pipe.step()
    torch.cuda.synchronize()
    tiles = ((s.W + 15) // 16) * ((s.H + 15) // 16)
    KEEP = cabi.GS_FLAG_KEEP_RECORDS
    g = cabi.gaussian_forward(pipe.cam, pipe.means, pipe.scales, pipe.quats, pipe.opac, pipe.features_dc,
                              pipe.features_rest, pipe.cam_pos, s.degrees_to_use, 0, out=pipe.gfwd,
                              viewmat_dev=pipe.vm_dev, projmat_dev=pipe.pm_dev)
    b = cabi.bin_and_sort(s.W, s.H, None, g["depths"], None, None, None, None, None, pipe.ws, speculative=True,
                          packed=g["packed"])
    out = {"config": cfg, "tiles": tiles}
    for _ in range(3):
        f = cabi.rasterize_forward(s.W, s.H, b, pipe.background, 0, out=pipe.fwd)
    assert cabi.validate_binning(b)
    raw = os.environ.get("WAVE_TIMELINE_RAW")      # prefix: the raw records as .npy next to the summary
    a = read_log(min(4 * tiles, 65536))
    if raw:
        np.save(raw + "_" + cfg + "_forward.npy", a)
    out["forward"] = analyse(a, "k_rasterize_forward", 30)
    for _ in range(3):
        cabi.rasterize_backward(s.W, s.H, s.N, b, pipe.background, f["final_Ts"], f["final_idx"], pipe.v_out,
                                KEEP | (1 << 25), workspace=pipe.bwd_ws)
    a = read_log(min(tiles, 65536))
    if raw:
        np.save(raw + "_" + cfg + "_backward_q.npy", a)
    out["backward_q"] = analyse(a, "k_rasterize_backward_q", 19)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
