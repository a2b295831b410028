#!/usr/bin/env python3
"""Host-side model of the compositing backward's work-to-lane mappings (no GPU needed).

Takes a window of tiles of a benchmark scene, rebuilds on the CPU what the kernels see — per-tile depth
lists, per-entry coverage of the tile's sixteen 4x4-pixel blocks, per-pixel last contributors — and counts,
for each candidate geometry, the steps a wave takes, the lanes that do needed work and the VALU instructions
a step costs (the per-pass / per-step instruction counts are the ones of gs_raster.hip):

  cur     a wave = a tile, 4 groups of 16 lanes, a group = an 8x8 block with 4 pixels per lane, 64-entry
          chunks, scalar walk (k_rasterize_backward<.,.,4>)
  q64     a wave = a tile, 16 groups of 4 lanes, a group = a 4x4 block with 4 pixels per lane (a lane = one
          column of the block), 64-entry chunks, per-block queues in LDS (backward_wave_q)
  q128    the same with 128-entry chunks

The model of `cur` is checked against the instrumented kernel's counters (profiles/work_stats_r04_c2.json):
steps per list entry, passes with a needing lane per step, live lanes per pass.

usage: python scripts/sim_bwd_geometry.py [--config c2|c3|hot] [--tiles 24x14] [--out profiles/...json]
"""
import argparse
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from opensplat_amd import scenes  # noqa: E402
import oracle  # noqa: E402  (analysis tooling: allowed to use the checker)


def build_window(s, tx0, ty0, ntx, nty):
    orc = oracle.restated()
    p = orc.project_forward(s.means, s.scales, s.quats, s.viewmat, s.projmat, s.fx, s.fy, s.cx, s.cy,
                            s.H, s.W)
    xys, conics, cov2d, z, radii = p["xys"], p["conics"], p["cov2d"], p["cam_depths"], p["radii"]
    op = s.opacities.reshape(-1)
    ok = radii > 0
    sqx = 3.0 * np.sqrt(cov2d[:, 0, 0])
    sqy = 3.0 * np.sqrt(cov2d[:, 1, 1])
    x0 = np.maximum(0, np.floor(xys[:, 0] - sqx).astype(np.int64) - 2)
    x1 = np.minimum(s.W, np.ceil(xys[:, 0] + sqx).astype(np.int64) + 2)
    y0 = np.maximum(0, np.floor(xys[:, 1] - sqy).astype(np.int64) - 2)
    y1 = np.minimum(s.H, np.ceil(xys[:, 1] + sqy).astype(np.int64) + 2)
    # gs_pack_splats: the rectangle is intersected with the bounding box of the sigma <= sigma_max ellipse
    with np.errstate(divide="ignore", invalid="ignore"):
        smax = np.log(255.0 * op) + 2e-3
        A, B, Cc = conics[:, 0], conics[:, 1], conics[:, 2]
        det = A * Cc - B * B
        um = np.sqrt(np.maximum(2.0 * smax * Cc / det, 0.0))
        vm = np.sqrt(np.maximum(2.0 * smax * A / det, 0.0))
    good = (smax >= 0) & (det > 0) & np.isfinite(um) & np.isfinite(vm)
    ok &= smax >= 0
    ex0 = np.where(good, np.ceil(xys[:, 0] - um - 1e-3), -1e9).astype(np.int64)
    ex1 = np.where(good, np.floor(xys[:, 0] + um + 1e-3) + 1, 1e9).astype(np.int64)
    ey0 = np.where(good, np.ceil(xys[:, 1] - vm - 1e-3), -1e9).astype(np.int64)
    ey1 = np.where(good, np.floor(xys[:, 1] + vm + 1e-3) + 1, 1e9).astype(np.int64)
    x0, x1, y0, y1 = np.maximum(x0, ex0), np.minimum(x1, ex1), np.maximum(y0, ey0), np.minimum(y1, ey1)
    wx0, wy0, wx1, wy1 = 16 * tx0, 16 * ty0, 16 * (tx0 + ntx), 16 * (ty0 + nty)
    sel = ok & (x1 > wx0) & (x0 < wx1) & (y1 > wy0) & (y0 < wy1) & (x1 > x0) & (y1 > y0)
    idx = np.nonzero(sel)[0]
    idx = idx[np.argsort(z[idx], kind="stable")]
    return dict(idx=idx, xys=xys[idx], conics=conics[idx], op=op[idx], x0=x0[idx], x1=x1[idx], y0=y0[idx],
                y1=y1[idx])


def tile_work(g, tx, ty):
    """Per tile: need0 [L,16,16] (sigma <= sigma_max inside the rectangle), last [16,16]."""
    X0, Y0 = 16 * tx, 16 * ty
    m = (g["x1"] > X0) & (g["x0"] < X0 + 16) & (g["y1"] > Y0) & (g["y0"] < Y0 + 16)
    e = np.nonzero(m)[0]
    L = len(e)
    if L == 0:
        return None
    px = (X0 + np.arange(16, dtype=np.float32))[None, None, :]
    py = (Y0 + np.arange(16, dtype=np.float32))[None, :, None]
    cx = g["xys"][e, 0][:, None, None]
    cy = g["xys"][e, 1][:, None, None]
    A = g["conics"][e, 0][:, None, None]
    B = g["conics"][e, 1][:, None, None]
    Cc = g["conics"][e, 2][:, None, None]
    o = g["op"][e][:, None, None]
    dx = cx - px
    dy = cy - py
    sig = 0.5 * (A * dx * dx + Cc * dy * dy) + B * dx * dy
    inr = ((px >= g["x0"][e][:, None, None]) & (px < g["x1"][e][:, None, None]) &
           (py >= g["y0"][e][:, None, None]) & (py < g["y1"][e][:, None, None]))
    with np.errstate(divide="ignore", invalid="ignore"):
        smax = np.log(255.0 * o) + 2e-3
    need0 = inr & (sig >= 0) & (sig <= smax)
    # forward compositing: last contributor per pixel (gsplat_cpu.cpp:220-236)
    alpha = np.minimum(0.999, o * np.exp(-sig))
    contrib = need0 & (alpha >= 1.0 / 255.0)
    T = np.ones((16, 16), np.float64)
    done = np.zeros((16, 16), bool)
    last = np.full((16, 16), -1, np.int64)
    for i in range(L):
        c = contrib[i] & ~done
        nT = T * (1.0 - alpha[i])
        sat = c & (nT <= 1e-4)
        done |= sat
        c &= ~sat
        T = np.where(c, nT, T)
        last = np.where(c, i, last)
    return need0, last


def block16(need0):
    """[L,16]: bit 4 r + c — the entry reaches the 4x4 block at block column c, block row r."""
    L = need0.shape[0]
    return need0.reshape(L, 4, 4, 4, 4).any(axis=(2, 4)).reshape(L, 16)


def sim_tile(need0, last, stats):
    L = need0.shape[0]
    b16 = block16(need0)
    need = need0 & (np.arange(L)[:, None, None] <= last[None])
    stats["M"] += L
    stats["needed"] += int(need.sum())
    stats["empty_mask_entries"] += int((~b16.any(axis=1)).sum())
    wave_last = int(last.max())
    if wave_last < 0:
        return
    # ---- cur: 8x8 blocks, 4 px / lane (lane = column, rows r, r+2, r+4, r+6 of the block) ----
    b8 = b16.reshape(L, 2, 2, 2, 2).any(axis=(2, 4)).reshape(L, 4)          # [L, (R, C)]
    gl8 = last.reshape(2, 8, 2, 8).max(axis=(1, 3)).reshape(4)
    need8 = need.reshape(L, 2, 8, 2, 8)                                       # [L, R, y, C, x]
    for hi in range(wave_last, -1, -64):
        lo = max(hi - 63, 0)
        ent = np.arange(hi, lo - 1, -1)
        lists = []
        for gidx in range(4):
            R, Cb = gidx >> 1, gidx & 1
            t = ent[b8[ent, gidx] & (ent <= gl8[gidx])]
            lists.append((R, Cb, t))
        n = max(len(t) for _, _, t in lists)
        stats["cur_chunks"] += 1
        stats["cur_pairs"] += sum(len(t) for _, _, t in lists)
        stats["cur_steps"] += n
        for k in range(n):
            # pass p covers block rows {2p, 2p+1} (LH = 2): live lanes per pass
            for p in range(4):
                live = 0
                for R, Cb, t in lists:
                    if k < len(t):
                        live += int(need8[t[k], R, 2 * p:2 * p + 2, Cb, :].sum())
                if live:
                    stats["cur_passes"] += 1
                    stats["cur_live"] += live
    # ---- q: 4x4 blocks, 16 groups of 4 lanes, 4 px / lane ----
    gl4 = last.reshape(4, 4, 4, 4).max(axis=(1, 3)).reshape(16)
    for name, chunk in (("q32", 32), ("q48", 48), ("q64", 64), ("q128", 128), ("q256", 256)):
        for hi in range(wave_last, -1, -chunk):
            lo = max(hi - chunk + 1, 0)
            ent = np.arange(hi, lo - 1, -1)
            cnt = [int((b16[ent, gidx] & (ent <= gl4[gidx])).sum()) for gidx in range(16)]
            stats[name + "_chunks"] += 1
            stats[name + "_pairs"] += sum(cnt)
            stats[name + "_steps"] += max(cnt)
            stats[name + "_maxpairs"] = max(stats[name + "_maxpairs"], sum(cnt))
            stats[name + "_pairs_gt256"] += int(sum(cnt) > 256)
            stats[name + "_pairs_gt320"] += int(sum(cnt) > 320)
            stats[name + "_maxpc"] += int(b16[ent].sum(axis=1).max()) if len(ent) else 0
    # ---- o8: 8 groups of 8 lanes, a group = a 4x8 block (4 wide, 8 high), 4 px / lane (lane = (column, row parity)) ----
    b48 = b16.reshape(L, 2, 2, 4).any(axis=2).reshape(L, 8)                   # [L, (R, c)]
    gl48 = last.reshape(2, 8, 4, 4).max(axis=(1, 3)).reshape(8)
    for name, chunk in (("o64", 64), ("o128", 128)):
        for hi in range(wave_last, -1, -chunk):
            lo = max(hi - chunk + 1, 0)
            ent = np.arange(hi, lo - 1, -1)
            cnt = [int((b48[ent, gidx] & (ent <= gl48[gidx])).sum()) for gidx in range(8)]
            stats[name + "_chunks"] += 1
            stats[name + "_pairs"] += sum(cnt)
            stats[name + "_steps"] += max(cnt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--tiles", default="20x12")
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    if a.config == "c2":
        s = scenes.config_c2(a.n or 1_000_000)
    elif a.config == "c3":
        s = scenes.config_c3(a.n or 5_000_000)
    else:
        s = scenes.camera_scene(a.n or 1_000_000, 1920, 1080, K=16, seed=1, sigma_px=(0.5, 4.0), hot=(0.02, 48))
    ntx, nty = (int(v) for v in a.tiles.split("x"))
    TX, TY = (s.W + 15) // 16, (s.H + 15) // 16
    tx0, ty0 = (TX - ntx) // 2, (TY - nty) // 2
    g = build_window(s, tx0, ty0, ntx, nty)
    from collections import defaultdict
    st = defaultdict(int)
    for ty in range(ty0, ty0 + nty):
        for tx in range(tx0, tx0 + ntx):
            r = tile_work(g, tx, ty)
            if r is not None:
                sim_tile(r[0], r[1], st)
    M, needed = st["M"], st["needed"]
    out = dict(config=a.config, tiles=a.tiles, M=M, needed_evals=needed, needed_per_entry=needed / M,
               empty_mask_fraction=st["empty_mask_entries"] / M)
    # instruction model (gs_raster.hip): a pass is 31 VALU; cur: + 21 (row_reduce9) + 14 (walk, record, atomics'
    # addresses) per step; q: + 14 (two-stage reduce) + 14; chunk overhead (staging, queues, flush) per chunk
    cur_valu = st["cur_passes"] * 31 + st["cur_steps"] * 35 + st["cur_chunks"] * 120
    out["cur"] = dict(steps_per_entry=st["cur_steps"] / M, pairs_per_entry=st["cur_pairs"] / M,
                      passes_per_step=st["cur_passes"] / max(st["cur_steps"], 1),
                      live_lanes_per_pass=st["cur_live"] / max(st["cur_passes"], 1),
                      valu_per_needed=cur_valu / needed, valu_per_entry=cur_valu / M)
    for name, groups in (("q32", 16), ("q48", 16), ("q64", 16), ("q128", 16), ("q256", 16), ("o64", 8), ("o128", 8)):
        steps, pairs, chunks = st[name + "_steps"], st[name + "_pairs"], st[name + "_chunks"]
        per_step = 4 * 31 + (14 if groups == 16 else 18) + 14
        per_chunk = 120 + (5 * groups) * max(int(name[1:]) // 64, 1)
        valu = steps * per_step + chunks * per_chunk
        out[name] = dict(steps_per_entry=steps / M, pairs_per_entry=pairs / M,
                         fill=pairs / max(groups * steps, 1),
                         useful_lane_frac=needed / max(steps * 256, 1),
                         valu_per_needed=valu / needed, valu_per_entry=valu / M,
                         vs_cur=valu / cur_valu, chunks_per_entry=chunks / M, steps_per_chunk=steps / max(chunks, 1),
                         pairs_per_chunk=pairs / max(chunks, 1), max_pairs=st[name + "_maxpairs"],
                         frac_chunks_over_256=st[name + "_pairs_gt256"] / max(chunks, 1),
                         frac_chunks_over_320=st[name + "_pairs_gt320"] / max(chunks, 1),
                         mean_max_blocks_per_entry=st[name + "_maxpc"] / max(chunks, 1))
    print(json.dumps(out, indent=1))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()


def claim_rounds(config="c2", tiles="12x8"):
    """Rounds of the claim (read-add-write by the group that holds the entry) per step, for one accumulator
    copy, two (checkerboard of blocks) and four (2 x 2 parity of the block coordinates)."""
    s = scenes.config_c2() if config == "c2" else scenes.config_c3()
    ntx, nty = (int(v) for v in tiles.split("x"))
    TX, TY = (s.W + 15) // 16, (s.H + 15) // 16
    tx0, ty0 = (TX - ntx) // 2, (TY - nty) // 2
    g = build_window(s, tx0, ty0, ntx, nty)
    tot = {1: 0, 2: 0, 4: 0}
    steps = 0
    for ty in range(ty0, ty0 + nty):
        for tx in range(tx0, tx0 + ntx):
            r = tile_work(g, tx, ty)
            if r is None:
                continue
            need0, last = r
            L = need0.shape[0]
            b16 = block16(need0)
            wave_last = int(last.max())
            if wave_last < 0:
                continue
            gl4 = last.reshape(4, 4, 4, 4).max(axis=(1, 3)).reshape(16)
            for hi in range(wave_last, -1, -64):
                lo = max(hi - 63, 0)
                ent = np.arange(hi, lo - 1, -1)
                qs = [ent[b16[ent, gi] & (ent <= gl4[gi])] for gi in range(16)]
                n = max(len(q) for q in qs)
                for k in range(n):
                    steps += 1
                    for banks in (1, 2, 4):
                        cl = {}
                        for gi in range(16):
                            if k < len(qs[gi]):
                                br, bc = gi >> 2, gi & 3
                                bank = 0 if banks == 1 else ((br + bc) & 1) if banks == 2 else ((br & 1) * 2 + (bc & 1))
                                key = (bank, int(qs[gi][k]))
                                cl[key] = cl.get(key, 0) + 1
                        tot[banks] += max(cl.values()) if cl else 0
    return {b: tot[b] / max(steps, 1) for b in tot}


def claim_rounds_delayed(config="c2", tiles="8x6", delays=(0, 1, 2, 3, 4)):
    """Round 6: rounds of the claim per step with ONE accumulator copy when the groups of one checkerboard colour
    (or of the 2 x 2 parity classes) start d (or d * class) steps late — neighbouring blocks, whose queues hold the
    same Gaussians at nearly the same ranks, then reach an entry at different steps.  Also the steps per chunk the
    delay adds.  -> {name: (rounds per step, steps relative to no delay)}"""
    s = scenes.config_c2() if config == "c2" else scenes.config_c3()
    ntx, nty = (int(v) for v in tiles.split("x"))
    TX, TY = (s.W + 15) // 16, (s.H + 15) // 16
    tx0, ty0 = (TX - ntx) // 2, (TY - nty) // 2
    g = build_window(s, tx0, ty0, ntx, nty)
    names = [("cb%d" % d, 2, d) for d in delays] + [("p4_%d" % d, 4, d) for d in delays if d]
    rounds = {n: 0 for n, _, _ in names}
    steps = {n: 0 for n, _, _ in names}
    for ty in range(ty0, ty0 + nty):
        for tx in range(tx0, tx0 + ntx):
            r = tile_work(g, tx, ty)
            if r is None:
                continue
            need0, last = r
            b16 = block16(need0)
            wave_last = int(last.max())
            if wave_last < 0:
                continue
            gl4 = last.reshape(4, 4, 4, 4).max(axis=(1, 3)).reshape(16)
            for hi in range(wave_last, -1, -64):
                lo = max(hi - 63, 0)
                ent = np.arange(hi, lo - 1, -1)
                qs = [ent[b16[ent, gi] & (ent <= gl4[gi])] for gi in range(16)]
                for name, classes, d in names:
                    off = []
                    for gi in range(16):
                        br, bc = gi >> 2, gi & 3
                        cls = ((br + bc) & 1) if classes == 2 else ((br & 1) * 2 + (bc & 1))
                        off.append(cls * d)
                    n = max((len(q) + off[gi]) if len(q) else 0 for gi, q in enumerate(qs))
                    for k in range(n):
                        cl = {}
                        for gi in range(16):
                            kk = k - off[gi]
                            if 0 <= kk < len(qs[gi]):
                                e = int(qs[gi][kk])
                                cl[e] = cl.get(e, 0) + 1
                        steps[name] += 1
                        rounds[name] += max(cl.values()) if cl else 0
    base = steps["cb0"]
    return {n: (rounds[n] / max(steps[n], 1), steps[n] / max(base, 1)) for n in rounds}
