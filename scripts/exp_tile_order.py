#!/usr/bin/env python3
"""EXPERIMENT (round 6): what the compositing launches gain from the ORDER their tiles are started in.

Runs the C2 (or C3 / hot-spot) step once, then re-times k_rasterize_forward / k_rasterize_backward alone with the
tile order replaced by: the exact longest-first order of the tile-level scan (counting sort on the list length),
the strip-adjacent order (strips by intersection count, a strip's tiles side by side), the same with the strips'
tiles interleaved (tile j of every strip, then tile j + 1), a random permutation, and the identity (row-major).
    python scripts/exp_tile_order.py [c2|c3|hot]
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from opensplat_amd.pipeline import HotPath as Pipeline  # noqa: E402
from opensplat_amd import cabi, scenes  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c2"
    s = scenes.config_c3() if which == "c3" else scenes.config_c2()
    if which == "hot":
        s = scenes.camera_scene(1_000_000, 1920, 1080, K=16, seed=1, sigma_px=(0.5, 4.0), name="C2", hot=(0.02, 48))
    dev = torch.device("cuda:0")
    pipe = Pipeline(s, dev, 0)
    for _ in range(3):
        pipe.step()
    torch.cuda.synchronize()
    g = cabi.gaussian_forward(pipe.cam, pipe.means, pipe.scales, pipe.quats, pipe.opac, pipe.features_dc,
                              pipe.features_rest, pipe.cam_pos, s.degrees_to_use, 0, out=pipe.gfwd,
                              viewmat_dev=pipe.vm_dev, projmat_dev=pipe.pm_dev)
    b = cabi.bin_and_sort(s.W, s.H, None, g["depths"], None, None, None, None, None, pipe.ws, speculative=True,
                          packed=g["packed"])
    f = cabi.rasterize_forward(s.W, s.H, b, pipe.background, 0, out=pipe.fwd)
    assert cabi.validate_binning(b)
    torch.cuda.synchronize()
    tiles_x, tiles_y = (s.W + 15) // 16, (s.H + 15) // 16
    tiles = tiles_x * tiles_y
    bins = b.tile_bins.view(-1)[: 2 * tiles].view(tiles, 2).long()
    lens = bins[:, 1] - bins[:, 0]
    t = torch.arange(tiles, device=dev)
    strip = (t // tiles_x) * ((tiles_x + 15) // 16) + (t % tiles_x) // 16
    n_strips = int(strip.max()) + 1
    strip_tot = torch.zeros(n_strips, device=dev, dtype=torch.long).scatter_add_(0, strip, lens)
    orders = {}
    orders["exact_lpt"] = torch.argsort(lens, descending=True, stable=True)
    # strips by total, tiles of a strip adjacent (longest first)
    key = strip_tot[strip] * (1 << 40) + (n_strips - strip) * (1 << 20) + lens
    orders["strip_adjacent"] = torch.argsort(key, descending=True, stable=True)
    # interleaved: rank of the tile inside its strip is the major key, the strip's total the minor one
    o = orders["strip_adjacent"]
    pos_in_strip = torch.zeros(tiles, device=dev, dtype=torch.long)
    so = strip[o]
    first = torch.ones(tiles, dtype=torch.bool, device=dev)
    first[1:] = so[1:] != so[:-1]
    start_idx = torch.cummax(torch.where(first, torch.arange(tiles, device=dev), torch.zeros_like(so)), 0)[0]
    pos_in_strip[o] = torch.arange(tiles, device=dev) - start_idx
    key2 = (16 - pos_in_strip) * (1 << 50) + strip_tot[strip] * (1 << 14) + strip
    orders["strip_interleaved"] = torch.argsort(key2, descending=True, stable=True)
    gen = torch.Generator(device="cpu").manual_seed(0)
    orders["random"] = torch.randperm(tiles, generator=gen).to(dev)
    orders["row_major"] = t.clone()
    # LPT on lengths rounded down to 32 entries, random inside a class
    orders["lpt32_random"] = torch.argsort((lens // 32) * (1 << 20) + orders["random"].argsort(), descending=True)
    orders["library"] = b.tile_order.long().clone()
    rnd = orders["random"].argsort()        # a random rank per tile
    for w in (4, 8, 16, 64, 128, 256):
        orders["lpt%d_random" % w] = torch.argsort((lens // w) * (1 << 20) + rnd, descending=True)
    # deterministic scatter inside a class: multiplicative hash of the tile index
    mult = int(tiles * 0.6180339887) | 1
    while np.gcd(mult, tiles) != 1:
        mult += 2
    scat = (t * mult) % tiles
    orders["lpt32_mult"] = torch.argsort((lens // 32) * (1 << 20) + scat, descending=True)
    # inside a class: every XCD (slot % 8) keeps to its own band of tile rows, random inside the band
    band = (t // tiles_x) * 8 // tiles_y
    for w in (32,):
        cls = lens // w
        k = cls * (1 << 24) + rnd            # random order inside the class first
        o1 = torch.argsort(k, descending=True)
        # stable regroup inside each class: rank of the tile among its (class, band) mates, then band
        cb = cls[o1] * 8 + band[o1]
        o2 = o1[torch.argsort(cb, stable=True, descending=True)]
        cb2 = cls[o2] * 8 + band[o2]
        firstcb = torch.ones(tiles, dtype=torch.bool, device=dev)
        firstcb[1:] = cb2[1:] != cb2[:-1]
        st = torch.cummax(torch.where(firstcb, torch.arange(tiles, device=dev), torch.zeros_like(cb2)), 0)[0]
        r_in = torch.arange(tiles, device=dev) - st
        k3 = cls[o2] * (1 << 24) - r_in * 8 - band[o2] + (1 << 23)
        orders["lpt%d_xcdband" % w] = o2[torch.argsort(k3, descending=True, stable=True)]
    out = {"config": which, "bin_mode": cabi._BIN_MODE, "M": int(b.num_isects), "rows": {}}
    KEEP = cabi.GS_FLAG_KEEP_RECORDS
    for name, od in orders.items():
        assert torch.equal(torch.sort(od)[0], t)
        b.tile_order.copy_(od.to(torch.int32))
        ms = {"fwd": [], "bwd": []}
        for _ in range(12):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            for x in e:
                x.record()          # (the hook needs events whose HIP handles exist)
            cabi.time_next_kernel(e[0], e[1])
            f = cabi.rasterize_forward(s.W, s.H, b, pipe.background, 0, out=pipe.fwd)
            cabi.time_next_kernel(e[2], e[3])
            cabi.rasterize_backward(s.W, s.H, s.N, b, pipe.background, f["final_Ts"], f["final_idx"], pipe.v_out,
                                    KEEP, workspace=pipe.bwd_ws)
            torch.cuda.synchronize()
            ms["fwd"].append(e[0].elapsed_time(e[1]))
            ms["bwd"].append(e[2].elapsed_time(e[3]))
        out["rows"][name] = {k: round(float(np.median(v[2:])) * 1e3, 1) for k, v in ms.items()}
        print(name, out["rows"][name], file=sys.stderr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
