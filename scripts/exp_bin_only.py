"""Timing harness for the binning stage alone (measurement script, not a test): gs_gaussian_forward once, then
`--iters` speculative binning calls (gs_bin_speculative: memset, k_count_tiles, k_scatter_scan, k_bucket_sort_*) and
nothing behind them — so library variants whose lists are WRONG on purpose (scripts/build_variant.sh with
-DGS_EXP_SCATTER=n: what the scatter kernel's time hangs on) can be timed without a compositing kernel reading them.

    GSPLAT_HIP_LIB=.../libgsplat_hip_<variant>.so rocprofv3 --kernel-trace --stats -d out -- \
        python scripts/exp_bin_only.py --config c2 --iters 200

Prints one JSON line: HIP-event time per binning call (the stage), M, longest list.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "hot", "c3hot"])
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    import torch

    from opensplat_amd import cabi, scenes
    from opensplat_amd.pipeline import HotPath

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cabi.lib()
    if args.config in ("c3", "c3hot"):
        scene = scenes.camera_scene(5_000_000, 3840, 2160, K=16, seed=2, sigma_px=(1.0, 8.0), name="C3",
                                    hot=(0.004 if args.config == "c3hot" else 0.0, 48))
    else:
        scene = scenes.camera_scene(1_000_000, 1920, 1080, K=16, seed=1, sigma_px=(0.5, 4.0), name="C2",
                                    hot=(0.02 if args.config == "hot" else 0.0, 48))
    hp = HotPath(scene, dev, 0)
    s = scene
    g = cabi.gaussian_forward(hp.cam, hp.means, hp.scales, hp.quats, hp.opac, hp.features_dc, hp.features_rest,
                              hp.cam_pos, s.degrees_to_use, 0, out=hp.gfwd, viewmat_dev=hp.vm_dev,
                              projmat_dev=hp.pm_dev)

    def bin_once():
        return cabi.bin_and_sort(s.W, s.H, None, g["depths"], None, None, None, None, None, hp.ws,
                                 speculative=True, packed=g["packed"])

    for _ in range(3):       # the id list's capacity settles after the first call
        b = bin_once()
        torch.cuda.synchronize()
        cabi.validate_binning(b)
    for _ in range(20):
        bin_once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        b = bin_once()
    e1.record()
    torch.cuda.synchronize()
    ls = hp.ws.list_stats
    print(json.dumps({"config": args.config, "lib": os.environ.get("GSPLAT_HIP_LIB", "in-tree"),
                      "bin_stage_us": 1000.0 * e0.elapsed_time(e1) / args.iters,
                      "M": int(ls[0]), "longest": int(ls[1])}))


if __name__ == "__main__":
    main()
