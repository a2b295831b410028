"""The C2 step as ONE HIP graph: launch overhead of the eleven launches of a step against the same
launches replayed from a captured graph (torch.cuda.CUDAGraph = hipGraph on ROCm).

    python scripts/bench_graph.py [--steps 200] [--out profiles/graph_r03.json]

The step is the timed path of bench.py (Pipeline.step_fused) without the host-side validation of the
speculative id-list capacity, which a captured stream cannot contain: the count is checked once after the
replays instead (static camera: the capacity validated during warm-up holds)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--out", default=None)
    ap.add_argument("--small", action="store_true", help="6000 Gaussians at 384x288 (the launch-bound regime)")
    a = ap.parse_args()
    import torch

    from opensplat_amd.pipeline import HotPath as Pipeline
    from opensplat_amd import cabi, scenes

    dev = torch.device("cuda", 0)
    s = scenes.camera_scene(6000, 384, 288, K=16, seed=4) if a.small else scenes.config_c2()
    pipe = Pipeline(s, dev, 0)
    for _ in range(5):
        pipe.step()
    torch.cuda.synchronize()
    KEEP = cabi.GS_FLAG_KEEP_RECORDS

    def launches():
        g = cabi.gaussian_forward(pipe.cam, pipe.means, pipe.scales, pipe.quats, pipe.opac, pipe.features_dc,
                                  pipe.features_rest, pipe.cam_pos, s.degrees_to_use, 0, out=pipe.gfwd,
                                  viewmat_dev=pipe.vm_dev, projmat_dev=pipe.pm_dev)
        b = cabi.bin_and_sort(s.W, s.H, None, g["depths"], None, None, None, None, None, pipe.ws,
                              speculative=True, packed=g["packed"])
        f = cabi.rasterize_forward(s.W, s.H, b, pipe.background, 0, out=pipe.fwd)
        cabi.rasterize_backward(s.W, s.H, s.N, b, pipe.background, f["final_Ts"], f["final_idx"], pipe.v_out,
                                KEEP, workspace=pipe.bwd_ws)
        cabi.gaussian_backward(pipe.cam, pipe.means, pipe.scales, pipe.quats, pipe.opac, pipe.cam_pos, s.K,
                               s.degrees_to_use, g["radii"], g["rgb_raw"], pipe.bwd_ws, pipe.gout, 0,
                               viewmat_dev=pipe.vm_dev, projmat_dev=pipe.pm_dev)
        return b

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    ref_grads = None
    ms_stream = timed(launches, a.steps)
    ref_grads = pipe.grads.flat.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            launches()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        b = launches()
    ms_graph = timed(graph.replay, a.steps)
    M = int(b.m_host[0])
    assert M <= b.capacity, (M, b.capacity)
    # (fp32 atomics: the summation order differs from run to run; compare against the largest gradient)
    err = float((pipe.grads.flat - ref_grads).abs().max() / ref_grads.abs().max())
    same = err < 1e-5
    out = {"workload": ("6000 Gaussians at 384x288" if a.small else "C2") +
                       " step (bench.py's timed launches without the per-step host validation), static camera",
           "steps": a.steps, "ms_per_step_stream_launches": ms_stream, "ms_per_step_graph_replay": ms_graph,
           "rasterizations_per_s_stream": 1e3 / ms_stream, "rasterizations_per_s_graph": 1e3 / ms_graph,
           "intersections": M, "gradients_match": same, "gradients_max_rel_diff": err}
    print(json.dumps(out))
    if a.out:
        prev = json.load(open(a.out)) if os.path.exists(a.out) else {}
        prev["small" if a.small else "c2"] = out
        json.dump(prev, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
