import sys, os, json, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from opensplat_amd.pipeline import HotPath as Pipeline  # noqa: E402
from opensplat_amd import scenes, cabi
for (N, sig) in [(100000, (8.0, 40.0)), (300000, (2.0, 16.0)), (10000, (20.0, 120.0))]:
    s = scenes.camera_scene(N, 1920, 1080, K=16, seed=3, sigma_px=sig)
    pipe = Pipeline(s, torch.device('cuda:0'), 0)
    for _ in range(3): pipe.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): pipe.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    ev = []; pipe.step(ev); torch.cuda.synchronize()
    st = {n: round(ev[k].elapsed_time(ev[k+1]), 3) for k, n in enumerate(pipe.stage_names)}
    print(N, sig, 'M', pipe.num_isects, 'ms/step', round(ms, 3), st, 'stats', list(pipe.ws.list_stats))
