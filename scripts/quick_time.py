import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from opensplat_amd import scenes, cabi
from tests.util import to_dev
import os
print(open('/proc/self/maps').read().count('libamdhip64'), [l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l][:3])
for (N,W,H,sp) in [(200000,1920,1080,(0.5,4.0)), (1000000,1920,1080,(0.5,4.0))]:
    s = scenes.camera_scene(N, W, H, K=16, seed=1, sigma_px=sp)
    cam = cabi.make_camera(s.viewmat, s.projmat, s.fx, s.fy, s.cx, s.cy, s.W, s.H)
    means, scales, quats = to_dev(s.means), to_dev(s.scales), to_dev(s.quats)
    opac = to_dev(s.opacities.reshape(-1)); dirs, coeffs = to_dev(s.dirs), to_dev(s.sh_coeffs); v_out = to_dev(s.v_out)
    ws = cabi.BinWorkspace()
    def step(flags=0, timing=None):
        ev = lambda: torch.cuda.Event(enable_timing=True)
        marks = []
        def mark(name):
            e = ev(); e.record(); marks.append((name, e))
        mark('start')
        p = cabi.project_forward(cam, means, scales, quats); mark('proj_fwd')
        rgb = cabi.sh_forward(3, dirs, coeffs); colors = torch.clamp_min(rgb + 0.5, 0.0); mark('sh_fwd')
        b = cabi.bin_and_sort(W, H, p['xys'], p['depths'], p['radii'], p['conics'], colors, opac, p['cov2d'], ws, keep_unsorted=False); mark('bin')
        f = cabi.rasterize_forward(W, H, b, s.background, flags); mark('rast_fwd')
        g = cabi.rasterize_backward(W, H, N, b, s.background, f['final_Ts'], f['final_idx'], v_out, flags); mark('rast_bwd')
        vrgb = (g['v_colors'] * (rgb + 0.5 > 0).float()).contiguous()
        vc = cabi.sh_backward(3, 16, dirs, vrgb); mark('sh_bwd')
        pb = cabi.project_backward(cam, means, scales, quats, p['radii'], g['v_xy'], g['v_conic']); mark('proj_bwd')
        return b.num_isects, marks
    for flags in (0, 1):
        for _ in range(3): M, marks = step(flags)
        torch.cuda.synchronize(); t0 = time.time()
        K = 10
        allm = []
        for _ in range(K): M, marks = step(flags); allm.append(marks)
        torch.cuda.synchronize(); dt = (time.time() - t0) / K
        agg = {}
        for marks in allm:
            for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
                agg[n1] = agg.get(n1, 0) + e0.elapsed_time(e1) / K
        print(f"N={N} flags={flags} M={M} step={dt*1e3:.3f} ms  ({1/dt:.1f} it/s)  " + " ".join(f"{k}={v:.3f}" for k, v in agg.items()), flush=True)
