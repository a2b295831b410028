#!/usr/bin/env python3
"""One-off stress of the compositing kernels beyond the 15 seeded cases of tests/test_gpu_parity.py:
random configurations START..END, forward bit-exact against the oracle (image, final_Ts, last
contributor), backward within 2e-5 with 1, 2 and 4 pixels per lane and in the deterministic mode.
Test infrastructure (uses oracle/ like the tests do).   python tests/extended_sweep.py [START END]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle  # noqa: E402
from opensplat_amd import cabi  # noqa: E402
from tests.test_gpu_parity import _random_scene  # noqa: E402
from tests.util import hip_pipeline, np_, oracle_raster, rel_err, to_dev  # noqa: E402


def main():
    a, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (15, 135)
    restated = oracle.restated()
    bad = []
    worst = 0.0
    for i in range(a, b):
        s = _random_scene(i)
        out = hip_pipeline(s)
        f, g = oracle_raster(restated, s, np_(out["xys"]), np_(out["conics"]), np_(out["colors"]),
                             np_(out["cov2d"]), np_(out["depths"]), v_out=s.v_out)
        ok = np.array_equal(np_(out["img"]), f["img"]) and np.array_equal(np_(out["final_Ts"]), f["final_Ts"])
        fi = np_(out["final_idx"]).ravel()
        counts = f["px_counts"].ravel()
        offs = np.concatenate([[0], np.cumsum(counts)])[:-1]
        has = counts > 0
        ok = ok and np.array_equal(fi >= 0, has)
        ids_sorted = np_(out["binned"].gaussian_ids_sorted)
        ok = ok and np.array_equal(ids_sorted[fi[has]], f["contributors"][offs[has]])
        if not ok:
            bad.append((i, "forward"))
        for px, extra in ((0, 0), (1, 0), (2, 0), (4, 0), (4, cabi.GS_FLAG_DETERMINISTIC), (2, cabi.GS_FLAG_DETERMINISTIC)):
            flag = ({1: 1, 2: 2, 4: 3}[px] << 21 if px else 0) | extra
            gr = cabi.rasterize_backward(s.W, s.H, s.N, out["binned"], s.background, out["final_Ts"],
                                         out["final_idx"], to_dev(s.v_out), flag)
            torch.cuda.synchronize()
            for k in ("v_xy", "v_conic", "v_colors", "v_opacity"):
                ref = g[k].reshape(np_(gr[k]).shape)
                e = rel_err(np_(gr[k]), ref)
                worst = max(worst, e)
                if not e < 2e-5:
                    bad.append((i, "backward px=%d flags=%d %s %.3g" % (px, extra, k, e)))
    print({"cases": b - a, "failures": bad, "worst_backward_rel_err": worst})
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
