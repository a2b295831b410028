#!/usr/bin/env python3
"""One-off stress of the compositing kernels beyond the 15 seeded cases of tests/test_gpu_parity.py:
random configurations START..END, forward bit-exact against the oracle (image, final_Ts, last
contributor), backward within 2e-5 with 1, 2 and 4 pixels per lane and in the deterministic mode; the
checkpointed forward bit for bit and the backward in pieces (random piece length / record count / pixels per
lane) within 2e-5 of the oracle too.
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
    worst_pieces = 0.0
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
        # (px 16: sixteen four-lane groups per wave, flag bit 25 — the full-frame default since round 5)
        for px, extra in ((0, 0), (1, 0), (2, 0), (4, 0), (4, cabi.GS_FLAG_DETERMINISTIC), (2, cabi.GS_FLAG_DETERMINISTIC),
                          (16, 0), (16, cabi.GS_FLAG_DETERMINISTIC)):
            flag = ((1 << 25) if px == 16 else {1: 1, 2: 2, 4: 3}[px] << 21 if px else 0) | extra
            gr = cabi.rasterize_backward(s.W, s.H, s.N, out["binned"], s.background, out["final_Ts"],
                                         out["final_idx"], to_dev(s.v_out), flag)
            torch.cuda.synchronize()
            for k in ("v_xy", "v_conic", "v_colors", "v_opacity"):
                ref = g[k].reshape(np_(gr[k]).shape)
                e = rel_err(np_(gr[k]), ref)
                worst = max(worst, e)
                if not e < 2e-5:
                    bad.append((i, "backward px=%d flags=%d %s %.3g" % (px, extra, k, e)))
        # round 4: the checkpointed forward (same bits, one / two entries per step) and the backward in pieces —
        # piece length, records per tile (also fewer than the lists need: the last piece takes the rest) and
        # pixels per lane drawn per case
        rs = np.random.RandomState(1000 + i)
        seg_len = int(rs.choice([64, 128, 256]))
        bins = np_(out["binned"].tile_bins)
        longest = int((bins[:, 1] - bins[:, 0]).max()) if len(bins) else 0
        need = longest // seg_len + 2
        max_segments = max(2, int(rs.choice([need, max(need // 2, 2), need + 3])))
        ck = cabi.Checkpoints()
        ck.plan(s.W, s.H, None, "cuda:0", seg_len=seg_len, max_segments=max_segments)
        ck.buf.fill_(0xFF)
        ilp = int(rs.choice([1, 2])) << 23
        f2 = cabi.rasterize_forward(s.W, s.H, out["binned"], s.background, ilp, checkpoints=ck)
        torch.cuda.synchronize()
        if not (torch.equal(f2["img"], out["img"]) and torch.equal(f2["final_Ts"], out["final_Ts"])
                and torch.equal(f2["final_idx"], out["final_idx"])):
            bad.append((i, "checkpointed forward, flags %#x" % ilp))
        for px in (0, 1, 2, 4):
            flag = ({1: 1, 2: 2, 4: 3}[px] << 21) if px else 0
            gr = cabi.rasterize_backward(s.W, s.H, s.N, out["binned"], s.background, f2["final_Ts"], f2["final_idx"],
                                         to_dev(s.v_out), flag | (cabi.GS_FLAG_DETERMINISTIC if i % 3 == 0 else 0),
                                         checkpoints=ck)
            torch.cuda.synchronize()
            for k in ("v_xy", "v_conic", "v_colors", "v_opacity"):
                ref = g[k].reshape(np_(gr[k]).shape)
                e = rel_err(np_(gr[k]), ref)
                worst_pieces = max(worst_pieces, e)
                if not (np.isfinite(np_(gr[k])).all() and e < 2e-5):
                    bad.append((i, "pieces px=%d S=%d segs=%d %s %.3g" % (px, seg_len, max_segments, k, e)))
    print({"cases": b - a, "failures": bad, "worst_backward_rel_err": worst,
           "worst_backward_rel_err_in_pieces": worst_pieces})
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
