"""-m gpu: the full-frame forward that compacts its chunks (k_rasterize_forward_c, round 6; flag bits 23..24 = 3)
against the default forward on the same lists: image, final_Ts and final_idx — the LIST index of the last composited
entry, which the backward starts from — must agree bit for bit, whatever the share of a tile's list a quadrant
touches, for lists shorter than a chunk, empty lists, lists of several thousand entries and pixels that saturate
early.  Replaces rasterizer/gsplat/forward.cu:294-365 (the oracle checks of the default kernel are
tests/test_gpu_baseline_parity.py and tests/test_gpu_parity.py)."""
import numpy as np
import pytest

from opensplat_amd import scenes
from tests.util import hip_pipeline, np_

pytestmark = pytest.mark.gpu

COMPACT = 3 << 23

CASES = {
    "c2_like_300k": lambda: scenes.camera_scene(300_000, 1920, 1080, K=0, seed=1),
    "sparse_short_lists": lambda: scenes.camera_scene(20_000, 1920, 1080, K=0, seed=2),        # most lists < 64 entries
    "big_splats_long_lists": lambda: scenes.camera_scene(150_000, 1280, 720, K=0, seed=3, sigma_px=(3.0, 12.0)),
    "opaque_saturating": lambda: scenes.camera_scene(400_000, 1600, 900, K=0, seed=4, sigma_px=(2.0, 6.0)),
    "ragged_frame": lambda: scenes.camera_scene(120_000, 1501, 1003, K=0, seed=5, sigma_px=(0.5, 8.0)),
    "hot_spot": lambda: scenes.camera_scene(200_000, 1920, 1080, K=0, seed=6, hot=(0.05, 48)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_compacting_forward_equals_the_chunked_forward_bit_for_bit(name):
    import torch

    from opensplat_amd import cabi

    s = CASES[name]()
    out = hip_pipeline(s, backward=False)
    b = out["binned"]
    assert b.tile_bins.numel() // 2 > 2560          # a full frame: one entry per step
    got = cabi.rasterize_forward(s.W, s.H, b, s.background, COMPACT)
    torch.cuda.synchronize()
    for k in ("img", "final_Ts", "final_idx"):
        assert np.array_equal(np_(out[k]), np_(got[k])), k
    idx = np_(got["final_idx"])
    assert idx.max() >= 0 and idx.max() < b.num_isects
    lens = np_(b.tile_bins)[:, 1] - np_(b.tile_bins)[:, 0]
    print(name, "mean list %.0f longest %d, pixels with a contributor %.3f" % (lens.mean(), lens.max(), (idx >= 0).mean()))
