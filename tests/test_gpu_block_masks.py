"""-m gpu: the per-entry coverage masks of the tile lists (gs_block_masks / block_mask16).

A mask bit must be set for every 4x4-pixel block of a tile that holds a pixel the compositing kernels
composite for that entry: inside the record's rectangle with 0 <= sigma <= sigma_max, sigma evaluated
in the kernels' own fp32 operation order (gsplat_cpu.cpp:213-217).  The masks may be a superset (they
only prune work) — how tight they are is asserted on benchmark-like scenes.  The bit-exact image tests
of the whole suite run on top of these masks; this file checks the property directly, including
needle-shaped, huge, tiny and degenerate conics."""
import numpy as np
import pytest

from opensplat_amd import scenes
from tests.util import hip_pipeline, np_

pytestmark = pytest.mark.gpu


def _live_block_masks(packed, ids, bins, tiles_x, W, H):
    """Brute force: per list entry the set of 4x4 blocks of its tile with a live pixel (16-bit mask)."""
    f = np.float32
    out = np.zeros(len(ids), np.uint32)
    ly, lx = np.meshgrid(np.arange(16), np.arange(16), indexing="ij")
    blk = ((ly // 4) * 4 + (lx // 4)).astype(np.uint32)            # [16,16] bit index of each pixel
    for t, (a, b) in enumerate(bins):
        if b <= a:
            continue
        g = ids[a:b]
        r = packed[g]
        x, y, A, B, C = (r[:, k].astype(f)[:, None, None] for k in (0, 1, 2, 3, 4))
        smax_bits = r[:, 6].view(np.uint32)[:, None, None]
        rx, ry = r[:, 7].view(np.uint32), r[:, 11].view(np.uint32)
        x0, x1 = (rx & 0xFFFF)[:, None, None], (rx >> 16)[:, None, None]
        y0, y1 = (ry & 0xFFFF)[:, None, None], (ry >> 16)[:, None, None]
        px = ((t % tiles_x) * 16 + lx)[None].astype(np.int64)
        py = ((t // tiles_x) * 16 + ly)[None].astype(np.int64)
        dx, dy = x - px.astype(f), y - py.astype(f)
        sg = ((A * dx) * dx + (C * dy) * dy).astype(f)
        sg = (f(0.5) * sg).astype(f)
        sg = (sg + ((B * dx) * dy).astype(f)).astype(f)
        inrect = (px >= x0) & (px < x1) & (py >= y0) & (py < y1) & (px < W) & (py < H)
        # the kernels' test: unsigned compare of the bit patterns (negatives and NaNs fail)
        live = inrect & (sg.view(np.uint32) <= smax_bits)
        bits = np.zeros(len(g), np.uint32)
        for k in range(16):
            bits |= (live & (blk[None] == k)).any(axis=(1, 2)).astype(np.uint32) << np.uint32(k)
        out[a:b] = bits
    return out


def _check(s, max_excess):
    out = hip_pipeline(s, backward=False)
    b = out["binned"]
    tiles_x = (s.W + 15) // 16
    ids, masks = np_(b.gaussian_ids_sorted), np_(b.block_masks).view(np.uint16).astype(np.uint32)
    bins = np_(b.tile_bins).reshape(-1, 2)
    live = _live_block_masks(np_(b.packed), ids, bins, tiles_x, s.W, s.H)
    missing = live & ~masks
    assert not missing.any(), "coverage mask misses a live block for %d entries" % int((missing != 0).sum())
    pop = lambda a: int(np.unpackbits(a.astype(np.uint16).view(np.uint8)).sum())
    n_mask, n_live = pop(masks), pop(live)
    assert n_mask <= max_excess * max(n_live, 1) + 16, (n_mask, n_live)
    return n_mask, n_live, len(ids)


@pytest.mark.parametrize("sigma_px,seed", [((0.5, 4.0), 1), ((1.0, 8.0), 2), ((0.3, 1.2), 3), ((4.0, 30.0), 4)])
def test_masks_cover_every_live_block_and_are_tight(sigma_px, seed):
    s = scenes.camera_scene(6000, 320, 192, K=1, seed=seed, sigma_px=sigma_px, znear=1.0, zfar=100.0)
    n_mask, n_live, n = _check(s, max_excess=1.03)
    assert n > 1000


def test_masks_with_needles_giants_and_faint_gaussians():
    """Aspect ratios up to 1:300 (the ellipse test is distrusted below det = 1e-3 A C and the rectangle
    is used alone: no tightness to speak of there), Gaussians covering the whole frame, opacities around the 1/255 threshold, ragged
    image edges."""
    s = scenes.camera_scene(4000, 203, 117, K=1, seed=9, sigma_px=(0.5, 6.0), znear=1.0, zfar=100.0)
    rs = np.random.RandomState(1)
    s.scales[0::5, 0] *= rs.uniform(10, 300, len(s.scales[0::5]))       # needles
    s.scales[1::50] *= 40.0                                             # giants
    op = s.opacities.reshape(-1)
    op[2::7] = rs.uniform(0.0035, 0.0045, len(op[2::7]))                # around 1/255
    op[3::11] = 1.0
    s.opacities = op.reshape(s.opacities.shape).astype(np.float32)
    _check(s, max_excess=8.0)


def test_masks_are_what_the_compositing_kernels_walk():
    """Clearing a mask bit of a live block must change the image (the kernels really use the masks),
    setting all bits must not (a superset only costs time)."""
    import torch

    from opensplat_amd import cabi

    s = scenes.camera_scene(3000, 160, 96, K=1, seed=5, sigma_px=(1.0, 5.0), znear=1.0, zfar=100.0)
    out = hip_pipeline(s, backward=False)
    b = out["binned"]
    img0 = np_(out["img"]).copy()
    keep = b.block_masks.clone()
    b.block_masks.fill_(-1)                                            # all 16 bits
    f = cabi.rasterize_forward(s.W, s.H, b, s.background)
    torch.cuda.synchronize()
    assert np.array_equal(np_(f["img"]), img0)
    b.block_masks.zero_()
    f = cabi.rasterize_forward(s.W, s.H, b, s.background)
    torch.cuda.synchronize()
    bg = np.broadcast_to(np.asarray(s.background, np.float32), img0.shape)
    assert np.array_equal(np_(f["img"]), bg)                           # nothing composited
    b.block_masks.copy_(keep)
