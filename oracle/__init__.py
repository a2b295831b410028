"""TEST INFRASTRUCTURE ONLY — numpy/ctypes front-ends of the two CPU checkers.

* ``restated`` : oracle/libgsplat_oracle.so, the plain-C restatement (oracle/gsplat_oracle.c)
* ``reference``: oracle/_ref/libgsplat_ref.so, OpenSplat's own gsplat-cpu sources compiled in
  place from /root/reference (oracle/Makefile) behind oracle/ref_shim.cpp.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (opensplat_amd/) never does.

Both front-ends expose the same functions on numpy arrays:

    project_forward, project_backward, sh_forward, sh_backward,
    rasterize_forward (-> img, final_Ts, px_counts, contributor ids, state), rasterize_backward
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _fo(shape):
    a = np.zeros(shape, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _io(shape):
    a = np.zeros(shape, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


def build(ref: bool = True) -> None:
    """Compile the C restatement and (when /root/reference is present) the reference build."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    if ref:
        subprocess.check_call(["make", "-s", "-j5", "-C", _HERE, "ref"])


class _Base:
    """Shared numpy marshalling; subclasses bind the symbols."""

    name = "?"

    # --- projection ---------------------------------------------------------------------------
    def project_forward(self, means, scales, quats, viewmat, projmat, fx, fy, cx, cy, H, W,
                        glob_scale=1.0, clip=0.01):
        raise NotImplementedError

    def project_backward(self, means, scales, quats, viewmat, projmat, fx, fy, cx, cy, H, W,
                         v_xys, v_conics, glob_scale=1.0, clip=0.01):
        raise NotImplementedError


class Restated(_Base):
    name = "restated"

    def __init__(self):
        path = os.path.join(_HERE, "libgsplat_oracle.so")
        if not os.path.exists(path):
            build(ref=False)
        self.lib = C.CDLL(path)
        self.lib.orc_rasterize_forward.restype = C.c_void_p
        self.lib.orc_rasterize_forward_window.restype = C.c_void_p
        self.lib.orc_rasterize_total.restype = C.c_int64
        self.lib.orc_rasterize_total.argtypes = [C.c_void_p]
        self.lib.orc_rasterize_contributors.argtypes = [C.c_void_p, _i32p]
        self.lib.orc_rasterize_free.argtypes = [C.c_void_p]
        self.lib.orc_expf.restype = C.c_float
        self.lib.orc_expf.argtypes = [C.c_float]

    def project_forward(self, means, scales, quats, viewmat, projmat, fx, fy, cx, cy, H, W,
                        glob_scale=1.0, clip=0.01):
        N = len(means)
        m, mp = _f(means); s, sp = _f(scales); q, qp = _f(quats)
        vm, vmp = _f(viewmat); pm, pmp = _f(projmat)
        xys, xp = _fo((N, 2)); radii, rp = _io((N,)); con, cp = _fo((N, 3))
        cov2d, c2p = _fo((N, 2, 2)); cd, cdp = _fo((N,)); dv, dvp = _fo((N,)); c3, c3p = _fo((N, 6))
        kr, krp = _fo((N,))
        self.lib.orc_project_forward(C.c_int(N), mp, sp, C.c_float(glob_scale), qp, vmp, pmp,
                                     C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
                                     C.c_int(H), C.c_int(W), C.c_float(clip), xp, rp, cp, c2p, cdp,
                                     dvp, c3p, krp)
        return dict(xys=xys, radii=radii, conics=con, cov2d=cov2d, cam_depths=cd, depths=dv,
                    cov3d=c3, depth_keys_as_read=kr)

    def project_backward(self, means, scales, quats, viewmat, projmat, fx, fy, cx, cy, H, W,
                         v_xys, v_conics, glob_scale=1.0, clip=0.01, v_depth=None):
        N = len(means)
        m, mp = _f(means); s, sp = _f(scales); q, qp = _f(quats)
        vm, vmp = _f(viewmat); pm, pmp = _f(projmat)
        vx, vxp = _f(v_xys); vc, vcp = _f(v_conics)
        if v_depth is not None:
            vd, vdp = _f(v_depth)
        else:
            vdp = None
        vmn, vmnp = _fo((N, 3)); vs, vsp = _fo((N, 3)); vq, vqp = _fo((N, 4))
        self.lib.orc_project_backward(C.c_int(N), mp, sp, C.c_float(glob_scale), qp, vmp, pmp,
                                      C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
                                      C.c_int(H), C.c_int(W), C.c_float(clip), vxp, vcp, vdp,
                                      vmnp, vsp, vqp)
        return dict(v_means=vmn, v_scales=vs, v_quats=vq)

    def project_gpu_semantics(self, o, means, viewmat, projmat, cx, cy, H, W, clip=0.01):
        """DESIGN.md P2 + P3 applied to a project_forward() result `o` (copy returned): near-plane
        cull (forward.cu:49-52) and the principal-point offset (helpers.cuh:13-15).  Adds
        'visible' (bool [N]) and 'xys_gpu_formula' (helpers.cuh's own expression)."""
        N = len(means)
        o = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in o.items()}
        m, mp = _f(means); vm, vmp = _f(viewmat); pm, pmp = _f(projmat)
        vis, vp = _io((N,)); xg, xgp = _fo((N, 2))
        self.lib.orc_project_gpu_semantics(C.c_int(N), mp, vmp, pmp, C.c_float(cx), C.c_float(cy),
                                           C.c_int(H), C.c_int(W), C.c_float(clip),
                                           o["xys"].ctypes.data_as(_f32p),
                                           o["radii"].ctypes.data_as(_i32p), vp, xgp)
        o["visible"] = vis.astype(bool)
        o["xys_gpu_formula"] = xg
        return o

    def sh_forward(self, degrees_to_use, dirs, coeffs):
        N, K = coeffs.shape[0], coeffs.shape[1]
        d, dp = _f(dirs); c, cp = _f(coeffs); out, op = _fo((N, 3))
        self.lib.orc_sh_forward(C.c_int(N), C.c_int(K), C.c_int(degrees_to_use), dp, cp, op)
        return out

    def sh_backward(self, degrees_to_use, dirs, coeffs, v_colors):
        N, K = coeffs.shape[0], coeffs.shape[1]
        d, dp = _f(dirs); v, vp = _f(v_colors); out, op = _fo((N, K, 3))
        self.lib.orc_sh_backward(C.c_int(N), C.c_int(K), C.c_int(degrees_to_use), dp, vp, op)
        return out

    def rasterize_forward(self, W, H, xys, conics, colors, opacities, background, cov2d,
                          cam_depths, want_contributors=True, window=None, tile_rect=None):
        """window = (x0, y0, x1, y1): evaluate only those pixels (full-image coordinates).
        tile_rect [N, 4] int32 {tx0, tx1, ty0, ty1}: a caller-supplied binning contract — Gaussian g is
        composited only inside those 16 x 16 tiles (orc_rasterize_forward_tiles)."""
        N = len(xys)
        x, xp = _f(xys); cn, cnp = _f(conics); co, cop = _f(colors); o, op = _f(opacities)
        bg, bgp = _f(background); c2, c2p = _f(cov2d); cd, cdp = _f(cam_depths)
        img, ip = _fo((H, W, 3)); fT, fp = _fo((H, W)); cnt, cntp = _io((H, W))
        wx0, wy0, wx1, wy1 = window if window is not None else (0, 0, W, H)
        if tile_rect is not None:
            assert window is None
            tr = np.ascontiguousarray(tile_rect, dtype=np.int32)
            assert tr.shape == (N, 4)
            self.lib.orc_rasterize_forward_tiles.restype = C.c_void_p
            st = self.lib.orc_rasterize_forward_tiles(
                C.c_int(W), C.c_int(H), C.c_int(N), xp, cnp, cop, op, bgp, c2p, cdp,
                tr.ctypes.data_as(_i32p), ip, fp, cntp)
        else:
            st = self.lib.orc_rasterize_forward_window(
                C.c_int(W), C.c_int(H), C.c_int(N), xp, cnp, cop, op, bgp, c2p, cdp, ip, fp, cntp,
                C.c_int(wx0), C.c_int(wy0), C.c_int(wx1), C.c_int(wy1))
        ids = None
        if want_contributors:
            total = self.lib.orc_rasterize_total(C.c_void_p(st))
            ids, idp = _io((max(int(total), 1),))
            self.lib.orc_rasterize_contributors(C.c_void_p(st), idp)
            ids = ids[: int(total)]
        return dict(img=img, final_Ts=fT, px_counts=cnt, contributors=ids, state=st)

    def rasterize_backward(self, W, H, xys, conics, colors, opacities, background, cov2d,
                           cam_depths, final_Ts, state, v_out, free=True, window=None):
        N = len(xys)
        x, xp = _f(xys); cn, cnp = _f(conics); co, cop = _f(colors); o, op = _f(opacities)
        bg, bgp = _f(background); fT, fp = _f(final_Ts); vo, vop = _f(v_out)
        vxy, vxyp = _fo((N, 2)); vcn, vcnp = _fo((N, 3)); vco, vcop = _fo((N, 3)); vop_, vopp = _fo((N,))
        wx0, wy0, wx1, wy1 = window if window is not None else (0, 0, W, H)
        self.lib.orc_rasterize_backward_window(
            C.c_int(W), C.c_int(H), C.c_int(N), xp, cnp, cop, op, bgp, fp, C.c_void_p(state), vop,
            None, vxyp, vcnp, vcop, vopp, C.c_int(wx0), C.c_int(wy0), C.c_int(wx1), C.c_int(wy1))
        if free:
            self.lib.orc_rasterize_free(C.c_void_p(state))
        return dict(v_xy=vxy, v_conic=vcn, v_colors=vco, v_opacity=vop_)

    def rasterize_free(self, state):
        self.lib.orc_rasterize_free(C.c_void_p(state))

    def expf(self, x):
        x, xp = _f(x)
        y, yp = _fo(x.shape)
        self.lib.orc_expf_array(C.c_int64(x.size), xp, yp)
        return y


    # ---- SURVEY.md §8 row f3 (oracle/image_oracle.c: OpenCV 4.5.4's published algorithms) ---------
    def resize_area(self, img_u8, dst_w=0, dst_h=0, inv_scale=0.0):
        sh, sw = img_u8.shape[:2]
        src = np.ascontiguousarray(img_u8, np.uint8)
        ow, oh = C.c_int(), C.c_int()
        args = (C.c_int(sw), C.c_int(sh), None, C.c_int(dst_w), C.c_int(dst_h), C.c_double(inv_scale),
                C.c_double(inv_scale), C.byref(ow), C.byref(oh))
        rc = self.lib.orc_resize_area_u8c3(src.ctypes.data_as(C.c_void_p), *args)
        assert rc == 0, "not a down-scale"
        out = np.zeros((oh.value, ow.value, 3), np.uint8)
        rc = self.lib.orc_resize_area_u8c3(src.ctypes.data_as(C.c_void_p), C.c_int(sw), C.c_int(sh),
                                           out.ctypes.data_as(C.c_void_p), C.c_int(dst_w), C.c_int(dst_h),
                                           C.c_double(inv_scale), C.c_double(inv_scale), C.byref(ow), C.byref(oh))
        assert rc == 0
        return out

    @staticmethod
    def _dist8(dist):
        d = np.zeros(8, np.float32)
        d[:len(dist)] = np.asarray(dist, np.float32)
        return d

    def optimal_new_camera_matrix(self, K, dist, W, H, alpha=0.0):
        K9, kp = _f(np.asarray(K, np.float32).reshape(9)); d8, dp = _f(self._dist8(dist))
        nk, nkp = _fo((9,)); roi, rp = _io((4,))
        self.lib.orc_optimal_new_camera_matrix(kp, dp, C.c_int(W), C.c_int(H), C.c_double(alpha), nkp, rp)
        return nk.reshape(3, 3), tuple(int(v) for v in roi)

    def undistort(self, img_u8, K, dist, newK):
        H, W = img_u8.shape[:2]
        src = np.ascontiguousarray(img_u8, np.uint8)
        K9, kp = _f(np.asarray(K, np.float32).reshape(9)); d8, dp = _f(self._dist8(dist))
        N9, np_ = _f(np.asarray(newK, np.float32).reshape(9))
        out = np.zeros_like(src)
        self.lib.orc_undistort_u8c3(src.ctypes.data_as(C.c_void_p), C.c_int(W), C.c_int(H), kp, dp, np_,
                                    out.ctypes.data_as(C.c_void_p))
        return out

    # ---- SURVEY.md §8 row f2 (oracle/train_oracle.c) ------------------------------------------
    def ssim_window(self):
        g, gp = _fo((11,)); w2, wp = _fo((11, 11))
        self.lib.orc_ssim_gaussian(C.c_float(1.5), gp)
        self.lib.orc_ssim_window(wp)
        return g, w2

    def main_loss(self, rendered, gt, ssim_weight, want_grad=True):
        H, W = rendered.shape[:2]
        r, rp = _f(rendered); g, gp = _f(gt)
        loss, lp = _fo((3,)); v, vp = _fo((H, W, 3))
        rc = self.lib.orc_main_loss(C.c_int(W), C.c_int(H), rp, gp, C.c_float(ssim_weight), lp,
                                    vp if want_grad else None)
        assert rc == 0
        return loss, (v if want_grad else None)

    def adam_step(self, p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
        """In place on float32 numpy arrays p, m, v."""
        self.lib.orc_adam_step.argtypes = [C.c_int64, _f32p, _f32p, _f32p, _f32p, C.c_double,
                                           C.c_double, C.c_double, C.c_double, C.c_int64]
        for a in (p, m, v):
            assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
        gg, ggp = _f(g)
        self.lib.orc_adam_step(p.size, p.ctypes.data_as(_f32p), ggp, m.ctypes.data_as(_f32p),
                               v.ctypes.data_as(_f32p), lr, beta1, beta2, eps, step)

    def sched_lr(self, lr_init, lr_final, max_steps, step):
        self.lib.orc_sched_lr.restype = C.c_float
        self.lib.orc_sched_lr.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int]
        return float(self.lib.orc_sched_lr(lr_init, lr_final, max_steps, step))

    # ---- SURVEY.md §8 row f4 (oracle/densify_oracle.c) ----------------------------------------
    def densify_stats(self, xys_grad, radii, height, width, first, gnorm, vis, m2d):
        """In place on float32 arrays gnorm, vis, m2d."""
        N = len(radii)
        g, gp = _f(xys_grad)
        r = np.ascontiguousarray(radii, np.int32)
        self.lib.orc_densify_stats(C.c_int(N), gp, r.ctypes.data_as(_i32p), C.c_int(height),
                                   C.c_int(width), C.c_int(int(first)), gnorm.ctypes.data_as(_f32p),
                                   vis.ctypes.data_as(_f32p), m2d.ctypes.data_as(_f32p))

    def densify_refine(self, prob, grad_thresh, size_thresh, check_screen, split_screen, cull_huge,
                       samples_fn):
        return _densify_refine(self.lib.orc_densify_refine, prob, grad_thresh, size_thresh,
                               check_screen, split_screen, cull_huge, samples_fn)

    def reset_opacity(self, logits, reset_value=0.2):
        out = np.ascontiguousarray(logits, np.float32).copy()
        self.lib.orc_reset_opacity(C.c_int(out.size), C.c_float(reset_value), out.ctypes.data_as(_f32p))
        return out

class Reference(_Base):
    name = "reference"

    def __init__(self):
        import torch  # noqa: F401  (libgsplat_ref.so links libtorch_cpu; load it first)

        path = os.path.join(_HERE, "_ref", "libgsplat_ref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self.lib.ref_last_error.restype = C.c_char_p

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("reference threw: " + self.lib.ref_last_error().decode())

    def num_threads(self):
        return int(self.lib.ref_num_threads())

    def project_forward(self, means, scales, quats, viewmat, projmat, fx, fy, cx, cy, H, W,
                        glob_scale=1.0, clip=0.01):
        N = len(means)
        m, mp = _f(means); s, sp = _f(scales); q, qp = _f(quats)
        vm, vmp = _f(viewmat); pm, pmp = _f(projmat)
        xys, xp = _fo((N, 2)); radii, rp = _io((N,)); con, cp = _fo((N, 3))
        cov2d, c2p = _fo((N, 2, 2)); cd, cdp = _fo((N,)); kr, krp = _fo((N,))
        self._chk(self.lib.ref_project_forward(
            C.c_int(N), mp, sp, C.c_float(glob_scale), qp, vmp, pmp, C.c_float(fx), C.c_float(fy),
            C.c_float(cx), C.c_float(cy), C.c_int(H), C.c_int(W), C.c_float(clip), xp, rp, cp, c2p,
            cdp, krp))
        return dict(xys=xys, radii=radii, conics=con, cov2d=cov2d, cam_depths=cd,
                    depth_keys_as_read=kr)

    def project_backward(self, means, scales, quats, viewmat, projmat, fx, fy, cx, cy, H, W,
                         v_xys, v_conics, glob_scale=1.0, clip=0.01):
        N = len(means)
        m, mp = _f(means); s, sp = _f(scales); q, qp = _f(quats)
        vm, vmp = _f(viewmat); pm, pmp = _f(projmat)
        vx, vxp = _f(v_xys); vc, vcp = _f(v_conics)
        vmn, vmnp = _fo((N, 3)); vs, vsp = _fo((N, 3)); vq, vqp = _fo((N, 4))
        self._chk(self.lib.ref_project_backward(
            C.c_int(N), mp, sp, C.c_float(glob_scale), qp, vmp, pmp, C.c_float(fx), C.c_float(fy),
            C.c_float(cx), C.c_float(cy), C.c_int(H), C.c_int(W), C.c_float(clip), vxp, vcp, vmnp,
            vsp, vqp))
        return dict(v_means=vmn, v_scales=vs, v_quats=vq)

    def sh_forward(self, degrees_to_use, dirs, coeffs):
        N, K = coeffs.shape[0], coeffs.shape[1]
        d, dp = _f(dirs); c, cp = _f(coeffs); out, op = _fo((N, 3))
        self._chk(self.lib.ref_sh_forward(C.c_int(N), C.c_int(K), C.c_int(degrees_to_use), dp, cp, op))
        return out

    def sh_backward(self, degrees_to_use, dirs, coeffs, v_colors):
        N, K = coeffs.shape[0], coeffs.shape[1]
        d, dp = _f(dirs); c, cp = _f(coeffs); v, vp = _f(v_colors); out, op = _fo((N, K, 3))
        self._chk(self.lib.ref_sh_backward(C.c_int(N), C.c_int(K), C.c_int(degrees_to_use), dp, cp,
                                           vp, op))
        return out

    def rasterize_forward(self, W, H, xys, conics, colors, opacities, background, cov2d,
                          cam_depths, want_contributors=True):
        N = len(xys)
        x, xp = _f(xys); cn, cnp = _f(conics); co, cop = _f(colors); o, op = _f(opacities)
        bg, bgp = _f(background); c2, c2p = _f(cov2d); cd, cdp = _f(cam_depths)
        img, ip = _fo((H, W, 3)); fT, fp = _fo((H, W)); cnt, cntp = _io((H, W))
        st = C.c_void_p()
        self._chk(self.lib.ref_rasterize_forward(C.c_int(W), C.c_int(H), C.c_int(N), xp, cnp, cop,
                                                 op, bgp, c2p, cdp, ip, fp, cntp, C.byref(st)))
        ids = None
        if want_contributors:
            total = int(cnt.sum())
            ids, idp = _io((max(total, 1),))
            self.lib.ref_rasterize_contributors(st, idp)
            ids = ids[:total]
        return dict(img=img, final_Ts=fT, px_counts=cnt, contributors=ids, state=st)

    def rasterize_backward(self, W, H, xys, conics, colors, opacities, background, cov2d,
                           cam_depths, final_Ts, state, v_out, free=True):
        N = len(xys)
        x, xp = _f(xys); cn, cnp = _f(conics); co, cop = _f(colors); o, op = _f(opacities)
        bg, bgp = _f(background); c2, c2p = _f(cov2d); cd, cdp = _f(cam_depths)
        fT, fp = _f(final_Ts); vo, vop = _f(v_out)
        vxy, vxyp = _fo((N, 2)); vcn, vcnp = _fo((N, 3)); vco, vcop = _fo((N, 3)); vop_, vopp = _fo((N,))
        # NB the reference deletes px2gid itself only inside RasterizeGaussiansCPU::backward;
        # through the raw *_tensor_cpu functions the shim owns it.
        self._chk(self.lib.ref_rasterize_backward(C.c_int(W), C.c_int(H), C.c_int(N), xp, cnp, cop,
                                                  op, bgp, c2p, cdp, fp, state, vop, vxyp, vcnp,
                                                  vcop, vopp))
        if free:
            self.lib.ref_rasterize_free(state)
        return dict(v_xy=vxy, v_conic=vcn, v_colors=vco, v_opacity=vop_)

    def rasterize_free(self, state):
        self.lib.ref_rasterize_free(state)

    def chain_fwd_bwd(self, means, scales, quats, dirs, coeffs, opacities, viewmat, projmat, fx,
                      fy, cx, cy, H, W, background, v_out, degrees_to_use=0):
        """Whole hot path through the reference's op wrappers; coeffs [N,K,3] or colours [N,3]."""
        N = len(means)
        K = coeffs.shape[1] if coeffs.ndim == 3 else 0
        m, mp = _f(means); s, sp = _f(scales); q, qp = _f(quats)
        if dirs is not None:
            d, dp = _f(dirs)
        else:
            dp = None
        c, cp = _f(coeffs); o, op = _f(opacities)
        vm, vmp = _f(viewmat); pm, pmp = _f(projmat); bg, bgp = _f(background)
        img, ip = _fo((H, W, 3))
        vmn, vmnp = _fo((N, 3)); vs, vsp = _fo((N, 3)); vq, vqp = _fo((N, 4))
        vc, vcp = _fo(coeffs.shape); vo_, vop_ = _fo((N,))
        if v_out is not None:
            vout, voutp = _f(v_out)
        else:
            voutp = None
        times = (C.c_double * 2)()
        self._chk(self.lib.ref_chain_fwd_bwd(
            C.c_int(N), C.c_int(K), C.c_int(degrees_to_use), mp, sp, qp, dp, cp, op, vmp, pmp,
            C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), C.c_int(H), C.c_int(W),
            bgp, voutp, ip, vmnp, vsp, vqp, vcp, vop_, times))
        return dict(img=img, v_means=vmn, v_scales=vs, v_quats=vq, v_coeffs=vc, v_opacities=vo_,
                    fwd_ms=times[0], bwd_ms=times[1])


    # ---- SURVEY.md §8 row f2 (oracle/ref_train_shim.cpp) --------------------------------------
    def _tchk(self, rc):
        if rc != 0:
            self.lib.ref_train_last_error.restype = C.c_char_p
            raise RuntimeError("reference threw: " + self.lib.ref_train_last_error().decode())

    def ssim_window(self):
        g, gp = _fo((11,)); w2, wp = _fo((11, 11))
        self._tchk(self.lib.ref_ssim_window(wp, gp))
        return g, w2

    def main_loss(self, rendered, gt, ssim_weight, want_grad=True):
        H, W = rendered.shape[:2]
        r, rp = _f(rendered); g, gp = _f(gt)
        loss, lp = _fo((3,)); v, vp = _fo((H, W, 3))
        ms = C.c_double()
        self._tchk(self.lib.ref_main_loss(C.c_int(W), C.c_int(H), rp, gp, C.c_float(ssim_weight), lp,
                                          vp if want_grad else None, C.byref(ms)))
        self.last_ms = ms.value
        return loss, (v if want_grad else None)

    def adam_steps(self, p, grads, lr):
        """`len(grads)` libtorch Adam steps; returns (param, exp_avg, exp_avg_sq)."""
        p = np.ascontiguousarray(p, np.float32).copy()
        gr, grp = _f(np.stack(grads))
        n = p.size
        m, mp = _fo((n,)); v, vp = _fo((n,))
        self._tchk(self.lib.ref_adam_steps(C.c_int64(n), p.ctypes.data_as(_f32p), grp,
                                           C.c_int(len(grads)), C.c_double(lr), mp, vp))
        return p, m, v

    def sched_lr(self, lr_init, lr_final, max_steps, step):
        self.lib.ref_sched_lr.restype = C.c_float
        self.lib.ref_sched_lr.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int]
        return float(self.lib.ref_sched_lr(lr_init, lr_final, max_steps, step))


    # ---- SURVEY.md §8 row f4 (oracle/ref_train_shim.cpp: afterTrain restated under libtorch) ----
    def densify_stats(self, xys_grad, radii, height, width, first, gnorm, vis, m2d):
        N = len(radii)
        g, gp = _f(xys_grad)
        r = np.ascontiguousarray(radii, np.int32)
        self._tchk(self.lib.ref_densify_stats(C.c_int(N), gp, r.ctypes.data_as(_i32p), C.c_int(height),
                                              C.c_int(width), C.c_int(int(first)),
                                              gnorm.ctypes.data_as(_f32p), vis.ctypes.data_as(_f32p),
                                              m2d.ctypes.data_as(_f32p)))

    def densify_refine(self, prob, grad_thresh, size_thresh, check_screen, split_screen, cull_huge,
                       samples_fn):
        def call(*a):
            self._tchk(self.lib.ref_densify_refine(*a))
            return 0
        return _densify_refine(call, prob, grad_thresh, size_thresh, check_screen, split_screen,
                               cull_huge, samples_fn)


    # ---- the reference's own Model, compiled in place (oracle/ref_model_shim.cpp) ------------------
    def model(self, xyz, rgb_u8, **cfg):
        return RefModel(self.lib, xyz, rgb_u8, **cfg)


class RefModel:
    """Handle on OpenSplat's `Model` (model.hpp, CPU device) inside oracle/_ref/libgsplat_ref.so:
    forward / mainLoss / optimizersStep / schedulersStep / afterTrain / savePly / saveSplat are the
    reference's compiled code.  TEST INFRASTRUCTURE."""

    DEFAULTS = dict(numCameras=10, numDownscales=0, resolutionSchedule=3000, shDegree=3, shDegreeInterval=1000,
                    refineEvery=100, warmupLength=500, resetAlphaEvery=30, densifyGradThresh=0.0002,
                    densifySizeThresh=0.01, stopScreenSizeAt=4000, splitScreenSize=0.05, maxSteps=30000,
                    keepCrs=False, scale=1.0, translation=None)

    def __init__(self, lib, xyz, rgb_u8, **cfg):
        c = dict(self.DEFAULTS)
        c.update(cfg)
        self.cfg, self.lib = c, lib
        lib.refm_create.restype = C.c_void_p
        lib.refm_last_error.restype = C.c_char_p
        lib.refm_means_lr.restype = C.c_float
        for f in (lib.refm_num_points, lib.refm_sh_bases, lib.refm_destroy, lib.refm_means_lr):
            f.argtypes = [C.c_void_p]
        lib.refm_downscale_factor.argtypes = [C.c_void_p, C.c_int]
        x, xp = _f(xyz)
        r = np.ascontiguousarray(rgb_u8, np.uint8)
        tr = None if c["translation"] is None else _f(c["translation"])
        h = lib.refm_create(C.c_int(len(x)), xp, r.ctypes.data_as(C.c_void_p), C.c_int(c["numCameras"]),
                            C.c_int(c["numDownscales"]), C.c_int(c["resolutionSchedule"]), C.c_int(c["shDegree"]),
                            C.c_int(c["shDegreeInterval"]), C.c_int(c["refineEvery"]), C.c_int(c["warmupLength"]),
                            C.c_int(c["resetAlphaEvery"]), C.c_float(c["densifyGradThresh"]),
                            C.c_float(c["densifySizeThresh"]), C.c_int(c["stopScreenSizeAt"]),
                            C.c_float(c["splitScreenSize"]), C.c_int(c["maxSteps"]), C.c_int(int(c["keepCrs"])),
                            C.c_float(c["scale"]), tr[1] if tr else None)
        if not h:
            raise RuntimeError("reference threw: " + lib.refm_last_error().decode())
        self.h = C.c_void_p(h)

    def __del__(self):
        try:
            self.lib.refm_destroy(self.h)
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("reference threw: " + self.lib.refm_last_error().decode())

    @property
    def N(self):
        return int(self.lib.refm_num_points(self.h))

    @property
    def K(self):
        return int(self.lib.refm_sh_bases(self.h))

    @staticmethod
    def _ptrs(arrs):
        keep = [np.ascontiguousarray(a, np.float32) for a in arrs]
        return keep, (_f32p * 6)(*[a.ctypes.data_as(_f32p) for a in keep])

    def set_state(self, params, exp_avg=None, exp_avg_sq=None, adam_step=1):
        N, K = params[0].shape[0], params[5].shape[1] + 1
        kp, pp = self._ptrs(params)
        if exp_avg is not None:
            ka, pa = self._ptrs(exp_avg)
            ks, ps = self._ptrs(exp_avg_sq)
        else:
            pa = ps = None
        self._chk(self.lib.refm_set_state(self.h, C.c_int(N), C.c_int(K), pp, pa, ps, C.c_int64(adam_step)))

    def get_state(self, moments=True):
        N, K = self.N, self.K
        shapes = [(N, 3), (N, 3), (N, 4), (N, 1), (N, 3), (N, K - 1, 3)]
        outs = [[np.zeros(sh, np.float32) for sh in shapes] for _ in range(3)]
        ptr = [(_f32p * 6)(*[a.ctypes.data_as(_f32p) for a in o]) for o in outs]
        self._chk(self.lib.refm_get_state(self.h, ptr[0], ptr[1] if moments else None, ptr[2] if moments else None))
        return dict(params=outs[0], exp_avg=outs[1], exp_avg_sq=outs[2])

    def after_train(self, step, radii, xys_grad, stats, height, width, seed=0):
        """Model::afterTrain(step).  stats = (xysGradNorm, visCounts, max2DSize) or None.
        -> (refined?, stats after the call or None)"""
        N = len(radii)
        r = np.ascontiguousarray(radii, np.int32)
        g = _f(xys_grad) if xys_grad is not None else (None, None)
        st = [(_f(a)) for a in stats] if stats is not None else [(None, None)] * 3
        out = np.zeros((3, N), np.float32)
        refined = C.c_int(0)
        self._chk(self.lib.refm_after_train(self.h, C.c_int(step), C.c_int(N), r.ctypes.data_as(_i32p), g[1],
                                            st[0][1], st[1][1], st[2][1], C.c_int(height), C.c_int(width),
                                            C.c_uint64(seed), out.ctypes.data_as(_f32p), C.byref(refined)))
        return bool(refined.value), (None if refined.value else (out[0], out[1], out[2]))

    def save(self, path, step=0):
        self._chk(self.lib.refm_save(self.h, str(path).encode(), C.c_int(step)))

    def downscale_factor(self, step):
        return int(self.lib.refm_downscale_factor(self.h, step))

    def means_lr(self):
        return float(self.lib.refm_means_lr(self.h))

    def train_iteration(self, step, cam, gt, ssim_weight=0.2, seed=0, after_train=True):
        """One iteration of opensplat.cpp:151-170 on the CPU.  cam: dict(width, height, fx, fy, cx, cy,
        camToWorld [4,4]).  -> dict(loss, rgb, xys_grad, radii)."""
        N = self.N
        d = self.downscale_factor(step)
        h, w = cam["height"] // d, cam["width"] // d     # model.cpp:88-89 (integer division of floats truncated)
        g, gp = _f(gt)
        assert g.shape == (h, w, 3), (g.shape, (h, w, 3))
        c2w, cp = _f(cam["camToWorld"])
        loss = C.c_float()
        rgb, rp = _fo((h, w, 3)); xg, xgp = _fo((N, 2)); rad, radp = _io((N,))
        self._chk(self.lib.refm_train_iteration(self.h, C.c_int(step), C.c_int(cam["width"]), C.c_int(cam["height"]),
                                                C.c_float(cam["fx"]), C.c_float(cam["fy"]), C.c_float(cam["cx"]),
                                                C.c_float(cam["cy"]), cp, gp, C.c_float(ssim_weight),
                                                C.c_uint64(seed), C.byref(loss), rp, xgp, radp,
                                                C.c_int(int(after_train))))
        return dict(loss=float(loss.value), rgb=rgb, xys_grad=xg, radii=rad)


def _densify_refine(fn, prob, grad_thresh, size_thresh, check_screen, split_screen, cull_huge, samples_fn):
    """Shared driver of orc_densify_refine / ref_densify_refine (same C signature).  samples_fn(n)
    -> float32 [2 n, 3] normal samples.  Returns dict(params, exp_avg, exp_avg_sq, counts)."""
    N, K = prob["N"], prob["K"]
    PP = C.POINTER(_f32p)

    def ptrs(arrs):
        keep = [np.ascontiguousarray(a, np.float32) for a in arrs]
        arr = (_f32p * 6)(*[a.ctypes.data_as(_f32p) for a in keep])
        return keep, arr
    kp, pp = ptrs(prob["params"]); ka, pa = ptrs(prob["exp_avg"]); ks, ps = ptrs(prob["exp_avg_sq"])
    g, gp = _f(prob["xys_grad_norm"]); v, vp = _f(prob["vis_counts"]); m, mp = _f(prob["max_2d_size"])
    lens = [3, 3, 4, 1, 3, (K - 1) * 3]
    outs = [[np.zeros((4 * N, l), np.float32) for l in lens] for _ in range(3)]
    optr = [(_f32p * 6)(*[a.ctypes.data_as(_f32p) for a in o]) for o in outs]
    counts = (C.c_int32 * 4)()
    args = lambda smp: (C.c_int(N), C.c_int(K), pp, pa, ps, gp, vp, mp, C.c_int(prob["width"]),
                        C.c_int(prob["height"]), C.c_float(grad_thresh), C.c_float(size_thresh),
                        C.c_int(int(check_screen)), C.c_float(split_screen), C.c_int(int(cull_huge)),
                        smp, optr[0], optr[1], optr[2], counts)
    rc = fn(*args(None))
    assert rc == 0
    n_splits = counts[0]
    samples = np.ascontiguousarray(samples_fn(n_splits), np.float32)
    assert samples.shape == (2 * n_splits, 3)
    if n_splits > 0:
        rc = fn(*args(samples.ctypes.data_as(_f32p)))
        assert rc == 0
    new_n = counts[1]
    shapes = [(new_n, 3), (new_n, 3), (new_n, 4), (new_n, 1), (new_n, 3), (new_n, K - 1, 3)]
    cut = lambda o: [a[:new_n].reshape(sh).copy() for a, sh in zip(o, shapes)]
    return dict(params=cut(outs[0]), exp_avg=cut(outs[1]), exp_avg_sq=cut(outs[2]),
                n_splits=n_splits, new_n=new_n, n_dups=counts[2], culled=counts[3], samples=samples)


_restated = None
_reference = None


def restated() -> Restated:
    global _restated
    if _restated is None:
        _restated = Restated()
    return _restated


def reference() -> Reference:
    global _reference
    if _reference is None:
        _reference = Reference()
    return _reference


def have_reference() -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", "libgsplat_ref.so"))
