#!/usr/bin/env python3
"""Regenerates tests/golden/io_*.{ply,splat}: Model::savePly / saveSplat's
statements (restated in oracle/ref_train_shim.cpp) executed under libtorch on a seeded Gaussian set
(scenes.densify_problem(40, 4, 31)); one PLY with keepCrs (scale 2.5, translation (1,-2,3)).
Build container only.  Usage: python tests/golden/make_golden_io.py"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from opensplat_amd import scenes  # noqa: E402


def main():
    R = oracle.reference()
    prob = scenes.densify_problem(40, 4, 31)
    means, ls, q, op, dc, rest = [np.ascontiguousarray(a, np.float32) for a in prob["params"]]
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    tr = np.array([1.0, -2.0, 3.0], np.float32)
    l = R.lib
    assert l.ref_save_ply(os.path.join(HERE, "io_plain.ply").encode(), 40, 4, fp(means), fp(ls), fp(q),
                          fp(op), fp(dc), fp(rest), 1234, 0, C.c_float(1.0), None) == 0
    assert l.ref_save_ply(os.path.join(HERE, "io_crs.ply").encode(), 40, 4, fp(means), fp(ls), fp(q),
                          fp(op), fp(dc), fp(rest), 7, 1, C.c_float(2.5), fp(tr)) == 0
    assert l.ref_save_splat(os.path.join(HERE, "io_plain.splat").encode(), 40, fp(means), fp(ls), fp(q),
                            fp(op), fp(dc), 0, C.c_float(1.0), None) == 0
    print("wrote io_plain.ply io_crs.ply io_plain.splat")


if __name__ == "__main__":
    main()
