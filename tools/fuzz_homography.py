#!/usr/bin/env python3
"""Randomised parity sweep of nvxuFindHomography against the oracle (oracle/vxoracle_homography.c): matrix, inlier mask and mask
item count, bit for bit.  Covers what the fixed cases of tests/test_gpu_homography.py cannot: every point count from 0 up,
array capacities equal to / above the count (odd and even: the last pair of the on-chip kernel's 16-byte loads), source, destination
and mask arrays of DIFFERENT capacities (the smallest bounds the count), lost points at random, thresholds from 0.3 to 6 px,
hypothesis counts from 1 to 2000, 0 to 10 refits, with and without a mask, and both kernels (capacities up to 4096: the
96-register kernel; beyond: the any-size kernel).
usage: tools/fuzz_homography.py [cases] [seed]   (GPU; prints a summary line, exits non-zero on the first difference)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import oracle as O  # noqa: E402
from test_oracle_homography import H_TRUE, scene  # noqa: E402
from visualslam_amd import ovx  # noqa: E402

L = ovx.lib()


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
    rng = np.random.default_rng(seed)
    ctx = L.vxCreateContext()
    ovx.check(L.vxGetStatus(ctx), "vxCreateContext")
    stats = {"cases": 0, "on_chip_kernel": 0, "any_size_kernel": 0, "identity_results": 0, "with_mask": 0, "max_points": 0}
    for c in range(cases):
        kind = rng.integers(0, 10)
        if kind == 0:
            n = int(rng.integers(0, 12))
        elif kind == 1:
            n = int(rng.choice([255, 256, 257, 511, 513, 4095, 4096]))
        elif kind == 2:
            n = int(rng.integers(4097, 9000))
        else:
            n = int(rng.integers(12, 4097))
        Hs = H_TRUE.copy()
        Hs[:2, :2] += rng.normal(0, 0.01, (2, 2))
        Hs[:2, 2] += rng.normal(0, 8.0, 2)
        src, dst, _ = scene(n=max(n, 1), noise=float(rng.choice([0.0, 0.1, 0.3, 1.0])), outliers=float(rng.choice([0.0, 0.2, 0.45, 0.7])),
                            lost=float(rng.choice([0.0, 0.05, 0.5])), seed=int(rng.integers(1 << 30)), H=Hs)
        src, dst = src[:n], dst[:n]
        if n and rng.random() < 0.1:
            src = src.copy()
            src[rng.random(n) < 0.1] = -1.0                       # unusable source points as well
        cap_s = n + int(rng.choice([0, 0, 1, 2, 3, 64])) or 1
        cap_d = n + int(rng.choice([0, 0, 1, 5])) or 1
        use_mask = rng.random() < 0.8
        cap_m = max(1, n - int(rng.choice([0, 0, 0, 1, 7])) if rng.random() < 0.3 else n + int(rng.choice([0, 1, 2])))
        thr = float(rng.choice([0.3, 0.5, 1.0, 3.0, 6.0]))
        est = int(rng.choice([1, 5, 56, 2000]))
        ref = int(rng.choice([0, 1, 3, 10]))
        conf = float(rng.choice([0.5, 0.9, 0.995, 0.99999]))
        outr = float(rng.choice([0.0, 0.3, 0.45, 0.6]))
        a = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, cap_s)
        b = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, cap_d)
        if n:
            ovx.add_items(a, src)
            ovx.add_items(b, dst)
        m = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3)
        mask = L.vxCreateArray(ctx, ovx.VX_TYPE_UINT8, cap_m) if use_mask else None
        ovx.check(L.nvxuFindHomography(ctx, a, b, m, ovx.NVX_FIND_HOMOGRAPHY_METHOD_RANSAC, thr, est, ref, conf, outr, mask), "nvxuFindHomography")
        n_eff = min(n, cap_m) if use_mask else n                 # the mask's capacity bounds the correspondences looked at
        Hw, mw, cw = O.find_homography(src[:n_eff], dst[:n_eff], thr, est, ref, conf, outr)
        got = np.zeros((3, 3), np.float32)
        ovx.check(L.vxCopyMatrix(m, got.ctypes.data, ovx.VX_READ_ONLY, ovx.VX_MEMORY_TYPE_HOST), "vxCopyMatrix")
        what = dict(case=c, n=n, caps=(cap_s, cap_d, cap_m if use_mask else None), thr=thr, est=est, ref=ref, conf=conf, outr=outr)
        if got.tobytes() != Hw.tobytes():
            print("MATRIX DIFFERS", what, got, Hw, sep="\n")
            return 1
        if use_mask:
            gm = ovx.read_array(mask, np.uint8)
            if len(gm) != n_eff or gm.tobytes() != mw.tobytes():
                print("MASK DIFFERS", what, len(gm), int((gm != 0).sum()), cw, sep="\n")
                return 1
            stats["with_mask"] += 1
        bound = min(cap_s, cap_d, cap_m) if use_mask else min(cap_s, cap_d)
        stats["on_chip_kernel" if bound <= 4096 else "any_size_kernel"] += 1
        stats["identity_results"] += int(np.array_equal(got, np.eye(3, dtype=np.float32)))
        stats["max_points"] = max(stats["max_points"], n)
        stats["cases"] += 1
        for h, t in ((a, "Array"), (b, "Array"), (m, "Matrix")) + (((mask, "Array"),) if use_mask else ()):
            ovx.release(t, h)
    hh = C.c_void_p(ctx)
    L.vxReleaseContext(C.byref(hh))
    print("fuzz_homography: seed %d, %d cases identical to the oracle (matrix, mask, count): %s" % (seed, stats["cases"], stats))
    return 0


if __name__ == "__main__":
    sys.exit(main())
