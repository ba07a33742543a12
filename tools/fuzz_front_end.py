#!/usr/bin/env python3
"""Randomised parity sweep of the thin layer (NVX/nvxcu.h) against the oracle, bit for bit: colour convert, Gaussian pyramid,
Harris / FAST track and pyramidal Lucas-Kanade on random frame sizes (odd ones, a few pixels up to ~700 wide), level counts,
windows 3..32, termination modes, iteration limits, cell sizes, thresholds, FAST arc lengths, masks, tracked-point lists with
lost and out-of-frame items, capacities that truncate, both item types, aligned and unaligned pitches, with and without
initial estimates and calcError.  What the fixed cases of tests/test_gpu_thin_layer.py cannot enumerate; a second generator beside
tools/fuzz_parity.py (which runs for a time budget): counted cases, frames from 1 x 1 up, noise and plateau images, KEYPOINTF
tracked lists, estimates, calcError, flat second frames.  tests/test_gpu_fuzz.py runs a few hundred cases of it in the GPU suite.
usage: tools/fuzz_front_end.py [cases] [seed]   (GPU; prints a summary line, exits non-zero on the first difference)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import oracle as O  # noqa: E402
from test_gpu_thin_layer import _run_lk, _run_track, sync  # noqa: E402
from visualslam_amd import nvxcu, synth  # noqa: E402
from visualslam_amd.devmem import DeviceImage, DevicePyramid, current_target  # noqa: E402

OK = nvxcu.NVXCU_SUCCESS


def frame(rng, w, h):
    kind = rng.integers(0, 4)
    seed = int(rng.integers(1 << 30))
    if kind == 0:
        return rng.integers(0, 256, (h, w), dtype=np.uint8)                     # noise: corners everywhere
    g = synth.gray_frame(synth.base_frame(w, h, seed), 0, seed)
    if kind == 1:
        g = (g.astype(np.int32) // 64 * 64).astype(np.uint8)                    # plateaus: ties in the cell selection
    return g


def pyramid_levels(rng, w, h):
    lv = 1
    while lv < 7 and min(w, h) >> lv >= 1:
        lv += 1
    return int(rng.integers(1, lv + 1))


def case_convert_pyramid(rng, stats):
    w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
    rgbx = synth.color_rgbx(w, h, seed=int(rng.integers(1 << 30)))
    aligned = rng.random() < 0.5
    src = DeviceImage(w, h, nvxcu.NVXCU_DF_IMAGE_RGBX, pitch=None if aligned else 4 * w + 4 * int(rng.integers(1, 4))).upload(rgbx)
    dst = DeviceImage(w, h, pitch=None if aligned else w + int(rng.integers(1, 4)))
    assert nvxcu.color_convert(src.desc, dst.desc, current_target()) == OK
    sync()
    gray = O.color_convert(rgbx)
    if not np.array_equal(dst.download(), gray):
        return "colour convert %dx%d aligned=%s" % (w, h, aligned)
    levels = pyramid_levels(rng, w, h)
    pyr = DevicePyramid(w, h, levels, pitch_pad=None if aligned else int(rng.integers(1, 9)))
    pyr.levels[0].upload(gray)
    assert nvxcu.gaussian_pyramid(pyr.desc, current_target()) == OK
    sync()
    want = O.gaussian_pyramid(gray, levels)
    for lvl, (g, x) in enumerate(zip(pyr.download(), want)):
        if not np.array_equal(g, x):
            return "pyramid %dx%d levels=%d aligned=%s level %d" % (w, h, levels, aligned, lvl)
    stats["convert+pyramid"] += 1
    return None


def case_track(rng, stats):
    w, h = int(rng.integers(7, 500)), int(rng.integers(7, 400))
    gray = frame(rng, w, h)
    cell = int(rng.choice([1, 2, 3, 5, 7, 18, 18, 18, 33, 64, 65, 130]))
    cap = int(rng.choice([1, 50, 700, 4000, 20000]))
    item = O.KEYPOINTF if rng.random() < 0.5 else O.POINT2F
    it = nvxcu.NVXCU_TYPE_KEYPOINTF if item == O.KEYPOINTF else nvxcu.NVXCU_TYPE_POINT2F
    mask = None
    if rng.random() < 0.4:
        mask = (rng.random((h, w)) < rng.choice([0.02, 0.5, 0.98])).astype(np.uint8) * int(rng.choice([1, 255]))
    tracked = None
    if rng.random() < 0.6:
        n = int(rng.choice([0, 1, 40, 900, 9000]))
        xy = np.stack([rng.uniform(-4, w + 4, n), rng.uniform(-4, h + 4, n)], 1).astype(np.float32)
        if n:
            xy[rng.random(n) < 0.2] = (-1.0, -1.0)
            m = rng.random(n) < 0.1
            xy[m] = np.floor(xy[m])                                               # integer positions: cell borders
        if item == O.KEYPOINTF:
            tr = np.zeros(n, O.keypointf_dtype)
            tr["x"], tr["y"] = xy[:, 0], xy[:, 1]
            tr["strength"] = rng.uniform(0, 1e4, n)
            tr["tracking_status"] = rng.choice([0, 1, 255], n)
            tr["error"] = rng.uniform(0, 9, n)
            tracked = tr
        else:
            tracked = xy
        if n == 0:
            tracked = None if rng.random() < 0.5 else tracked
    pitch = None if rng.random() < 0.6 else w + int(rng.integers(1, 8))
    what = dict(w=w, h=h, cell=cell, cap=cap, item=item, mask=mask is not None, tracked=None if tracked is None else len(tracked), pitch=pitch)
    if rng.random() < 0.5:
        k = float(rng.choice([0.04, 0.01, 0.15]))
        thr = float(rng.choice([0.0, 5.0, 100.0, 2000.0, 1e7]))
        got, gtot = _run_track("harris", gray, mask, tracked, it, cell, cap, pitch=pitch, k=k, threshold=thr)
        want, wtot, _ = O.harris_track(gray, mask, tracked, k, thr, cell, cap, item_type=item)
        what.update(det="harris", k=k, thr=thr)
        stats["harris track"] += 1
    else:
        n9 = int(rng.choice([9, 10, 11, 12]))
        thr = int(rng.choice([1, 5, 25, 80, 254]))
        got, gtot = _run_track("fast", gray, mask, tracked, it, cell, cap, pitch=pitch, type_n=n9, threshold=thr)
        want, wtot, _ = O.fast_track(gray, mask, tracked, n9, thr, cell, cap, item_type=item)
        what.update(det="fast", n=n9, thr=thr)
        stats["fast track"] += 1
    if gtot != wtot or len(got) != len(want) or got.tobytes() != want.tobytes():
        return "track %s: %d/%d corners against %d/%d" % (what, len(got), gtot, len(want), wtot)
    stats["corners"] += len(want)
    return None


def case_lk(rng, stats):
    w, h = int(rng.integers(12, 420)), int(rng.integers(12, 320))
    levels = pyramid_levels(rng, w, h)
    seed = int(rng.integers(1 << 30))
    base = synth.base_frame(w, h, seed)
    dx, dy = float(rng.uniform(-6, 6)), float(rng.uniform(-6, 6))
    f0 = np.clip(np.rint(base), 0, 255).astype(np.uint8)
    f1 = np.clip(np.rint(synth.shift_bilinear(base, dx, dy)), 0, 255).astype(np.uint8)
    if rng.random() < 0.15:
        f1 = np.full_like(f1, 77)                                                  # a flat image: every point is lost
    p0, p1 = O.gaussian_pyramid(f0, levels), O.gaussian_pyramid(f1, levels)
    n = int(rng.choice([0, 1, 63, 64, 65, 500, 3000]))
    xy = np.stack([rng.uniform(-3, w + 2, n), rng.uniform(-3, h + 2, n)], 1).astype(np.float32)
    if n:
        xy[rng.random(n) < 0.1] = (-1.0, -1.0)
        m = rng.random(n) < 0.15
        xy[m] = np.floor(xy[m])
        xy[rng.random(n) < 0.05, 0] = np.float32(w - 1)
    item = O.KEYPOINTF if rng.random() < 0.4 else O.POINT2F
    pts = xy
    if item == O.KEYPOINTF:
        kp = np.zeros(n, O.keypointf_dtype)
        kp["x"], kp["y"] = xy[:, 0], xy[:, 1]
        kp["strength"] = rng.uniform(1, 100, n)
        kp["scale"], kp["orientation"] = 1.5, 0.25
        kp["tracking_status"] = rng.choice([0, 1, 255], n)
        kp["error"] = 3.0
        pts = kp
    est = None
    if rng.random() < 0.3 and n:
        est = pts.copy()
        if item == O.KEYPOINTF:
            est["x"] += np.float32(dx) * rng.uniform(0.5, 1.5, n).astype(np.float32)
            est["y"] += np.float32(dy)
        else:
            est[:, 0] += np.float32(dx)
            est[:, 1] += np.float32(dy) * rng.uniform(0.5, 1.5, n).astype(np.float32)
    win = int(rng.integers(3, 33))
    term = int(rng.choice([nvxcu.NVXCU_TERM_CRITERIA_BOTH, nvxcu.NVXCU_TERM_CRITERIA_ITERATIONS, nvxcu.NVXCU_TERM_CRITERIA_EPSILON]))
    iters = int(rng.choice([1, 2, 5, 5, 5, 20, 40]))
    eps = float(rng.choice([0.01, 0.1, 0.0005]))
    pad = None if rng.random() < 0.6 else int(rng.integers(1, 9))
    calc = 1 if (item == O.KEYPOINTF and rng.random() < 0.5) else 0
    what = dict(w=w, h=h, levels=levels, n=n, item=item, win=win, term=term, iters=iters, eps=eps, pad=pad, est=est is not None, calc=calc, d=(dx, dy))
    if n == 0:
        stats["pyrlk"] += 1
        return None                                                                 # (empty arrays: tests/test_gpu_thin_layer.py::test_pyrlk_empty_input)
    got = _run_lk(p0, p1, pts, est, item, term, eps, iters, win, pitch_pad=pad, calc_error=calc)
    want = O.optical_flow_pyr_lk(p0, p1, pts, est, term, eps, iters, win, item_type=item, calc_error=bool(calc))
    if got.tobytes() != want.tobytes():
        return "pyrlk %s" % (what,)
    stats["pyrlk"] += 1
    stats["points"] += n
    return None


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2025
    rng = np.random.default_rng(seed)
    stats = {"convert+pyramid": 0, "harris track": 0, "fast track": 0, "corners": 0, "pyrlk": 0, "points": 0}
    for c in range(cases):
        fn = (case_convert_pyramid, case_track, case_track, case_lk, case_lk)[c % 5]
        bad = fn(rng, stats)
        if bad:
            print("DIFFERS (case %d, seed %d): %s" % (c, seed, bad))
            return 1
        if c % 50 == 49:
            print("  ... %d cases" % (c + 1), flush=True)
    print("fuzz_front_end: seed %d, %d cases identical to the oracle: %s" % (seed, cases, stats))
    return 0


if __name__ == "__main__":
    sys.exit(main())
