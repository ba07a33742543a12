#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: colour convert, Gaussian pyramid, Harris / FAST track (masks, tracked points, cell sizes,
capacities) and PyrLK (windows 3..32, levels 1..7, terminations, item types, pitches, motions) at random sizes and parameters, each
case compared with the oracle byte for byte, through the thin C ABI, with the helpers of tests/test_gpu_thin_layer.py.
usage: fuzz_parity.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle as O  # noqa: E402
import test_gpu_thin_layer as T  # noqa: E402
from visualslam_amd import nvxcu, synth  # noqa: E402
from visualslam_amd.devmem import DeviceImage, DevicePyramid, current_target  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
rng = np.random.default_rng(seed)
counts = {"colour convert": 0, "pyramid": 0, "harris track": 0, "fast track": 0, "pyrlk": 0}
t_end = time.time() + budget
case = 0
while time.time() < t_end:
    case += 1
    kind = ["colour convert", "pyramid", "harris track", "fast track", "pyrlk", "pyrlk"][case % 6]
    w, h = int(rng.integers(24, 700)), int(rng.integers(24, 500))
    s = int(rng.integers(1, 1 << 30))
    desc = f"case {case} ({kind}, {w}x{h}, seed {s})"
    try:
        if kind == "colour convert":
            f = synth.color_rgbx(w, h, seed=s)
            src = DeviceImage(w, h, nvxcu.NVXCU_DF_IMAGE_RGBX).upload(f)
            dst = DeviceImage(w, h)
            assert nvxcu.color_convert(src.desc, dst.desc, current_target()) == 0
            T.sync()
            assert dst.download().tobytes() == O.color_convert(f).tobytes()
        elif kind == "pyramid":
            levels = int(rng.integers(1, 8))
            while levels > 1 and min(w, h) >> (levels - 1) < 2:
                levels -= 1
            g = T._scene(w, h, s % 1000)
            pyr = DevicePyramid(w, h, levels)
            pyr.levels[0].upload(g)
            assert nvxcu.gaussian_pyramid(pyr.desc, current_target()) == 0
            T.sync()
            for a, b in zip(pyr.download(), O.gaussian_pyramid(g, levels)):
                assert a.tobytes() == b.tobytes()
        elif kind in ("harris track", "fast track"):
            g = T._scene(w, h, s % 1000)
            cell = int(rng.choice([1, 3, 7, 18, 18, 18, 25, 70]))
            cap = int(rng.integers(1, 5000))
            it, oit = (nvxcu.NVXCU_TYPE_KEYPOINTF, O.KEYPOINTF) if rng.random() < 0.5 else (nvxcu.NVXCU_TYPE_POINT2F, O.POINT2F)
            mask = T._center_mask(w, h) if rng.random() < 0.4 else None
            tracked = None
            if rng.random() < 0.6:
                n = int(rng.integers(1, 3000))
                xy = np.stack([rng.uniform(-3, w + 2, n), rng.uniform(-3, h + 2, n)], 1).astype(np.float32)
                xy[::7] = (-1.0, -1.0)
                if oit == O.POINT2F:
                    tracked = xy
                else:
                    tracked = np.zeros(n, O.keypointf_dtype)
                    tracked["x"], tracked["y"] = xy[:, 0], xy[:, 1]
                    tracked["strength"] = rng.uniform(1, 100, n)
                    tracked["tracking_status"] = rng.choice([0, 255], n)
            if kind == "harris track":
                k, thr = float(rng.choice([0.04, 0.01, 0.06])), float(rng.choice([5.0, 100.0, 2000.0]))
                got, gtot = T._run_track("harris", g, mask, tracked, it, cell, cap, k=k, threshold=thr)
                want, wtot, _ = O.harris_track(g, mask, tracked, k, thr, cell, cap, item_type=oit)
            else:
                n9, thr = int(rng.integers(9, 13)), int(rng.choice([5, 10, 25, 40]))
                got, gtot = T._run_track("fast", g, mask, tracked, it, cell, cap, type_n=n9, threshold=thr)
                want, wtot, _ = O.fast_track(g, mask, tracked, n9, thr, cell, cap, item_type=oit)
            assert gtot == wtot and got.tobytes() == want.tobytes()
        else:
            levels = int(rng.integers(1, 8))
            while levels > 1 and min(w, h) >> (levels - 1) < 4:
                levels -= 1
            win = int(rng.integers(3, 33))
            oit = O.KEYPOINTF if rng.random() < 0.3 else O.POINT2F
            dx, dy = float(rng.uniform(-6, 6)), float(rng.uniform(-6, 6))
            p0, p1, pts = T._lk_case(w, h, levels, s % 1000, dx, dy, int(rng.integers(1, 1500)), item_type=oit)
            term = [nvxcu.NVXCU_TERM_CRITERIA_BOTH, nvxcu.NVXCU_TERM_CRITERIA_ITERATIONS, nvxcu.NVXCU_TERM_CRITERIA_EPSILON][int(rng.integers(0, 3))]
            iters, eps = int(rng.integers(1, 12)), float(rng.choice([0.01, 0.03, 0.1]))
            pad = int(rng.choice([0, 0, 1, 3, 8])) or None
            ce = 1 if (oit == O.KEYPOINTF and rng.random() < 0.5) else 0
            got = T._run_lk(p0, p1, pts, None, oit, term, eps, iters, win, pitch_pad=pad, calc_error=ce)
            want = O.optical_flow_pyr_lk(p0, p1, pts, None, int(term), eps, iters, win, item_type=oit, calc_error=bool(ce))
            assert got.tobytes() == want.tobytes(), f"win {win} levels {levels} term {term} iters {iters} eps {eps} pad {pad} item {oit} calc_error {ce}"
        counts[kind] += 1
    except AssertionError as e:
        print("MISMATCH", desc, e, flush=True)
        sys.exit(1)
    if case % 50 == 0:
        print(f"{case} cases identical so far", flush=True)
print("FUZZ_OK seed", seed, "cases identical to the oracle:", ", ".join(f"{k} {v}" for k, v in counts.items()))
