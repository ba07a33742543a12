#!/usr/bin/env python3
"""Randomised parity sweep of the OpenVX graph path (the fused launches): short sequences through VxFeatureTracker (the call sequence of
demos/feature_tracker/feature_tracker.cpp) and through the oracle's frame loop at random sizes, pyramid depths, windows, cell sizes,
capacities, detectors and masks; previous, tracked and merged point arrays must be identical on every frame.
usage: fuzz_graph.py [seconds] [seed]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

from oracle.pipeline import OracleTracker  # noqa: E402
from visualslam_amd import ovx, synth  # noqa: E402
from visualslam_amd.thin_tracker import Params  # noqa: E402
from visualslam_amd.vx_tracker import VxFeatureTracker  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rng = np.random.default_rng(seed)
L = ovx.lib()
t_end = time.time() + budget
case = frames_total = fused = 0
while time.time() < t_end:
    case += 1
    w, h = int(rng.integers(40, 900)), int(rng.integers(40, 600))
    levels = int(rng.integers(1, 7))
    while levels > 1 and min(w, h) >> (levels - 1) < 4:
        levels -= 1
    harris = bool(rng.random() < 0.7)
    p = Params(pyr_levels=levels, lk_win_size=int(rng.integers(3, 33)), lk_num_iters=int(rng.integers(1, 8)),
               array_capacity=int(rng.integers(1, 3000)), detector_cell_size=int(rng.choice([3, 7, 18, 18, 18, 30])),
               use_harris_detector=harris, harris_k=0.04, harris_thresh=float(rng.choice([100.0, 2000.0])),
               fast_type=int(rng.integers(9, 13)), fast_thresh=int(rng.choice([10, 25, 40])))
    n = int(rng.integers(2, 6))
    s = int(rng.integers(1, 1000))
    base = synth.base_frame(w, h, s)
    frames = [synth.to_rgbx(synth.gray_frame(base, t, s)) for t in range(n)]
    use_mask = rng.random() < 0.3
    mask_np = None
    desc = f"case {case}: {w}x{h}, {levels} levels, win {p.lk_win_size}, cell {p.detector_cell_size}, capacity {p.array_capacity}, " \
           f"{'harris' if harris else 'fast-%d' % p.fast_type}, mask {use_mask}, seed {s}"
    ctx = L.vxCreateContext()
    ovx.check(L.vxGetStatus(ctx), "vxCreateContext")
    try:
        mask = None
        if use_mask:
            mask_np = np.zeros((h, w), np.uint8)
            mask_np[h // 5:, : (4 * w) // 5] = 255
            mask = L.vxCreateImage(ctx, w, h, ovx.VX_DF_IMAGE_U8)
            ovx.write_image(mask, mask_np)
        imgs = []
        for f in frames:
            img = L.vxCreateImage(ctx, w, h, ovx.VX_DF_IMAGE_RGBX)
            ovx.write_image(img, f)
            imgs.append(img)
        trk = VxFeatureTracker(ctx, p, w, h, imgs[0], mask)
        ref = OracleTracker(p)
        ref.init(frames[0], mask_np)
        assert trk.merged().tobytes() == ref.prev_pts.tobytes(), "first frame"
        for t in range(1, n):
            m_now = mask if (use_mask and t % 2 == 1) else None        # the mask is rebound every frame, sometimes to none
            trk.track(imgs[t], m_now)
            old, cur = ref.track(frames[t], mask_np if m_now else None)
            assert trk.prev().tobytes() == old.tobytes(), f"frame {t}: previous"
            assert trk.curr().tobytes() == cur.tobytes(), f"frame {t}: tracked"
            assert trk.merged().tobytes() == ref.prev_pts.tobytes(), f"frame {t}: merged"
            frames_total += 1
        fused += 1 if harris else 0
    except AssertionError as e:
        print("MISMATCH", desc, e, flush=True)
        sys.exit(1)
    finally:
        assert L.vxReleaseContext(C.byref(C.c_void_p(ctx))) == ovx.VX_SUCCESS
    if case % 50 == 0:
        print(f"{case} sequences identical so far", flush=True)
print(f"FUZZ_GRAPH_OK seed {seed}: {case} sequences ({frames_total} tracked frames, {fused} of the sequences with the Harris cell pass "
      f"inside the PyrLK launch) identical to the oracle")
