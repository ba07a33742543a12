#!/usr/bin/env python3
"""Soak: N frames of the headline workload (1080p, 5 levels, 4096 points, 21x21, Harris track; the bench's frame ring) through ONE
OpenVX graph -- the fused launches -- and through the oracle's frame loop (OpenMP build); the tracked and the merged point arrays
must be identical, byte for byte, on every frame.   usage: soak.py [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from oracle.pipeline import OracleTracker  # noqa: E402
from visualslam_amd import ovx, synth  # noqa: E402
from visualslam_amd.thin_tracker import Params  # noqa: E402
from visualslam_amd.vx_tracker import VxFeatureTracker  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, H, RING = 1920, 1080, 64
p = Params(pyr_levels=5, lk_win_size=21, array_capacity=4096, harris_thresh=100.0)
base = synth.base_frame(W, H, 1234)
frames = [synth.to_rgbx(synth.gray_frame(base, t, 1234)) for t in range(RING)]
L = ovx.lib()
ctx = L.vxCreateContext()
ovx.check(L.vxGetStatus(ctx), "vxCreateContext")
imgs = []
for f in frames:
    img = L.vxCreateImage(ctx, W, H, ovx.VX_DF_IMAGE_RGBX)
    ovx.write_image(img, f)
    imgs.append(img)
trk = VxFeatureTracker(ctx, p, W, H, imgs[0], None)
ref = OracleTracker(p, omp=True)
ref.init(frames[0], None)
assert trk.merged().tobytes() == ref.prev_pts.tobytes()
t0 = time.time()
lost = 0
for k in range(1, n + 1):
    trk.track(imgs[k % RING], None)
    _, cur = ref.track(frames[k % RING], None)
    got = trk.curr()
    assert got.tobytes() == cur.tobytes(), f"frame {k}: tracked points differ"
    assert trk.merged().tobytes() == ref.prev_pts.tobytes(), f"frame {k}: merged points differ"
    lost += int((cur[:, 0] < 0).sum())
    if k % 50 == 0:
        print(f"frame {k}: identical so far, {len(cur)} points, {lost} lost and re-detected in total, {time.time() - t0:.0f} s", flush=True)
print(f"SOAK_OK {n} frames identical to the oracle")
