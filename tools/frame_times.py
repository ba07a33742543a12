#!/usr/bin/env python3
"""Per-frame time of ONE camera stream's graph over its first frames (wall clock around vxProcessGraph), with the number of tracked
points: does the work per frame change while a tracker settles?  usage: frame_times.py [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from visualslam_amd import ovx, synth  # noqa: E402
from visualslam_amd.thin_tracker import Params  # noqa: E402
from visualslam_amd.vx_tracker import VxFeatureTracker  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
W, H = 1920, 1080
p = Params(pyr_levels=5, lk_win_size=21, array_capacity=4096, harris_thresh=100.0)
base = synth.base_frame(W, H, 1234)
L = ovx.lib()
ctx = L.vxCreateContext()
ring = []
for t in range(n + 1):
    img = L.vxCreateImage(ctx, W, H, ovx.VX_DF_IMAGE_RGBX)
    ovx.write_image(img, synth.to_rgbx(synth.gray_frame(base, t, 1234)))
    ring.append(img)
trk = VxFeatureTracker(ctx, p, W, H, ring[0], None)
torch.cuda.synchronize()
for k in range(n):
    t0 = time.perf_counter()
    trk.track(ring[k + 1], None)
    dt = time.perf_counter() - t0
    cur = trk.curr()
    print("frame %2d: %6.1f us, %d points, %d tracked" % (k + 1, dt * 1e6, len(cur), int((cur[:, 0] >= 0).sum())), flush=True)
