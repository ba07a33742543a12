#!/usr/bin/env python3
"""One camera stream of the 1080p workload with nvxFindHomographyNode in its graph, frame after frame, for
rocprofv3 --kernel-trace: uncontended per-kernel durations.   usage: prof_hg.py [frames]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402,F401  (first: one HIP runtime per process)

from visualslam_amd import ovx, synth  # noqa: E402
from visualslam_amd.thin_tracker import Params  # noqa: E402
from visualslam_amd.vx_tracker import VxFeatureTracker  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
W, H = 1920, 1080
L = ovx.lib()
ctx = L.vxCreateContext()
ovx.check(L.vxGetStatus(ctx), "vxCreateContext")
p = Params(pyr_levels=5, lk_win_size=21, array_capacity=4096)
base = synth.base_frame(W, H, 1234)
ring = []
for t in range(4):
    img = L.vxCreateImage(ctx, W, H, ovx.VX_DF_IMAGE_RGBX)
    ovx.write_image(img, synth.to_rgbx(synth.gray_frame(base, t, 1234)))
    ring.append(img)
trk = VxFeatureTracker(ctx, p, W, H, ring[0], None, homography=True)
for k in range(reps):
    trk.track(ring[(k + 1) % 4], None)
Hm, mask = trk.homography()
print("done", reps, "inliers", int((mask != 0).sum()), "of", len(mask))
