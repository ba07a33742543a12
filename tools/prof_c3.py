#!/usr/bin/env python3
"""BASELINE configs[2] frame after frame (4K frames, 7 levels, FAST-9 track, 10k features, PyrLK 21x21), one stream,
for rocprofv3 --kernel-trace --stats.   usage: prof_c3.py [frames]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from visualslam_amd import nvxcu, synth  # noqa: E402
from visualslam_amd.devmem import DeviceImage  # noqa: E402
from visualslam_amd.thin_tracker import Params, ThinFeatureTracker  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W, H = 3840, 2160
p = Params(pyr_levels=7, lk_win_size=21, array_capacity=10000, use_harris_detector=False, fast_type=9, fast_thresh=25)
base = synth.base_frame(W, H, 1234)
frames = [DeviceImage(W, H, nvxcu.NVXCU_DF_IMAGE_RGBX).upload(synth.to_rgbx(synth.gray_frame(base, t))) for t in range(2)]
trk = ThinFeatureTracker(p, W, H)
trk.init(frames[0].desc)
for k in range(reps):
    trk.track(frames[(k + 1) % 2].desc)
torch.cuda.synchronize()
cur = trk.curr_features()
print("done", reps, "points", len(cur), "tracked", int((cur[:, 0] >= 0).sum()))
