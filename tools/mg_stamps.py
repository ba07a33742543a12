#!/usr/bin/env python3
"""Phase timeline of the merge kernel (k_merge_compact) on BASELINE configs[2] (4K, FAST-9, 10k features) or, with
'1080p', on the headline workload (needs tools/bin/libvisualslam_amd_stamps.so, built by tools/build_stamps.sh)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VISUALSLAM_AMD_LIB"] = os.path.join(ROOT, "tools", "bin", "libvisualslam_amd_stamps.so")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from visualslam_amd import nvxcu, synth  # noqa: E402
from visualslam_amd.devmem import DeviceImage  # noqa: E402
from visualslam_amd.thin_tracker import Params, ThinFeatureTracker  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "1080p":
    W, H = 1920, 1080
    p = Params(pyr_levels=5, lk_win_size=21, array_capacity=4096)
else:
    W, H = 3840, 2160
    p = Params(pyr_levels=7, lk_win_size=21, array_capacity=10000, use_harris_detector=False, fast_type=9, fast_thresh=25)
base = synth.base_frame(W, H, 1234)
frames = [DeviceImage(W, H, nvxcu.NVXCU_DF_IMAGE_RGBX).upload(synth.to_rgbx(synth.gray_frame(base, t))) for t in range(2)]
trk = ThinFeatureTracker(p, W, H)
L = nvxcu.lib()
names = ["start", "flags cleared (+ keys loaded, register variant)", "tracked points marked (wave 0)", "barrier", "cells counted (wave 0)",
         "barrier", "written out (wave 0)", "survivors written (wave 0)"]
for k in range(6):
    if k == 0:
        trk.init(frames[0].desc)
    else:
        trk.track(frames[k % 2].desc)
    torch.cuda.synchronize()
    buf = np.zeros(16, dtype=np.uint64)
    rc = L.visualslam_amd_debug_mg_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
    assert rc == 0, rc
    if k not in (0, 4, 5):
        continue
    t0 = int(buf[0])
    prev = t0
    for tv, i in sorted((int(buf[i]), i) for i in range(8)):
        print("frame %d  %-50s +%7d cycles  (at %7d)" % (k, names[i], tv - prev, tv - t0))
        prev = tv
