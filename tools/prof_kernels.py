#!/usr/bin/env python3
"""Run one primitive of the 1080p workload repeatedly (for rocprofv3 --kernel-trace / --pmc passes).
usage: prof_kernels.py {lk|harris|pyramid|cvt|frame|graph} [reps] [c2|c3]      (c3: the 4K / 7 levels / FAST-9 / 10 k features workload)
  frame: whole frames through the thin layer (five launches: nvxcu primitives cannot fuse)
  graph: whole frames of ONE camera stream through the OpenVX graph as bench.py builds it (the fused launches)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from visualslam_amd import nvxcu, synth  # noqa: E402
from visualslam_amd.devmem import DeviceImage  # noqa: E402
from visualslam_amd.thin_tracker import Params, ThinFeatureTracker  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "lk"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W, H = 1920, 1080
p = Params(pyr_levels=5, lk_win_size=21, array_capacity=4096)
if len(sys.argv) > 3 and sys.argv[3] == "c3":
    W, H = 3840, 2160
    p = Params(pyr_levels=7, lk_win_size=21, array_capacity=10000, use_harris_detector=False, fast_type=9, fast_thresh=25)
base = synth.base_frame(W, H, 1234)
if what == "graph":
    from visualslam_amd import ovx
    from visualslam_amd.vx_tracker import VxFeatureTracker
    L = ovx.lib()
    ctx = L.vxCreateContext()
    ovx.check(L.vxGetStatus(ctx), "vxCreateContext")
    ring = []
    for t in range(4):
        img = L.vxCreateImage(ctx, W, H, ovx.VX_DF_IMAGE_RGBX)
        ovx.write_image(img, synth.to_rgbx(synth.gray_frame(base, t, 1234)))
        ring.append(img)
    trk = VxFeatureTracker(ctx, p, W, H, ring[0], None)
    for k in range(reps + 2):
        trk.track(ring[(k + 1) % 4], None)
    torch.cuda.synchronize()
    print("done", what, reps)
    sys.exit(0)
frames = [DeviceImage(W, H, nvxcu.NVXCU_DF_IMAGE_RGBX).upload(synth.to_rgbx(synth.gray_frame(base, t))) for t in range(2)]
trk = ThinFeatureTracker(p, W, H)
trk.init(frames[0].desc)
trk.track(frames[1].desc)
torch.cuda.synchronize()
t = trk.target
for _ in range(reps):
    if what == "lk":
        trk.kp_curr_list.buf.copy_(trk.prev_pts.buf)
        nvxcu.optical_flow_pyr_lk(trk.prev_pyr.desc, trk.curr_pyr.desc, trk.prev_pts.desc, trk.kp_curr_list.desc,
                                  nvxcu.NVXCU_TERM_CRITERIA_BOTH, 0.01, 5, 21, t)
    elif what == "harris":
        trk._detect(trk.kp_curr_list.desc)
    elif what == "pyramid":
        nvxcu.gaussian_pyramid(trk.curr_pyr.desc, t)
    elif what == "cvt":
        nvxcu.color_convert(frames[0].desc, trk.curr_pyr.levels[0].desc, t)
    else:
        trk.track(frames[_ % 2].desc)
torch.cuda.synchronize()
print("done", what, reps)
