#!/usr/bin/env python3
"""The FAST cell pass + merge (nvxcuFastTrack) launched back to back on one 4K frame: time per call from HIP events over a run long
enough for the GPU to be at its clocks.   usage: fast_probe.py [calls] [width height]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from visualslam_amd import nvxcu, synth  # noqa: E402
from visualslam_amd.devmem import DeviceArray, DeviceImage, Scratch, current_target  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
gray = synth.gray_frame(synth.base_frame(W, H, 1234), 0)
img = DeviceImage(W, H).upload(gray)
out = DeviceArray(10000)
tmp = Scratch(nvxcu.fast_track_buf_size(W, H, nvxcu.NVXCU_TYPE_POINT2F, 10000, 18).dev_buf_size)
t = current_target()
for _ in range(50):
    assert nvxcu.fast_track(img.desc, out.desc, None, None, 9, 25, 18, None, tmp.desc, t) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(calls):
    nvxcu.fast_track(img.desc, out.desc, None, None, 9, 25, 18, None, tmp.desc, t)
e1.record()
torch.cuda.synchronize()
print(f"{W}x{H}: {e0.elapsed_time(e1) / calls * 1000:.1f} us per nvxcuFastTrack (cell pass + merge), {out.num_items()} corners, "
      f"mode {os.environ.get('VSA_EXP_FAST_MODE', '-')} band {os.environ.get('VSA_EXP_FS_BAND', '-')}")
