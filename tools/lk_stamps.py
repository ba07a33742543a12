#!/usr/bin/env python3
"""Per-wave timeline of k_pyrlk on the 1080p workload (needs tools/bin/libvisualslam_amd_stamps.so,
built by tools/build_stamps.sh).  Prints dispatch ramp, wave lifetimes, per-SIMD residency."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VISUALSLAM_AMD_LIB"] = os.environ.get("VSA_STAMPS_LIB") or os.path.join(ROOT, "tools", "bin", "libvisualslam_amd_stamps.so")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from visualslam_amd import nvxcu, synth  # noqa: E402
from visualslam_amd.devmem import DeviceImage  # noqa: E402
from visualslam_amd.thin_tracker import Params, ThinFeatureTracker  # noqa: E402

W, H = 1920, 1080
p = Params(pyr_levels=5, lk_win_size=21, array_capacity=4096)
base = synth.base_frame(W, H, 1234)
frames = [DeviceImage(W, H, nvxcu.NVXCU_DF_IMAGE_RGBX).upload(synth.to_rgbx(synth.gray_frame(base, t))) for t in range(2)]
trk = ThinFeatureTracker(p, W, H)
trk.init(frames[0].desc)
for _ in range(3):
    trk.track(frames[1].desc)
    trk.track(frames[0].desc)
torch.cuda.synchronize()
for _ in range(3):
    nvxcu.optical_flow_pyr_lk(trk.prev_pyr.desc, trk.curr_pyr.desc, trk.prev_pts.desc, trk.kp_curr_list.desc,
                              nvxcu.NVXCU_TERM_CRITERIA_BOTH, 0.01, 5, 21, trk.target)
torch.cuda.synchronize()
L = nvxcu.lib()
n = 4096
buf = np.zeros(8 * n, dtype=np.uint64)
rc = L.visualslam_amd_debug_lk_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
assert rc == 0, rc
s = buf.reshape(n, 8)
t0, t1 = s[:, 0].astype(np.int64), s[:, 1].astype(np.int64)
hw, xcc = (s[:, 2] & 0xffffffff).astype(np.int64), (s[:, 2] >> 32).astype(np.int64) & 0xf
wall, iters = (s[:, 3] >> 32).astype(np.int64), (s[:, 3] & 0xff).astype(np.int64)
first_head = ((s[:, 3] >> 8) & 0xffffff).astype(np.int64)
life = t1 - t0
ph = s[:, 4:8].astype(np.int64)
print("phase cycles per wave (mean): level head + P commit %.0f, template pass %.0f, solve + J commit %.0f, iterations %.0f (%.0f per iteration); outside %.0f"
      % (*ph.mean(axis=0), ph[:, 3].sum() / max(1, iters.sum()), (t1 - t0 - ph.sum(axis=1)).mean()))
print("wave start .. top level's template tile in LDS (cycles): mean %.0f p10 %.0f p50 %.0f p90 %.0f" % (first_head.mean(), *np.percentile(first_head, [10, 50, 90])))
print("waves", n, "iters/wave mean %.2f min %d max %d" % (iters.mean(), iters.min(), iters.max()))
print("lifetime cycles: mean %.0f  p10 %.0f  p50 %.0f  p90 %.0f  max %.0f" % (life.mean(), *np.percentile(life, [10, 50, 90]), life.max()))
print("wall ticks (100 MHz) per wave: mean %.1f  max %d  -> shader clock ~ %.0f MHz" % (wall.mean(), wall.max(), life.mean() / wall.mean() * 100))
# cycle counters are per XCD: align per XCC
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    b = t0[m].min()
    st, en = t0[m] - b, t1[m] - b
    print(f"xcc {x}: waves {m.sum():4d} start p50 {np.percentile(st,50):7.0f} p90 {np.percentile(st,90):7.0f} max {st.max():7d} | "
          f"end p10 {np.percentile(en,10):7.0f} p50 {np.percentile(en,50):7.0f} p90 {np.percentile(en,90):7.0f} max {en.max():7d}")
# residency: waves per (xcc, se, cu, simd)
simd = (hw >> 4) & 3
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
u, cnt = np.unique(key, return_counts=True)
print("distinct SIMDs used", len(u), "waves per SIMD: min", cnt.min(), "max", cnt.max(), "hist", np.bincount(cnt).tolist())
cukey = key // 4
u2, cnt2 = np.unique(cukey, return_counts=True)
print("distinct CUs", len(u2), "waves per CU hist", np.bincount(cnt2).tolist())
# concurrency: for the busiest SIMD, show intervals
k = u[np.argmax(cnt)]
m = key == k
b = t0[xcc == xcc[m][0]].min()
for a_, b_, i_ in sorted(zip((t0[m] - b).tolist(), (t1[m] - b).tolist(), iters[m].tolist())):
    print("   busiest SIMD wave: start %7d end %7d iters %d" % (a_, b_, i_))
