#!/usr/bin/env python3
"""configs[4] under contention, every frame checked: S camera streams (own graph and HIP stream each, a Python thread per stream --
ctypes drops the interpreter lock across vxProcessGraph) run the tracker graph with the FindHomography node for F frames at the same
time, and EVERY frame's H, inlier mask and merged feature list of every stream is compared with the oracle's, bit for bit.  The
hypothesis workgroups of one FindHomography launch sit on different XCDs and hand their results to the refining workgroup past the
L2s (homography.hip); with 16 streams the launches of different streams overlap all the time.  tests/test_gpu_configs.py runs the
same thing with 12 x 10 frames; this is the long form.
usage: tools/soak_homography.py [streams] [frames]      (GPU)"""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402,F401

import oracle as O  # noqa: E402
from oracle.pipeline import OracleTracker  # noqa: E402
from visualslam_amd import ovx, synth  # noqa: E402
from visualslam_amd.thin_tracker import Params  # noqa: E402
from visualslam_amd.vx_tracker import VxFeatureTracker  # noqa: E402

L = ovx.lib()


def main():
    n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    w, h, ring_len = 800, 600, 5
    ctx = L.vxCreateContext()
    ovx.check(L.vxGetStatus(ctx), "vxCreateContext")
    p = Params(pyr_levels=4, lk_num_iters=5, lk_win_size=21, array_capacity=2000, detector_cell_size=18,
               use_harris_detector=True, harris_k=0.04, harris_thresh=100.0)
    base = synth.base_frame(w, h, 777)
    frames = [synth.to_rgbx(synth.gray_frame(base, t, 777)) for t in range(ring_len)]
    ring = []
    for f in frames:
        img = L.vxCreateImage(ctx, w, h, ovx.VX_DF_IMAGE_RGBX)
        ovx.write_image(img, f)
        ring.append(img)
    phases = [s % ring_len for s in range(n_streams)]
    trackers = [VxFeatureTracker(ctx, p, w, h, ring[ph], None, homography=True, homography_first=bool(s & 1)) for s, ph in enumerate(phases)]
    assert all(ovx.graph_fusions(t.g)["merge_in_homography"] == 1 for t in trackers)
    got = [[] for _ in range(n_streams)]
    errors = []
    start = threading.Barrier(n_streams)

    def run(s):
        try:
            start.wait()
            for k in range(n_frames):
                trackers[s].track(ring[(phases[s] + 1 + k) % ring_len], None)
                Hg, mg = trackers[s].homography()
                got[s].append((Hg.tobytes(), mg.tobytes(), trackers[s].merged().tobytes()))
        except Exception as e:      # noqa: BLE001
            errors.append((s, repr(e)))

    ths = [threading.Thread(target=run, args=(s,)) for s in range(n_streams)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errors:
        print("ERRORS", errors)
        return 1
    want = {}
    for ph in sorted(set(phases)):
        ref = OracleTracker(p, omp=True)
        ref.init(frames[ph], None)
        seq = []
        for k in range(n_frames):
            old, cur = ref.track(frames[(ph + 1 + k) % ring_len], None)
            Hw, mw, _ = O.find_homography(old, cur, 3.0, 2000, 10, 0.995, 0.45)
            seq.append((Hw.tobytes(), mw.tobytes(), ref.prev_pts.tobytes()))
        want[ph] = seq
    bad = [(s, k, i) for s, ph in enumerate(phases) for k in range(n_frames) for i in range(3) if got[s][k][i] != want[ph][k][i]]
    if bad:
        print("DIFFERENCES (stream, frame, 0 = H / 1 = mask / 2 = merged features):", bad[:20], "...", len(bad))
        return 1
    print("soak_homography: %d streams x %d frames at %dx%d run concurrently: every frame's H, inlier mask and merged feature list "
          "identical to the oracle (%d checks)" % (n_streams, n_frames, w, h, 3 * n_streams * n_frames))
    return 0


if __name__ == "__main__":
    sys.exit(main())
