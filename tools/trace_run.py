#!/usr/bin/env python3
"""Reads a VSA_DRIVER_TRACE file (one line per camera stream: host completion time of every frame of the run, microseconds)
and prints where the run's time goes: the first frame of every stream, frames completed per millisecond, the last frames."""
import sys

import numpy as np

t = np.array([[float(x) for x in l.split()] for l in open(sys.argv[1]) if l.strip()])
start, t = t[:, 0], t[:, 1:]          # first column: when the stream's host thread began the run
s, n = t.shape
print("thread started (us): min %.0f median %.0f max %.0f" % (start.min(), np.median(start), start.max()))
print("first frame's own time (us), streams in order: " + " ".join("%.0f" % v for v in (t[:, 0] - start)))
end = t.max()
print(f"{s} streams x {n} frames, run {end/1e3:.2f} ms -> {s*n/end*1e6:.0f} frames/s")
print("first frame done (us): min %.0f median %.0f max %.0f" % (t[:, 0].min(), np.median(t[:, 0]), t[:, 0].max()))
print("last frame done (us):  min %.0f median %.0f max %.0f  (spread %.0f)" % (t[:, -1].min(), np.median(t[:, -1]), t[:, -1].max(), t[:, -1].max() - t[:, -1].min()))
per = np.diff(np.concatenate([np.zeros((s, 1)), t], axis=1), axis=1)
print("frame period per stream (us), by frame index: " + " ".join("%.0f" % v for v in np.median(per, axis=0)))
edges = np.arange(0, end + 1000, 1000)
h, _ = np.histogram(t.ravel(), edges)
print("frames completed per ms: " + " ".join(str(v) for v in h))
