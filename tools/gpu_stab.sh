#!/bin/bash
# The reference's video stabilizer demo (oracle/_ref/nvx_demo_video_stabilizer, unmodified sources) on 1080p synthetic
# jittering frames under rocprofv3 --kernel-trace --stats: per-kernel durations of the whole stabilizer graph
# (tracker front-end + FindHomography + WarpPerspective) and the demo's own per-node times.
set -euo pipefail
R=${1:-r01_stab}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p "$OUT" /tmp/vstab/bin /tmp/vstab/data
cd "$GRAFT_REPO_ROOT"
cp oracle/_ref/nvx_demo_video_stabilizer /tmp/vstab/bin/
timeout -k 10 120 python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
from visualslam_amd import synth
w, h, n = 1920, 1080, 40
rng = np.random.default_rng(3)
base = synth.base_frame(w + 32, h + 32, 1234)
with open("/tmp/vstab/frames.rgbx", "wb") as f:
    for dx, dy in rng.integers(-4, 5, (n, 2)):
        f.write(synth.to_rgbx(np.ascontiguousarray(base[16 + dy:16 + dy + h, 16 + dx:16 + dx + w])).tobytes())
print("frames written")
PY
export TMPDIR=/tmp LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/visualslam_amd/lib:${LD_LIBRARY_PATH:-} VSA_HEADLESS_FRAMES=39
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o stab -- /tmp/vstab/bin/nvx_demo_video_stabilizer "--source=rawrgbx:///tmp/vstab/frames.rgbx?width=1920&height=1080" -n 5 --crop=0.05 --nvxio_fps_limit=1000 > "$OUT/demo_stdout.log" 2> "$OUT/trace.err"
grep -A11 "Graph Time" "$OUT/demo_stdout.log" | tail -12
find "$OUT" -name '*kernel_stats.csv' -exec head -12 {} \;
echo STAB_OK
