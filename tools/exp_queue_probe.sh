#!/bin/bash
set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_qprobe
mkdir -p $OUT
cd /tmp
i=0
for ARGS in "32 none 0" "32 none 2" "32 -1 0" "32 -1 2" "32 1 0"; do
  i=$((i+1))
  timeout -s KILL 60 rocprofv3 --kernel-trace --output-format csv -d $OUT/p$i -o q -- $GRAFT_REPO_ROOT/tools/bin/queue_probe $ARGS > $OUT/p$i.log 2>&1 || true
  echo "== queue_probe $ARGS" >> $OUT/map.txt
  python3 - $(find $OUT/p$i -name 'q_kernel_trace.csv' | head -1) >> $OUT/map.txt <<'PY'
import csv, sys, collections
m = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "k_probe" in r["Kernel_Name"]:
        m.setdefault(r["Stream_Id"], set()).add(r["Queue_Id"])
cnt = collections.Counter(q for v in m.values() for q in v)
print("stream -> queue:", " ".join(f"{s}:{'/'.join(sorted(v))}" for s, v in m.items()))
print("streams per queue:", dict(cnt))
PY
done
cat $OUT/map.txt
