#!/bin/bash
# kernel-trace only (no PMC passes): per-kernel average durations of bench.py's sweeps -> gpurun_out/trace_<tag>/stats.txt
TAG=${1:-t}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-solve "$@" > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
python3 - "$f" > $OUT/stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:9.2f} us  total {float(r["TotalDurationNs"])/1e6:9.3f} ms')
PY
cat $OUT/stats.txt
