#!/bin/bash
# kernel-trace stats of one full LM solve at the metric point (run on the GPU box via gpurun)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_solve
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/solve_timing.py metric > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:12]:
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} {float(r['Percentage']):6.2f}%")
PY
