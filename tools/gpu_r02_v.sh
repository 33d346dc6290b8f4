#!/bin/bash
# round 2, call V: bandchol3 harness again + per-kernel stats at the metric geometry
set -o pipefail
mkdir -p gpurun_out/r02
bash tools/gpu_r02_u.sh | grep -v "^cfg"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02/v_prof -- $GRAFT_REPO_ROOT/tools/ubench/cr_solve.out 334 9 17 20 > $GRAFT_REPO_ROOT/gpurun_out/r02/v_prof.log 2>&1
python3 - <<'PY'
import csv, glob, os
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r02/v_prof/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        print(row["Name"][:60], row["Calls"], row["AverageNs"], row["TotalDurationNs"], row["Percentage"])
PY
