#!/bin/bash
# instruction-cache and issue-stall counters of the sweep kernels (run on the GPU box via gpurun)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-solve"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1 || true
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2>&1 || true
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/c -- $CMD > $OUT/c.log 2>&1 || true
python3 - <<PY
import csv, glob
from collections import defaultdict
for d in "abc":
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]
            acc[k][r["Counter_Name"]][0] += float(r["Counter_Value"]); acc[k][r["Counter_Name"]][1] += 1
    for k, v in acc.items():
        if "sweep" in k or "front4" in k or "back4" in k:
            print(k, {c: round(x[0] / max(x[1], 1)) for c, x in v.items()})
PY
