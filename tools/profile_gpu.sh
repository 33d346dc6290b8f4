#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for HBM traffic of bench.py.
# Usage: tools/profile_gpu.sh <tag>   -> gpurun_out/prof_<tag>/...
set -e
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-solve"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
