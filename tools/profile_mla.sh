#!/bin/bash
# Runs on the GPU box (via gpurun): timing of the projectPointsToRawImage row + rocprofv3 kernel stats of the same command.
set -e
TAG=${1:-r01_f1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 tools/mla_timing.py metric 10 > $OUT/timing.json 2> $OUT/timing.err
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/mla_timing.py metric 10 --no-cpu > $OUT/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/timing.json; cat $OUT/summary.txt
