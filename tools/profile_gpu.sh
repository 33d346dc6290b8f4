#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes (HBM traffic, SQ issue counters) of bench.py's sweeps.
# Usage: tools/profile_gpu.sh <tag> [bench args]   -> gpurun_out/prof_<tag>/...
# (PMC passes are separate runs with no trace domain beside them; FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950.)
set -e
TAG=${1:-r02}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-solve $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq1 -- $CMD > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq3 -- $CMD > $OUT/pmc_sq3.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
