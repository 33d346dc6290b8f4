#!/bin/bash
# round 2, call AC: one host synchronisation per LM iteration (deferred sweep read-back): whole GPU suite, solve timing
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/ac_suite.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -6 gpurun_out/r02/ac_suite.log
[ $rc -eq 0 ] || exit $rc
for sy in 0 1; do
  LIFCAL_SYNC_SWEEP=$sy timeout -k 10 300 python tools/solve_timing.py cfg3 metric metric_web cfg4 > gpurun_out/r02/ac_solve_sync$sy.log 2>&1; echo "solve timing sync_sweep=$sy rc=$?"; grep -v "amdgpu.ids\|oracle" gpurun_out/r02/ac_solve_sync$sy.log | cut -c1-300
done
