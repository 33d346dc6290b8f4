#!/bin/bash
# round 2, call AE: k_cost with batched loads: whole GPU suite, solve timing + kernel stats
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/ae_suite.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -5 gpurun_out/r02/ae_suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/solve_timing.py cfg3 metric metric_web cfg4 > gpurun_out/r02/ae_solve.log 2>&1; echo "solve timing rc=$?"; grep -v "amdgpu.ids\|oracle" gpurun_out/r02/ae_solve.log | cut -c1-300
bash tools/profile_solve.sh > gpurun_out/r02/ae_profile_solve.log 2>&1; echo "profile_solve rc=$?"; tail -13 gpurun_out/r02/ae_profile_solve.log | cut -c1-150
