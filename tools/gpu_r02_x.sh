#!/bin/bash
# round 2, call X: whole GPU suite with the block odd-even reduction as the default for long sequences; solve timing both ways; solve profile
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/x_suite.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -6 gpurun_out/r02/x_suite.log
[ $rc -eq 0 ] || exit $rc
for cr in 1 0; do
  LIFCAL_CR=$cr timeout -k 10 300 python tools/solve_timing.py cfg3 metric metric_web cfg4 > gpurun_out/r02/x_solve_cr$cr.log 2>&1; echo "solve timing cr=$cr rc=$?"; grep -v amdgpu.ids gpurun_out/r02/x_solve_cr$cr.log | cut -c1-300
done
bash tools/profile_solve.sh > gpurun_out/r02/x_profile_solve.log 2>&1; echo "profile_solve rc=$?"; tail -16 gpurun_out/r02/x_profile_solve.log | cut -c1-150
