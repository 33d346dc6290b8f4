#!/bin/bash
# round 2, call P: twisted band factorisation — parity first (small), then the whole suite, then solve timing
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 300 python -m pytest tests/test_gpu_paths.py -x -q -m gpu -p no:cacheprovider -k twisted > gpurun_out/r02/p_twisted.log 2>&1; rc=$?; echo "twisted rc=$rc"; tail -25 gpurun_out/r02/p_twisted.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/p_suite.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -8 gpurun_out/r02/p_suite.log
[ $rc -eq 0 ] || exit $rc
for tw in 1 0; do
  LIFCAL_TWISTED=$tw timeout -k 10 200 python tools/solve_timing.py cfg3 metric > gpurun_out/r02/p_solve_tw$tw.log 2>&1; echo "solve timing twisted=$tw rc=$?"; grep -v amdgpu.ids gpurun_out/r02/p_solve_tw$tw.log | cut -c1-260
done
