#!/bin/bash
# round 2, call W: block odd-even reduction in the library: parity vs the chain, then solve timing both ways
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_paths.py -x -q -m gpu -p no:cacheprovider -k "odd_even" > gpurun_out/r02/w_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 gpurun_out/r02/w_tests.log
[ $rc -eq 0 ] || exit $rc
for cr in 1 0; do
  LIFCAL_CR=$cr timeout -k 10 300 python tools/solve_timing.py cfg3 metric metric_web > gpurun_out/r02/w_solve_cr$cr.log 2>&1; echo "solve timing cr=$cr rc=$?"; grep -v amdgpu.ids gpurun_out/r02/w_solve_cr$cr.log | cut -c1-260
done
