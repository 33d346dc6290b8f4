#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py tests/test_gpu_deterministic.py tests/test_gpu_multirank.py tests/test_gpu_windowed.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/r_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -6 gpurun_out/r02/r_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/solve_timing.py cfg3 metric > gpurun_out/r02/r_solve.log 2>&1; echo "solve timing rc=$?"; grep -v amdgpu.ids gpurun_out/r02/r_solve.log | cut -c1-260
