#!/bin/bash
# round 2, call Z: library with the 512-thread factor kernel: solver parity tests, multi-rank tests, solve timing
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_gpu_paths.py tests/test_gpu_configs.py tests/test_gpu_multirank.py tests/test_gpu_scale.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/z_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r02/z_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/solve_timing.py metric metric_web cfg4 > gpurun_out/r02/z_solve.log 2>&1; echo "solve timing rc=$?"; grep -v amdgpu.ids gpurun_out/r02/z_solve.log | cut -c1-300
