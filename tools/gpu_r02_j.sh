#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_deterministic.py tests/test_gpu_multirank.py -q -m gpu -x -p no:cacheprovider -k "deterministic or windowed_driver_on_two or reproducible or default_mode" > gpurun_out/r02/det_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -40 gpurun_out/r02/det_tests.log
exit $rc
