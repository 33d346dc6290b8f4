#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 1100 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/r02/gpu_tests_i.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -30 gpurun_out/r02/gpu_tests_i.log
exit $rc
