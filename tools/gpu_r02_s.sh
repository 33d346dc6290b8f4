#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -p no:cacheprovider -k "streaming" > gpurun_out/r02/s_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -25 gpurun_out/r02/s_tests.log
exit $rc
