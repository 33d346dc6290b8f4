#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_gpu_precision1.py tests/test_gpu_windowed.py -q -m gpu -p no:cacheprovider > gpurun_out/r02/prec1_windowed.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -60 gpurun_out/r02/prec1_windowed.log
exit $rc
