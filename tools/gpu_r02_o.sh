#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/o_multirank.log 2>&1; rc=$?; echo "multirank rc=$rc"; tail -25 gpurun_out/r02/o_multirank.log
exit $rc
