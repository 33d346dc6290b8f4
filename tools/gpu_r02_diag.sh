#!/bin/bash
# round 2 diagnostic of the 2-rank recalib hang (stress case 10): single-rank solve first, then the world-2 test with the host trace on
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 240 python tools/diag_solve.py 10 > gpurun_out/r02/case10_w1.log 2>&1; rc=$?; echo "w1 rc=$rc"; tail -5 gpurun_out/r02/case10_w1.log
[ $rc -eq 0 ] || exit $rc
LIFCAL_TRACE=1 timeout -k 10 300 python -m pytest tests/test_gpu_multirank.py -x -q -s -m gpu -k "deformed and 10-2" -p no:cacheprovider > gpurun_out/r02/case10_w2.log 2>&1; rc=$?; echo "w2 rc=$rc"; tail -5 gpurun_out/r02/case10_w2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/gpu_tests_a.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -15 gpurun_out/r02/gpu_tests_a.log
exit $rc
