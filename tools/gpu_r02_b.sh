#!/bin/bash
# round 2, call B: new cfg1/cfg2 parity tests, bench line with the reworked CPU baseline, MFMA occupancy ubench, rocSOLVER comparator
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 300 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "cfg1 or cfg2" -p no:cacheprovider > gpurun_out/r02/cfg12.log 2>&1; rc=$?; echo "cfg12 rc=$rc"; tail -3 gpurun_out/r02/cfg12.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 ./tools/ubench/mfma_f64_occ.out > gpurun_out/r02/mfma_f64_occ.log 2>&1; rc=$?; echo "ubench rc=$rc"; cat gpurun_out/r02/mfma_f64_occ.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > gpurun_out/r02/bench_b.json 2> gpurun_out/r02/bench_b.err; rc=$?; echo "bench rc=$rc"; tail -c 3000 gpurun_out/r02/bench_b.json
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/rocsolver_comparator.py cfg3 metric > gpurun_out/r02/rocsolver.log 2>&1; rc=$?; echo "rocsolver rc=$rc"; tail -5 gpurun_out/r02/rocsolver.log
exit $rc
