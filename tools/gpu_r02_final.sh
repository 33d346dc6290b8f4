#!/bin/bash
# end-of-round evidence: rocprofv3 kernel stats + HBM / SQ counters of the default bench, the default bench line itself, solve kernel stats
set -o pipefail
mkdir -p gpurun_out/r02
bash tools/profile_gpu.sh r02_final > gpurun_out/r02/profile_final.log 2>&1; rc=$?; echo "profile rc=$rc"; tail -42 gpurun_out/r02/profile_final.log | cut -c1-160
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > gpurun_out/r02/bench_final.json 2> gpurun_out/r02/bench_final.err; rc=$?; echo "bench rc=$rc"; cat gpurun_out/r02/bench_final.json | cut -c1-3000
[ $rc -eq 0 ] || exit $rc
bash tools/profile_solve.sh > gpurun_out/r02/q_profile_solve.log 2>&1; echo "profile_solve rc=$?"; tail -14 gpurun_out/r02/q_profile_solve.log | cut -c1-150
