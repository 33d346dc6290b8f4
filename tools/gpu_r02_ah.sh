#!/bin/bash
# round 2, call AH: camera-block reduction of k_sweep3 through DPP row sums: parity / deterministic / precision / multirank tests, bench
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py tests/test_gpu_deterministic.py tests/test_gpu_precision1.py tests/test_gpu_stress.py tests/test_gpu_multirank.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/ah_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r02/ah_tests.log
[ $rc -eq 0 ] || exit $rc
for arg in "--workload metric_web" "--workload metric" "--workload metric_web --precision 1" "--workload metric_web --deterministic"; do
  tag=$(echo $arg | tr -d ' -'); timeout -k 10 300 python bench.py --no-cpu-baseline --no-solve $arg > gpurun_out/r02/ah_bench_$tag.json 2> gpurun_out/r02/ah_bench.err; echo "bench $arg rc=$?"
  python - <<PY
import json; d=json.load(open("gpurun_out/r02/ah_bench_$tag.json")); r=d["roofline"]; print("$arg: kernel_ms %.4f step ms %.4f value %.3e" % (r["kernel_ms"], d["ms_per_step"], d["value"]))
PY
done
