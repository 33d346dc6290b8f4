#!/bin/bash
# round 2, call AA: pass-top waits of k_sweep3 (descriptors consumed early, dangling prefetches consumed before the emission): parity, bench, stamps
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py tests/test_gpu_deterministic.py tests/test_gpu_precision1.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/aa_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r02/aa_tests.log
[ $rc -eq 0 ] || exit $rc
for wl in metric_web metric; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-solve --workload $wl > gpurun_out/r02/aa_bench_$wl.json 2> gpurun_out/r02/aa_bench.err; echo "bench $wl rc=$?"
  python - <<PY
import json; d=json.load(open("gpurun_out/r02/aa_bench_$wl.json")); print("$wl kernel_ms", d["roofline"]["kernel_ms"], "step ms", d["ms_per_step"], "value", d["value"])
PY
done
timeout -k 10 200 python tools/stamps.py metric 2>&1 | grep -v amdgpu | head -24
