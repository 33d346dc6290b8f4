#!/bin/bash
# round 2, call AB: profiling through the dominant kernel's own dispatch events: bench on both scenes + precision / deterministic arms, smoke tests
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "sweep" > gpurun_out/r02/ab_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r02/ab_tests.log
[ $rc -eq 0 ] || exit $rc
for arg in "--workload metric_web" "--workload metric" "--workload metric_web --precision 1" "--workload metric_web --deterministic"; do
  tag=$(echo $arg | tr -d ' -'); timeout -k 10 300 python bench.py --no-cpu-baseline --no-solve $arg > gpurun_out/r02/ab_bench_$tag.json 2> gpurun_out/r02/ab_bench.err; echo "bench $arg rc=$?"
  python - <<PY
import json; d=json.load(open("gpurun_out/r02/ab_bench_$tag.json")); r=d["roofline"]; print("$arg: kernel_ms %.4f step ms %.4f value %.3e frac %.4f outside %.4f" % (r["kernel_ms"], d["ms_per_step"], d["value"], r["frac"], r["whole_sweep"]["ms_outside_dominant_kernel"]))
PY
done
