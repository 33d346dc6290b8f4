#!/bin/bash
# round 2, call L: pass-top loads hoisted above barrier P1: parity, then timing on both workloads
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py tests/test_gpu_stress.py tests/test_gpu_precision1.py tests/test_gpu_deterministic.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/l_parity.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -5 gpurun_out/r02/l_parity.log
[ $rc -eq 0 ] || exit $rc
for v in "" "--workload metric" "--precision 1"; do
  timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline --no-solve $v > "gpurun_out/r02/benchL_$(echo $v | tr -d ' -').json" 2> gpurun_out/r02/benchL.err; rc2=$?
  echo "bench [$v] rc=$rc2"; python - <<PY
import json
j = json.load(open("gpurun_out/r02/benchL_$(echo $v | tr -d ' -').json"))
print("   value %.3e obs/s  ms_per_step %.4f  kernel_ms %.4f frac %.4f" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["roofline"]["frac"]))
PY
done
