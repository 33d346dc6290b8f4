#!/bin/bash
# round 2, call K: profile of the round's default bench (kernel stats, HBM traffic, SQ counters), then bench variants
set -o pipefail
mkdir -p gpurun_out/r02
bash tools/profile_gpu.sh r02_final > gpurun_out/r02/profile_final.log 2>&1; rc=$?; echo "profile rc=$rc"; tail -45 gpurun_out/r02/profile_final.log
[ $rc -eq 0 ] || exit $rc
for v in "" "--deterministic" "--precision 1" "--workload metric"; do
  timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline $v > "gpurun_out/r02/benchK_$(echo $v | tr -d ' -').json" 2> gpurun_out/r02/benchK.err; rc2=$?
  echo "bench [$v] rc=$rc2"; python - <<PY
import json
j = json.load(open("gpurun_out/r02/benchK_$(echo $v | tr -d ' -').json"))
print("   value %.3e obs/s  ms_per_step %.4f  kernel_ms %.4f frac %.4f  solve it %s %.4fs" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["roofline"]["frac"], j["solve"]["iterations"], j["solve"]["seconds"]))
PY
done
