#!/bin/bash
# round 2, call M: deterministic mode after the wave-per-entry reduction; planner cost-model grid on the metric_web workload
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 300 python -m pytest tests/test_gpu_deterministic.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/m_det.log 2>&1; rc=$?; echo "det rc=$rc"; tail -3 gpurun_out/r02/m_det.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline --deterministic > gpurun_out/r02/benchM_det.json 2> gpurun_out/r02/benchM.err; echo "det bench rc=$?"
python - <<PY
import json
j = json.load(open("gpurun_out/r02/benchM_det.json"))
print("   det: value %.3e obs/s  ms_per_step %.4f  kernel_ms %.4f solve %.4fs" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["solve"]["seconds"]))
PY
for cost in "4700,21500,65" "4700,21500,130" "4700,21500,250" "4700,12000,65" "4700,35000,65" "3500,21500,65" "6500,21500,65" "4700,35000,200"; do
  LIFCAL_PLAN_COST=$cost timeout -k 10 200 python bench.py --steps 40 --no-cpu-baseline --no-solve > gpurun_out/r02/benchM_cost.json 2>> gpurun_out/r02/benchM.err; rc2=$?
  python - <<PY
import json
j = json.load(open("gpurun_out/r02/benchM_cost.json"))
print("   cost $cost: kernel_ms %.4f  ms_per_step %.4f" % (j["roofline"]["kernel_ms"], j["ms_per_step"]))
PY
done
