#!/bin/bash
# round 2, call F: precision = 1 tests again; bench on the metric_web workload (lenses chosen by lifcal_mla_project), both arithmetics
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_gpu_precision1.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/prec1_tests2.log 2>&1; rc=$?; echo "precision tests rc=$rc"; tail -30 gpurun_out/r02/prec1_tests2.log
for prec in 0 1; do
  timeout -k 10 300 python bench.py --steps 50 --precision $prec $( [ $prec -eq 1 ] && echo --no-cpu-baseline ) > gpurun_out/r02/benchF_p$prec.json 2> gpurun_out/r02/benchF_p$prec.err; rc2=$?
  echo "precision $prec rc=$rc2"; tail -3 gpurun_out/r02/benchF_p$prec.err; python - <<PY
import json
try:
    j = json.load(open("gpurun_out/r02/benchF_p$prec.json"))
    print("   ", j["config"]["workload"][:90]); print("   value %.3e obs/s  ms_per_step %.4f  kernel_ms %.4f frac %.4f solve %s" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["roofline"]["frac"], {k: j["solve"][k] for k in ("iterations", "seconds", "final_cost", "final_rms_reproj_px")}))
    if j.get("cpu_baseline"): print("   cpu", {k: j["cpu_baseline"][k] for k in ("value", "cores")}, j["cpu_baseline"]["arms"]["analytic"]["value"])
except Exception as e:
    print("   no bench line:", e)
PY
done
exit $rc
