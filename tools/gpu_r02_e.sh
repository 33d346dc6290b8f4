#!/bin/bash
# round 2, call E: options.precision = 1 — parity tests, then the bench in both arithmetics
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_gpu_precision1.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/prec1_tests.log 2>&1; rc=$?; echo "precision tests rc=$rc"; tail -30 gpurun_out/r02/prec1_tests.log
for prec in 1 0; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 50 --precision $prec > gpurun_out/r02/benchE_p$prec.json 2> gpurun_out/r02/benchE_p$prec.err; rc2=$?
  echo "precision $prec rc=$rc2"; python - <<PY
import json
try:
    j = json.load(open("gpurun_out/r02/benchE_p$prec.json"))
    print("   value %.3e obs/s  ms_per_step %.4f  kernel_ms %.4f frac %.4f solve %s" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["roofline"]["frac"], {k: j["solve"][k] for k in ("iterations", "final_cost", "final_rms_reproj_px")}))
except Exception as e:
    print("   no bench line:", e)
PY
done
exit $rc
