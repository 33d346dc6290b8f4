#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_gpu_precision1.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/prec1_tests3.log 2>&1; rc=$?; echo "precision tests rc=$rc"; tail -30 gpurun_out/r02/prec1_tests3.log
timeout -k 10 300 python bench.py --steps 50 --precision 1 --no-cpu-baseline > gpurun_out/r02/benchG_p1.json 2> gpurun_out/r02/benchG_p1.err; echo "bench rc=$?"
python - <<PY
import json
j = json.load(open("gpurun_out/r02/benchG_p1.json"))
print("   value %.3e obs/s  ms_per_step %.4f  kernel_ms %.4f frac %.4f solve %s" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["roofline"]["frac"], {k: j["solve"][k] for k in ("iterations", "seconds", "final_cost", "final_rms_reproj_px")}))
PY
exit $rc
