#!/bin/bash
# round 2, call D: two waves per role with the role flip of the CU's second workgroup; phase stamps of both variants
set -o pipefail
mkdir -p gpurun_out/r02
for cfg in "2 512" "4 256"; do
  set -- $cfg
  LIFCAL_SWEEP_WAVES=$1 LIFCAL_V2_BLOCKS=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-solve --steps 50 > gpurun_out/r02/benchD_w$1_b$2.json 2> gpurun_out/r02/benchD_w$1_b$2.err; rc=$?
  echo "waves $1 blocks $2 rc=$rc"; python - <<PY
import json
try:
    j = json.load(open("gpurun_out/r02/benchD_w$1_b$2.json"))
    print("   value %.3e obs/s  ms_per_step %.4f  kernel_ms %.4f" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"]))
except Exception as e:
    print("   no bench line:", e)
PY
  [ $rc -eq 0 ] || exit $rc
  LIFCAL_SWEEP_WAVES=$1 LIFCAL_V2_BLOCKS=$2 timeout -k 10 200 python tools/stamps.py metric > gpurun_out/r02/stamps_w$1.txt 2>&1; rc=$?; echo "stamps rc=$rc"
  [ $rc -eq 0 ] || exit $rc
done
cat gpurun_out/r02/stamps_w2.txt
