#!/bin/bash
# round 2, call C: k_sweep3 with two waves per role (256-thread workgroups, 128-lane passes, two workgroups per CU): parity, then timing
set -o pipefail
mkdir -p gpurun_out/r02
export LIFCAL_SWEEP_WAVES=2
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py tests/test_gpu_stress.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/w2_parity.log 2>&1; rc=$?; echo "w2 parity rc=$rc"; tail -5 gpurun_out/r02/w2_parity.log
[ $rc -eq 0 ] || exit $rc
for cfg in "2 512" "2 768" "2 1024" "4 256"; do
  set -- $cfg
  LIFCAL_SWEEP_WAVES=$1 LIFCAL_V2_BLOCKS=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-solve --steps 50 > gpurun_out/r02/bench_w$1_b$2.json 2> gpurun_out/r02/bench_w$1_b$2.err; rc=$?
  echo "waves $1 blocks $2 rc=$rc"; python - <<PY
import json
try:
    j = json.load(open("gpurun_out/r02/bench_w$1_b$2.json"))
    print("   value %.3e obs/s  ms_per_step %.4f  kernel_ms %.4f  tables %.4f  tail %.4f" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["roofline"]["whole_sweep"]["ms_tables"], j["roofline"]["whole_sweep"]["ms_schur"]))
except Exception as e:
    print("   no bench line:", e)
PY
  [ $rc -eq 0 ] || exit $rc
done
