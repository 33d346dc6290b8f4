#!/bin/bash
# round 2, call T: Schur product of k_sweep3 on the fp64 matrix pipe (LIFCAL_SCHUR_MFMA=1): parity, then bench A/B
set -o pipefail
mkdir -p gpurun_out/r02
LIFCAL_SCHUR_MFMA=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/t_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -8 gpurun_out/r02/t_tests.log
[ $rc -eq 0 ] || exit $rc
for m in 0 1; do
  for wl in metric_web metric; do
    LIFCAL_SCHUR_MFMA=$m timeout -k 10 300 python bench.py --no-cpu-baseline --no-solve --workload $wl > gpurun_out/r02/t_bench_m${m}_$wl.json 2> gpurun_out/r02/t_bench.err; echo "bench mfma=$m $wl rc=$?"
    python - <<PY
import json; d=json.load(open("gpurun_out/r02/t_bench_m${m}_$wl.json")); print("mfma=$m $wl kernel_ms", d["roofline"]["kernel_ms"], "step ms", d["ms_per_step"])
PY
  done
done
