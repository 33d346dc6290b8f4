#!/bin/bash
# One entry point for what runs on the GPU box (through `gpurun -- tools/gpurun.sh <recipe> [args]`); logs go to gpurun_out/<tag>/.
#   tests [pytest args]        GPU test-suite (or a selection), one process
#   bench [bench args]         bench.py, JSON line to gpurun_out/<tag>/bench.json
#   profile <name> [bench args]  tools/profile_gpu.sh (kernel trace + PMC passes of the bench sweeps) -> gpurun_out/prof_<name>/
#   quick [bench args]         parity selection + short bench (the inner loop of kernel work)
#   run <script.py> [args]     one python tool (tools/*.py) under a 15-minute limit, output to gpurun_out/<tag>/run.log
# TAG (environment) names the output directory (default: the recipe).
set -o pipefail
R=${1:-quick}; shift || true
TAG=${TAG:-$R}
OUT=gpurun_out/$TAG
mkdir -p $OUT
case $R in
  tests)   timeout -k 10 1100 python -m pytest tests -m gpu -x -q "$@" > $OUT/tests.log 2>&1; rc=$?; tail -15 $OUT/tests.log; exit $rc ;;
  bench)   timeout -k 10 600 python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err; rc=$?; cat $OUT/bench.json; tail -5 $OUT/bench.err; exit $rc ;;
  profile) N=${1:-r03}; shift || true; timeout -k 10 900 tools/profile_gpu.sh $N "$@" ;;
  quick)   timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; tail -15 $OUT/tests.log; [ $rc -ne 0 ] && exit $rc
           timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-solve "$@" > $OUT/bench.json 2> $OUT/bench.err; rc=$?; cat $OUT/bench.json; tail -5 $OUT/bench.err; exit $rc ;;
  run)     timeout -k 10 900 python "$@" > $OUT/run.log 2>&1; rc=$?; tail -40 $OUT/run.log; exit $rc ;;
  *) echo "unknown recipe $R"; exit 2 ;;
esac
