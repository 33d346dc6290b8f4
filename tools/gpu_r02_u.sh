#!/bin/bash
# round 2, call U: block odd-even reduction solver (bandchol3.hpp) on random band + arrow systems
set -o pipefail
mkdir -p gpurun_out/r02
: > gpurun_out/r02/u_cr.log
for cfg in "40 3 5" "37 4 17" "20 1 1" "50 2 3" "334 9 17" "333 9 17" "335 9 17" "1000 9 17" "2672 9 17" "334 10 9" "334 11 6" "100 12 0" "64 8 17"; do
  timeout -k 10 120 tools/ubench/cr_solve.out $cfg 20 >> gpurun_out/r02/u_cr.log 2>&1; echo "cfg $cfg rc=$?" >> gpurun_out/r02/u_cr.log
done
cat gpurun_out/r02/u_cr.log
