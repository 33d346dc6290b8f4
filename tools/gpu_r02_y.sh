#!/bin/bash
# round 2, call Y: factor kernel with two rows per thread (512 threads) against four (256); phase stamps of k_sweep3 on both metric scenes
set -o pipefail
mkdir -p gpurun_out/r02
: > gpurun_out/r02/y_cr.log
for t in 512 256; do for c in "334 9 17" "1000 9 17" "2672 9 17" "64 8 17" "40 3 5" "334 10 9"; do LIFCAL_CR_THREADS=$t timeout -k 10 60 tools/ubench/cr_solve.out $c 20 >> gpurun_out/r02/y_cr.log 2>&1; done; done
cut -c1-60,150-230 gpurun_out/r02/y_cr.log
for wl in metric; do timeout -k 10 200 python tools/stamps.py $wl > gpurun_out/r02/y_stamps_$wl.log 2>&1; echo "stamps $wl rc=$?"; grep -v amdgpu gpurun_out/r02/y_stamps_$wl.log; done
