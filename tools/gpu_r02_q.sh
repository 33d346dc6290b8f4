#!/bin/bash
# round 2, call Q: kernel stats of a full solve with the twisted factorisation; rocSOLVER comparator again
set -o pipefail
mkdir -p gpurun_out/r02
bash tools/profile_solve.sh > gpurun_out/r02/q_profile_solve.log 2>&1; echo "profile_solve rc=$?"; tail -16 gpurun_out/r02/q_profile_solve.log | cut -c1-150
timeout -k 10 300 python tools/rocsolver_comparator.py cfg3 metric > gpurun_out/r02/q_rocsolver.log 2>&1; echo "rocsolver rc=$?"; grep -v amdgpu gpurun_out/r02/q_rocsolver.log | cut -c1-400
