#!/bin/bash
# quick on-GPU check used while tuning k_sweep2: parity on two small scenes, phase stamps and the bench line at the metric point
set -e
timeout -k 10 120 python tools/debug_compare.py tiny_f06 cfg2 > gpurun_out/dc.log 2>&1
timeout -k 10 200 python tools/stamps.py metric > gpurun_out/stamps.log 2>&1
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2>&1
