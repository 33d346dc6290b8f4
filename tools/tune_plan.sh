#!/bin/bash
# tuning aid: bench line for several lane split sizes / block counts (run on the GPU box via gpurun)
run() { timeout -k 10 120 python bench.py --no-cpu-baseline --no-solve --steps 30 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '%.3f G obs/s  kernel %.4f ms' % (j['value']/1e9, j['roofline']['kernel_ms']))"; }
run auto
for t in 0 3 4 5 6 8; do export LIFCAL_GROUP_SPLIT=$t; run split=$t; done
unset LIFCAL_GROUP_SPLIT
for b in 128 192 255 256 320 384 512; do export LIFCAL_V2_BLOCKS=$b; run blocks=$b; done
