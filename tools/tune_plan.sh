#!/bin/bash
# Sweeps the planner's cost-model constants (LIFCAL_PLAN_COST="step,pass,lane") and the block count on the GPU box; prints the
# dominant kernel's time for each setting (sustained clocks: bench.py's preheat, 300 timed steps).  Usage: tools/tune_plan.sh  (through gpurun)
for cost in "4700,21500,65" "4700,30000,65" "4700,15000,65" "3900,21500,65" "5500,21500,65" "4700,21500,120" "4700,21500,30" "3900,28000,100"; do
  LIFCAL_PLAN_COST=$cost python bench.py --steps 300 --no-cpu-baseline --no-solve 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cost', '$cost', 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'ms_per_step', round(d['ms_per_step'],4))"
done
for nb in 240 252 256 264 272 288 320 384 512; do
  LIFCAL_V2_BLOCKS=$nb python bench.py --steps 300 --no-cpu-baseline --no-solve 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('blocks', $nb, 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'ms_per_step', round(d['ms_per_step'],4))"
done
for sp in 3 4 5 6 8; do
  LIFCAL_GROUP_SPLIT=$sp python bench.py --steps 300 --no-cpu-baseline --no-solve 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('group split', $sp, 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'ms_per_step', round(d['ms_per_step'],4))"
done
python bench.py --steps 300 --no-cpu-baseline --no-solve 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('default', 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'ms_per_step', round(d['ms_per_step'],4))"
