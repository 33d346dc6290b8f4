#!/bin/bash
cd "$(dirname "$0")/.."
grep -E "S:|solve" gpurun_out/dc.log; cat gpurun_out/stamps.log
python -c "
import json; j=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1]); print('bench G obs/s %.3f  ms/step %.4f  kernel ms %.4f' % (j['value']/1e9, j['ms_per_step'], j['roofline']['kernel_ms']))"
