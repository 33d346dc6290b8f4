#!/bin/bash
# round 2, call N: multi-process rehearsal of bench.py on ONE GPU (gloo through the hooks): weak scaling N=2,4 and strong scaling of cfg4 at N=2
set -o pipefail
mkdir -p gpurun_out/r02
for n in 2 4; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 10 --warmup 2 --comm gloo --no-cpu-baseline > gpurun_out/r02/benchN_w$n.json 2> gpurun_out/r02/benchN_w$n.err; rc=$?
  echo "weak N=$n rc=$rc"; tail -2 gpurun_out/r02/benchN_w$n.err | cut -c1-300
  python - <<PY
import json
try:
    j = json.load(open("gpurun_out/r02/benchN_w$n.json"))
    print("   value %.3e obs/s  n_gpus %d scaling %s ms_per_step %.4f obs_per_gpu %d comm %s" % (j["value"], j["n_gpus"], j["scaling"], j["ms_per_step"], j["config"]["obs_per_gpu"], j["config"]["comm"]))
except Exception as e:
    print("   no line", e)
PY
  [ $rc -eq 0 ] || exit $rc
done
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29510 bench.py --gpus 2 --steps 5 --warmup 1 --comm gloo --no-cpu-baseline --scaling strong --workload cfg4 > gpurun_out/r02/benchN_strong2.json 2> gpurun_out/r02/benchN_strong2.err; rc=$?
echo "strong cfg4 N=2 rc=$rc"; tail -2 gpurun_out/r02/benchN_strong2.err | cut -c1-300
python - <<PY
import json
try:
    j = json.load(open("gpurun_out/r02/benchN_strong2.json"))
    print("   value %.3e obs/s  n_gpus %d scaling %s ms_per_step %.4f obs_per_gpu %d workload %s" % (j["value"], j["n_gpus"], j["scaling"], j["ms_per_step"], j["config"]["obs_per_gpu"], j["config"]["workload"][:60]))
except Exception as e:
    print("   no line", e)
PY
exit $rc
