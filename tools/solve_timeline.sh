#!/bin/bash
# Timeline of ONE full LM solve (kernel trace, no counters): per-kernel totals, idle time between kernels, and the kernel sequence of
# one iteration -> gpurun_out/timeline_<tag>/timeline.txt.   gpurun -- tools/solve_timeline.sh <tag> [workload]
TAG=${1:-t}; W=${2:-metric}
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/solve_timing.py $W > $OUT/trace.log 2>&1
tail -3 $OUT/trace.log
k=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
m=$(find $OUT/trace -name '*memory_copy_trace.csv' | head -1)
python3 - "$k" "$m" > $OUT/timeline.txt <<'PY'
import csv, sys
from collections import defaultdict
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lifcal::", "")[:44]))
try:
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[:30]))
except Exception as e:  # noqa: BLE001
    print("no copy trace:", e)
ev.sort()
# the solve = from the first k_lm_control-less marker: take the span between the first and the last k_cr_backsub (+ the sweep in front)
idx = [i for i, e in enumerate(ev) if e[2].startswith("k_cr_backsub")]
if not idx:
    idx = [i for i, e in enumerate(ev) if e[2].startswith("k_band")]
a = idx[0]
while a > 0 and not ev[a][2].startswith("k_tables"): a -= 1
b = idx[-1]
while b + 1 < len(ev) and not ev[b][2].startswith("k_stats"): b += 1
span = ev[a:b]
t0, t1 = span[0][0], max(e[1] for e in span)
busy = 0; cur = t0; gaps = []
tot = defaultdict(lambda: [0, 0])
for s, e, n in span:
    tot[n][0] += e - s; tot[n][1] += 1
    if s > cur: gaps.append((s - cur, n)); cur_s = s
    else: cur_s = cur
    busy += max(0, e - max(s, cur)); cur = max(cur, e)
print(f"span {1e-6 * (t1 - t0):.3f} ms, busy {1e-6 * busy:.3f} ms, idle {1e-6 * (t1 - t0 - busy):.3f} ms in {len(gaps)} gaps")
for n, (ns, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"  {n:46s} calls {c:4d} avg {ns / c / 1e3:8.2f} us total {ns / 1e6:8.3f} ms")
g = defaultdict(lambda: [0, 0])
for d, n in gaps: g[n][0] += d; g[n][1] += 1
print("idle time in front of:")
for n, (ns, c) in sorted(g.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {n:46s} gaps {c:4d} avg {ns / c / 1e3:8.2f} us total {ns / 1e6:8.3f} ms")
# one iteration in the middle: between the 3rd and 4th k_update_reduced
u = [i for i, e in enumerate(span) if e[2].startswith("k_update_reduced")]
if len(u) >= 4:
    print("one iteration (start offset us, duration us, kernel):")
    base = span[u[2]][0]
    for s, e, n in span[u[2]:u[3]]:
        print(f"  {1e-3 * (s - base):9.2f} {1e-3 * (e - s):8.2f}  {n}")
PY
cat $OUT/timeline.txt | head -60
