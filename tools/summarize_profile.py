"""Condenses rocprofv3 output (kernel-trace --stats CSV + FETCH_SIZE / WRITE_SIZE PMC passes) into a short
text summary and a traffic.json fragment.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports
half the bytes of wide coalesced streaming reads, so it is doubled; WRITE_SIZE is taken as is; both are KiB."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]

def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))

print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", r.get("TotalDuration", 0)) or 0))
    print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s} {'pct':>6s}")
    for r in rows[:14]:
        name = r["Name"][:70]
        print(f"{name:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.2f} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['Percentage']):6.2f}")

def pmc(pattern, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in find(pattern):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"].split("(")[0]
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return {k: v[0] / max(v[1], 1) for k, v in acc.items()}

fetch = pmc("pmc_fetch/**/*counter_collection.csv", "FETCH_SIZE")
write = pmc("pmc_write/**/*counter_collection.csv", "WRITE_SIZE")
print("\n== HBM traffic per launch (KiB counters -> bytes; FETCH_SIZE x2 on gfx950) ==")
traffic = {}
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, 0) + write.get(k, 0))):
    fb = 2.0 * fetch.get(k, 0.0) * 1024.0
    wb = write.get(k, 0.0) * 1024.0
    short = k.split("<")[0].replace("void lifcal::", "")
    traffic[short] = traffic.get(short, 0.0) + 0.0
    print(f"{k[:80]:80s} fetch {fb/1e6:10.2f} MB  write {wb/1e6:10.2f} MB  total {(fb+wb)/1e6:10.2f} MB")
    traffic[short] = fb + wb
json.dump(traffic, open(os.path.join(out, "traffic_by_kernel.json"), "w"), indent=1)

# SQ issue counters of the sweep kernels (separate passes pmc_sq*), averaged per launch
print("\n== SQ counters per launch (sweep kernels; SQ_* cycle counters count quad-cycles) ==")
sq = defaultdict(dict)
for pat in ("pmc_sq1/**/*counter_collection.csv", "pmc_sq2/**/*counter_collection.csv", "pmc_sq3/**/*counter_collection.csv"):
    acc = defaultdict(lambda: [0.0, 0])
    for f in find(pat):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if not any(k in name for k in ("k_sweep3", "k_front4", "k_back4")):
                continue
            short = name.split("(")[0].replace("void lifcal::", "")
            acc[(short, r["Counter_Name"])][0] += float(r["Counter_Value"]); acc[(short, r["Counter_Name"])][1] += 1
    for (short, k), v in acc.items():
        sq[short][k] = v[0] / max(v[1], 1)
for short in sorted(sq):
    print(f" {short}")
    for k in sorted(sq[short]):
        print(f"  {k:24s} {sq[short][k]:16.0f}")
json.dump(sq, open(os.path.join(out, "sq_counters.json"), "w"), indent=1)
