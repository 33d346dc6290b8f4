"""Phases of lifcal_ba_create at the metric point (LIFCAL_PLAN_TIMING=1 prints them from the library); repeated to see the warm numbers.
gpurun -- tools/gpurun.sh run tools/create_timing.py [workload]"""
import os, sys, time
os.environ["LIFCAL_PLAN_TIMING"] = "1"
sys.path.insert(0, ".")
from lifcal_amd import BundleAdjustment, _capi as capi, scene   # noqa: E402
w = sys.argv[1] if len(sys.argv) > 1 else "metric"
sc = scene.make_scene(scene.baseline_spec(w))
for rep in range(4):
    pa = capi.ProblemArrays.from_scene(sc)
    t = time.perf_counter(); ba = BundleAdjustment(pa); t = time.perf_counter() - t
    print(f"create #{rep}: {1e3 * t:.2f} ms", flush=True)
    ba.close()
