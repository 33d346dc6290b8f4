"""Times lifcal_ba_create on the GPU box for a BASELINE scene (LIFCAL_PLAN_TIMING=1 prints the phases)."""
import sys
import time

from lifcal_amd import BundleAdjustment, _capi as capi, scene

name = sys.argv[1] if len(sys.argv) > 1 else "metric"
sc = scene.make_scene(scene.baseline_spec(name))
pa = capi.ProblemArrays.from_scene(sc)
for _ in range(3):
    t = time.perf_counter()
    ba = BundleAdjustment(pa)
    print("create seconds", round(time.perf_counter() - t, 4), flush=True)
    ba.close()
