"""Times the host planner (lifcal_ba_plan: no GPU needed) on a BASELINE scene; LIFCAL_PLAN_TIMING=1 prints its phases."""
import os
import pickle
import sys
import time

import lifcal_amd
from lifcal_amd import _capi as capi, scene

name = sys.argv[1] if len(sys.argv) > 1 else "metric"
cache = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", f"scene_{name}.pkl")
if os.path.exists(cache):
    sc = pickle.load(open(cache, "rb"))
else:
    sc = scene.make_scene(scene.baseline_spec(name))
    pickle.dump(sc, open(cache, "wb"))
pa = capi.ProblemArrays.from_scene(sc)
for _ in range(3):
    t = time.time()
    info, order, owner = lifcal_amd.plan(pa)
    print("plan seconds", round(time.time() - t, 4))
