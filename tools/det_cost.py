"""What options.deterministic = 1 costs, and whether the scenes of tests/test_gpu_deterministic.py can tell an ordered sum from an
unordered one: per scene the time of a sweep in both modes and the number of differing bits patterns over repeated ATOMIC sweeps
(0 differing repetitions would mean the scene is too small for the reproducibility test to mean anything).
Run on the GPU box: gpurun -- tools/gpurun.sh run tools/det_cost.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from lifcal_amd import BundleAdjustment, _capi as capi, scene   # noqa: E402
from tests.helpers import S, problem                            # noqa: E402
from tests.test_gpu_deterministic import CASES, SPECIAL_CASES   # noqa: E402

extra = [("poses_only_cfg3_size", S(200, 20000, 10, 0x306, 3116))]
for name, spec in CASES + SPECIAL_CASES + extra:
    sc = scene.make_scene(spec)
    row = [name, f"obs {len(sc.u)}"]
    for det in (0, 1):
        o = capi.default_options_py(); o.deterministic = det
        with BundleAdjustment(problem(sc), o) as ba:
            outs = []
            for rep in range(6):
                g = ba.sweep(1e3, want_matrices=True)
                outs.append((g.S.copy(), g.rhs.copy()))
            differing = sum(1 for a in outs[1:] if not (np.array_equal(a[0], outs[0][0]) and np.array_equal(a[1], outs[0][1])))
            ba.profile_begin(20, 1)
            for rep in range(20):
                ba.sweep_enqueue(1e3)
            pr = ba.profile_end()   # (synchronises)
            t = time.perf_counter()
            for rep in range(20):
                ba.sweep_enqueue(1e3)
            ba.sweep(1e3)
            t = (time.perf_counter() - t) / 21
        row.append(f"{'ordered' if det else 'atomic'}: {t * 1e3:.3f} ms/sweep, {differing}/5 repetitions differ")
    row.append(f"special points {pr.special_points:.0f}")
    print(" | ".join(row), flush=True)
