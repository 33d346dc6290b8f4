"""Diagnostic: per-phase cycles of the band Cholesky chain (LIFCAL_STAMPS build)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lifcal_amd import _capi as capi
capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "liblifcal_ba_stamps.so")
capi._lib = None
lib = capi.load_library(capi.LIB_PATH)
from lifcal_amd import BundleAdjustment, scene
sc = scene.make_scene(scene.baseline_spec(sys.argv[1] if len(sys.argv) > 1 else "metric"))
ba = BundleAdjustment(capi.ProblemArrays.from_scene(sc))
o = capi.default_options_py()
s = ba.performBundleAdjustment()
buf = np.zeros(32, np.uint64)
lib.lifcal_ba_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
lib.lifcal_ba_debug_stamps(ba._h, buf.ctypes.data, 1)
F = sc.spec.n_frames
names = ["factor (lane 0)", "barrier after factor", "panel rows", "update (+prefetch issue)", "barrier after update", "slide + barrier"]
print("solve", s.iterations, s.seconds_linear_solve)
for n, v in zip(names, buf[:6]):
    print(f"  {n:26s} {float(v)/F:9.0f} cycles per frame")
print("  total per frame", float(buf[:6].sum()) / F)
