"""Diagnostic: per-phase cycle shares of the LDS-window sweep kernel from the LIFCAL_STAMPS build (never used for timing
claims): thread 0 (evaluator wave 0) and, for k_sweep3, thread 256 (accumulator wave 0, rows "acc: ...")."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lifcal_amd import _capi as capi
capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "liblifcal_ba_stamps.so")
capi._lib = None
lib = capi.load_library(capi.LIB_PATH)
from lifcal_amd import BundleAdjustment, scene
name = sys.argv[1] if len(sys.argv) > 1 else "metric"
sc = scene.make_scene(scene.baseline_spec(name))
ba = BundleAdjustment(capi.ProblemArrays.from_scene(sc))
ba.sweep(1e4); r = ba.sweep(1e4)
info = ba.info()
buf = np.zeros(info.n_chunks * 32, np.uint64)
lib.lifcal_ba_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
n = lib.lifcal_ba_debug_stamps(ba._h, buf.ctypes.data, info.n_chunks)
st = buf.reshape(-1, 32)[:n].astype(np.float64)
names = ["zero+sync", "phase1 wait@sync", "phase2 factor", "phase3 W->HBM,Z", "phase4 schur", "flush to HBM", "phase1 emission(w0)", "phase1 obs loop(w0)",
         "(slack)", "tile emit", "cc/gc/cost reduce", "replica fold", "pass top (prefetch issue)", "zero stores", "prologue", "",
         "acc: pass top", "acc: obs loop", "acc: P2a+zero+P2", "acc: emission", "acc: wait P4", "acc: Z + Schur half", "", "", "", "", "", "", "", "", "", ""]
tot = st[:, :16].sum(1)
print(f"{name}: blocks {n}, sweep {r.seconds*1e6:.1f} us; cycles per block: mean {tot.mean():.0f} max {tot.max():.0f} min {tot.min():.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:18s} mean {st[:, i].mean():10.0f}  ({100*st[:, i].sum()/tot.sum():5.1f} %)  max {st[:, i].max():10.0f}")
