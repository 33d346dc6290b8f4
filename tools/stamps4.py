"""Diagnostic: per-phase cycle shares of k_front4's three roles from a LIFCAL_STAMPS build (never used for timing claims).
Usage: LIFCAL_BA_LIB=lifcal_amd/csrc/liblifcal_ba_devstamps.so python tools/stamps4.py [workload]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lifcal_amd import _capi as capi
lib = capi.load_library(capi.LIB_PATH)
from lifcal_amd import BundleAdjustment, scene
name = sys.argv[1] if len(sys.argv) > 1 else "metric_web"
spec = scene.baseline_spec(name)
selector = None
if name.endswith("_web"):
    from lifcal_amd.mla import MicroLensGrid
    grid = MicroLensGrid(spec.raw_width, spec.raw_height, spec.lens_diameter, spec.lens_base_y, spec.grid_rotation, spec.grid_offset, True, device=0)

    def selector(img_x, img_y, img_vd, img_fr, img_pt, scale):
        o = grid.projectPointsToRawImage(img_x, img_y, img_vd, int(scale), fr=img_fr, pt=img_pt)
        return o.src, o.mcx, o.mcy
sc = scene.make_scene(spec, lens_selector=selector)
ba = BundleAdjustment(capi.ProblemArrays.from_scene(sc))
ba.sweep(1e4); r = ba.sweep(1e4)
nmax = 4096
buf = np.zeros(nmax * 64, np.uint64)
lib.lifcal_ba_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
n = lib.lifcal_ba_debug_stamps(ba._h, buf.ctypes.data, nmax * 2)
st = buf[: n * 64].reshape(n, 4, 16).astype(np.float64)
st[:, :, 12:15] = 0.0
names = {
    0: ["tile top (wait xyz)", "step: wait lens row", "step: read row, wait hand-off", "step: eval + stores", "", "", "", "", "", "", "", "tail"],
    1: ["", "step: wait for E", "step: read + FMA", "", "", "", "", "", "", "", "", "tail"],
    2: ["tile prep (loads, xyz)", "step: wait lens slot", "step: write row, next loads", "step: wait for E", "step: read + FMA", "emission (geometry, AG)", "wait staging free", "staging + gather", "frame-level sums (LDS)", "", "", "tail"],
}
names[3] = names[2]
print(f"{name}: front workgroups {n}, sweep {r.seconds*1e6:.1f} us")
st[:, :, 12:15] = 0.0
for role, tag in ((0, "E"), (1, "Y"), (2, "M0"), (3, "M1")):
    tot = st[:, role, :].sum(1)
    print(f" role {tag}: cycles per workgroup mean {tot.mean():.0f} max {tot.max():.0f} min {tot.min():.0f}")
    for i, nm in enumerate(names[role]):
        if nm:
            print(f"   {nm:26s} mean {st[:, role, i].mean():10.0f}  ({100*st[:, role, i].sum()/tot.sum():5.1f} %)  max {st[:, role, i].max():10.0f}")
