"""Times the projectPointsToRawImage row (include/lifcal_mla.h) at the size of the metric scene: a 2048x2048 raw image,
the metric scene's virtual-image points (about 207 k of them, about 1 M micro-image observations), GPU library against the
CPU restatement on this host.  Prints one JSON line.  Run under rocprofv3 --kernel-trace --stats for the per-kernel times
(tools/profile_mla.sh)."""
import json
import sys
import time

import numpy as np

from lifcal_amd import MicroLensGrid, scene

name = sys.argv[1] if len(sys.argv) > 1 else "metric"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
with_cpu = "--no-cpu" not in sys.argv
sc = scene.make_scene(scene.baseline_spec(name))
sp = sc.spec
kw = dict(width=sp.raw_width, height=sp.raw_height, lens_diameter=sp.lens_diameter, lens_base_y=sp.lens_base_y, rotation=sp.grid_rotation, offset=sp.grid_offset)
t0 = time.perf_counter(); g = MicroLensGrid(**kw); t_create_first = time.perf_counter() - t0
g.close()
t0 = time.perf_counter(); g = MicroLensGrid(**kw); t_create = time.perf_counter() - t0
o = g.projectPointsToRawImage(sc.img_x, sc.img_y, sc.img_vd, sp.scale, fr=sc.img_fr, pt=sc.img_pt)
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    o = g.projectPointsToRawImage(sc.img_x, sc.img_y, sc.img_vd, sp.scale, fr=sc.img_fr, pt=sc.img_pt)
    ts.append(time.perf_counter() - t0)
out = {"scene": name, "image_points": int(len(sc.img_x)), "observations": int(len(o.u)), "n_lenses": g.n_lenses, "web_lines": g.n_web_lines,
       "gpu_create_s": t_create, "gpu_create_first_s": t_create_first, "gpu_project_s_median_host_to_host": float(np.median(ts))}
if with_cpu:
    from oracle.mla import MicroLensGrid as OracleGrid
    t0 = time.perf_counter(); og = OracleGrid(**kw); out["cpu_create_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    n = 0
    for f in range(sp.n_frames):
        m = sc.img_fr == f
        n += len(og.project_frame(sc.img_x[m], sc.img_y[m], sc.img_vd[m], sp.scale).xR)
    out["cpu_project_s"] = time.perf_counter() - t0
    out["cpu_observations"] = n
print(json.dumps(out))
