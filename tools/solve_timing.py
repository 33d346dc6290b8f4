"""Times full LM solves on the GPU (and the oracle where it is fast enough) for the BASELINE configs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lifcal_amd import BundleAdjustment, _capi as capi, scene
import oracle
for name in sys.argv[1:] or ["cfg2", "cfg3", "metric"]:
    sc = scene.make_scene(scene.baseline_spec(name))
    pa = capi.ProblemArrays.from_scene(sc)
    t = time.time(); ba = BundleAdjustment(pa); tc = time.time() - t
    t = time.time(); s = ba.performBundleAdjustment(); tg = time.time() - t
    st = ba.calcReprojectionError()
    print(f"{name}: N={sc.n_obs} create {tc:.3f}s solve {1e3*tg:.2f} ms it {s.iterations} ({s.successful_steps}+{s.unsuccessful_steps}) cost {s.initial_cost:.6e}->{s.final_cost:.6e} term {s.termination} "
          f"sweep {1e3*s.seconds_sweep:.3f} ms linear+candidate {1e3*s.seconds_linear_solve:.3f} ms ({1e3*s.seconds_linear_solve/max(1,s.iterations-1):.3f} per iteration) rms ({st.std_x:.4f},{st.std_y:.4f}) inliers {st.num_inliers}/{st.num_points}")
    if name in ("cfg2", "cfg3"):
        pb = capi.ProblemArrays.from_scene(sc)
        t = time.time(); so = oracle.solve(pb, threads=oracle.hardware_threads()); to = time.time() - t
        print(f"   oracle: {to:.2f}s it {so.iterations} cost ->{so.final_cost:.6e}; cam rel diff {np.abs(pa.cam[:9]-pb.cam[:9]).max()/np.abs(pb.cam[:9]).max():.2e}")
    ba.close()
