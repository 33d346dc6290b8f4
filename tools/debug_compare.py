"""Developer aid: compare one GPU sweep with the oracle block by block (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lifcal_amd import BundleAdjustment, _capi as capi, scene
import oracle

def rel(a, b, s):
    return float(np.max(np.abs(a - b) / s)) if a.size else 0.0

def compare(name, spec, radius=1e4, solve=True):
    sc = scene.make_scene(spec)
    pa = capi.ProblemArrays.from_scene(sc)
    ref = oracle.sweep(capi.ProblemArrays.from_scene(sc), radius=radius, threads=8)
    ba = BundleAdjustment(pa)
    got = ba.sweep(radius, want_matrices=True)
    F = spec.n_frames
    d = np.sqrt(np.abs(np.diag(ref.S))) + 1e-300
    sc2 = np.outer(d, d)
    E = np.abs(got.S - ref.S) / sc2
    print(f"== {name}: N={sc.n_obs} cfg={spec.config:#x} n_red={ref.n_reduced} prom={ref.n_promoted}")
    print(f"   cost gpu {got.cost:.12e} ref {ref.cost:.12e} rel {abs(got.cost-ref.cost)/abs(ref.cost):.2e}; gmax {got.gradient_max_norm:.6e} / {ref.gradient_max_norm:.6e}")
    print(f"   S: cam-cam {E[:17,:17].max():.2e} cam-pose {E[:17,17:17+6*F].max():.2e} pose-pose {E[17:17+6*F,17:17+6*F].max():.2e} prom {E[17+6*F:,:].max() if ref.n_promoted else 0:.2e}")
    gs = np.abs(ref.gradient_reduced).max() + 1e-300
    print(f"   gradB {np.abs(got.gradient_reduced-ref.gradient_reduced).max()/gs:.2e} rhs {np.abs(got.rhs-ref.rhs).max()/(np.abs(ref.rhs).max()+1e-300):.2e} "
          f"gP {np.abs(got.point_gradient-ref.point_gradient).max()/(np.abs(ref.point_gradient).max()+1e-300):.2e} "
          f"Uinv {np.abs(got.point_hessian_inv-ref.point_hessian_inv).max()/(np.abs(ref.point_hessian_inv).max()+1e-300):.2e}")
    # solution of the reduced system
    try:
        x_ref = np.linalg.solve(ref.S, ref.rhs); x_got = np.linalg.solve(got.S, got.rhs)
        print(f"   delta_reduced rel diff {np.abs(x_ref-x_got).max()/(np.abs(x_ref).max()+1e-300):.2e}")
    except Exception as e:
        print("   solve failed", e)
    if solve:
        t = time.time(); s = ba.performBundleAdjustment(); tg = time.time() - t
        pb = capi.ProblemArrays.from_scene(sc)
        t = time.time(); so = oracle.solve(pb, threads=8); to = time.time() - t
        print(f"   solve gpu: it {s.iterations} cost {s.initial_cost:.8e}->{s.final_cost:.10e} term {s.termination} ({tg:.2f}s) | oracle: it {so.iterations} ->{so.final_cost:.10e} term {so.termination} ({to:.2f}s)")
        print(f"   cam gpu {pa.cam[:9]}\n   cam ref {pb.cam[:9]}")
        st = ba.calcReprojectionError(); so2 = oracle.reproj_stats(pa)
        print(f"   stats gpu {st.std_x:.6f} {st.std_y:.6f} {st.mae_x:.5f} {st.mae_y:.5f} {st.num_inliers}/{st.num_points} | oracle(on gpu params) {so2.std_x:.6f} {so2.std_y:.6f} {so2.mae_x:.5f} {so2.mae_y:.5f} {so2.num_inliers}/{so2.num_points}")
    ba.close()

if __name__ == "__main__":
    S = scene.SceneSpec
    which = sys.argv[1:] or ["tiny", "tiny_f06", "tiny_cam", "tiny_pose", "tiny_con", "cfg1", "cfg2"]
    for w in which:
        if w == "tiny": compare(w, scene.baseline_spec("tiny"))
        elif w == "tiny_f06": compare(w, S(6, 40, None, 0xF06, 11, outlier_fraction=0.05))
        elif w == "tiny_cam": compare(w, S(6, 40, None, 0x006, 12))
        elif w == "tiny_pose": compare(w, S(6, 40, None, 0x305, 13))
        elif w == "tiny_con": compare(w, S(6, 40, None, 0x506, 14, n_constraints=3))
        elif w == "win": compare(w, S(30, 300, 8, 0xF06, 15, outlier_fraction=0.02))
        else: compare(w, scene.baseline_spec(w), solve=(w not in ("metric", "cfg4", "cfg5")))
