"""Generates tests/golden/*.npz: inputs + expected outputs of the hot path, computed with the CPU oracle.

PARITY UNPINNED: the reference ships no fixtures and cannot be built or imported in this image, so these
vectors come from this repo's own restatement (oracle/), cross-checked by tests/test_oracle_model.py against
an independent 40-digit model.  They pin the oracle against regressions and let the GPU tests run against
committed data.  Usage: python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from lifcal_amd import _capi as capi, scene  # noqa: E402

S = scene.SceneSpec
CASES = {
    "cfg506": S(6, 40, None, 0x506, 9001),
    "cfgF06_robust_adj": S(6, 40, None, 0xF06, 9002, outlier_fraction=0.05),
    "cfg006_camera_only": S(5, 40, None, 0x006, 9003),
    "cfg306_poses_only": S(5, 40, None, 0x306, 9004),
    "cfg506_constraints": S(6, 40, None, 0x506, 9005, n_constraints=3),
    "cfgF06_windowed": S(20, 90, 6, 0xF06, 9006, outlier_fraction=0.02),
    "cfgF06_recalib": S(8, 50, None, 0xF06, 9007, recalib=True, outlier_fraction=0.02),
    "cfgD01": S(5, 30, None, 0xD01, 9008),
}
RADIUS = 1e4


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for name, spec in CASES.items():
        sc = scene.make_scene(spec)
        pa = capi.ProblemArrays.from_scene(sc)
        sw = oracle.sweep(pa, radius=RADIUS)
        n_s = min(32, sc.n_obs)
        rs, Js = [], []
        arity = 3 if (sc.config & 0x500) == 0x500 else (2 if sc.config & 0x100 else 1)
        for i in range(n_s):
            f, p = sc.fr[i], sc.pt[i]
            r, J = oracle.residual_block(sc.config, arity, pa.cam, pa.views[6 * f:6 * f + 6], pa.pts[3 * p:3 * p + 3],
                                         sc.u[i], sc.v[i], sc.mcx[i], sc.mcy[i], sc.spx, sc.scale)
            rs.append(r); Js.append(J)
        pb = capi.ProblemArrays.from_scene(sc)
        summ = oracle.solve(pb, threads=1)
        st = oracle.reproj_stats(pb)
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"),
            u=sc.u, v=sc.v, mcx=sc.mcx, mcy=sc.mcy, pt=sc.pt, fr=sc.fr, cam=sc.cam0, views=sc.views0, pts=sc.pts0,
            spx=sc.spx, scale=sc.scale, config=sc.config, fixed_mask=sc.fixed_mask,
            lower=sc.lower if sc.lower is not None else np.zeros(0), upper=sc.upper if sc.upper is not None else np.zeros(0),
            c_i=sc.c_i, c_j=sc.c_j, c_dist=sc.c_dist, c_sigma=sc.c_sigma, use_constraints=sc.use_constraints,
            radius=RADIUS, arity=arity,
            cost=sw.cost, gradient_max_norm=sw.gradient_max_norm, S=sw.S, rhs=sw.rhs, gradient_reduced=sw.gradient_reduced,
            point_gradient=sw.point_gradient, point_hessian_inv=sw.point_hessian_inv,
            sample_r=np.array(rs), sample_J=np.array(Js),
            solve_cam=pb.cam, solve_views=pb.views, solve_pts=pb.pts, solve_initial_cost=summ.initial_cost, solve_final_cost=summ.final_cost,
            solve_iterations=summ.iterations, solve_termination=summ.termination,
            stats=np.array([st.std_x, st.std_y, st.mae_x, st.mae_y, st.num_points, st.num_inliers]),
        )
        print(name, "N", sc.n_obs, "n_red", sw.n_reduced, "cost", sw.cost, "->", summ.final_cost, "it", summ.iterations)


if __name__ == "__main__":
    main()
