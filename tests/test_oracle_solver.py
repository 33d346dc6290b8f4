"""Behaviour of the oracle's Ceres-style trust-region loop."""
import numpy as np
import pytest

import oracle
from lifcal_amd import _capi as capi, scene
from tests.helpers import S, problem


def test_noise_free_scene_recovers_ground_truth_cost():
    """noise-free synthetic scene: the minimum is the generating point, cost -> ~0 (gauge-invariant check)."""
    sc = scene.make_scene(S(6, 40, None, 0x506, 501, noise_px=0.0))
    pa = problem(sc)
    o = capi.default_options_py()
    o.function_tolerance = 1e-14; o.parameter_tolerance = 1e-14
    s = oracle.solve(pa, o, threads=4)
    assert s.initial_cost > 1e3
    assert s.final_cost < 1e-12 * s.initial_cost
    st = oracle.reproj_stats(pa)
    assert st.std_x < 1e-5 and st.std_y < 1e-5 and st.num_inliers == st.num_points


def test_intrinsics_only_recovers_camera():
    """camera-only arity with exact poses/points: every intrinsic is observable -> recovered to 1e-6 relative."""
    sc = scene.make_scene(S(6, 60, None, 0x006, 502, noise_px=0.0))
    pa = capi.ProblemArrays(sc.u, sc.v, sc.mcx, sc.mcy, sc.pt, sc.fr, sc.cam0, sc.views_gt, sc.pts_gt, sc.spx, sc.scale, sc.config)
    o = capi.default_options_py(); o.function_tolerance = 1e-16; o.parameter_tolerance = 1e-16
    s = oracle.solve(pa, o, threads=4)
    assert s.final_cost < 1e-14 * s.initial_cost
    assert np.allclose(pa.cam[:5], sc.cam_gt[:5], rtol=1e-7)
    assert np.allclose(pa.cam[5:9], sc.cam_gt[5:9], rtol=1e-4, atol=1e-12)
    assert np.all(pa.cam[9:] == 0.0)


def test_default_run_terminates_on_function_tolerance_and_reduces_cost():
    sc = scene.make_scene(S(8, 60, None, 0xF06, 503, outlier_fraction=0.03))
    pa = problem(sc)
    s = oracle.solve(pa, threads=4)
    assert s.termination in (capi_term("FUNCTION"), capi_term("PARAMETER"))
    assert s.final_cost < 0.1 * s.initial_cost   # Cauchy loss saturates, so the initial cost is modest
    assert 0 < s.iterations <= 200
    assert s.successful_steps + s.unsuccessful_steps == s.iterations or s.termination != 0


def capi_term(name):
    return {"FUNCTION": 1, "PARAMETER": 2, "GRADIENT": 3, "MAXIT": 4}[name]


def test_recalib_keeps_fixed_parameters_and_bounds():
    """reference :927-953: fL and B constant (SubsetManifold), bL0/cx/cy inside [0.7, 1.3] x init."""
    sc = scene.make_scene(S(8, 60, None, 0xF06, 504, recalib=True))
    pa = problem(sc)
    cam0 = pa.cam.copy()
    s = oracle.solve(pa, threads=4)
    assert pa.cam[0] == cam0[0] and pa.cam[2] == cam0[2]
    for k in (1, 3, 4):
        assert sc.lower[k] <= pa.cam[k] <= sc.upper[k]
    assert s.final_cost < s.initial_cost


def test_active_bound_is_respected():
    """a bound that cuts off the minimiser: the projected step (ParameterBlock::Plus) keeps cx on the box."""
    sc = scene.make_scene(S(6, 40, None, 0x506, 505))
    lower = np.full(17, -np.inf); upper = np.full(17, np.inf)
    upper[3] = sc.cam0[3] - 2.0 if sc.cam0[3] < sc.cam_gt[3] else np.inf
    lower[3] = sc.cam0[3] + 2.0 if sc.cam0[3] > sc.cam_gt[3] else -np.inf
    pa = capi.ProblemArrays(sc.u, sc.v, sc.mcx, sc.mcy, sc.pt, sc.fr, sc.cam0, sc.views0, sc.pts0, sc.spx, sc.scale, sc.config,
                            lower=lower, upper=upper)
    s = oracle.solve(pa, threads=4)
    assert lower[3] <= pa.cam[3] <= upper[3]
    assert np.isfinite(s.final_cost) and s.final_cost < s.initial_cost


def test_max_iterations_and_gradient_tolerance():
    sc = scene.make_scene(S(6, 40, None, 0x506, 506))
    pa = problem(sc)
    o = capi.default_options_py(); o.max_iterations = 2
    s = oracle.solve(pa, o)
    assert s.iterations == 2 and s.termination == capi_term("MAXIT")
    pb = problem(sc)
    o2 = capi.default_options_py(); o2.gradient_tolerance = 1e300
    s2 = oracle.solve(pb, o2)
    assert s2.iterations == 0 and s2.termination == capi_term("GRADIENT")
    assert np.array_equal(pb.cam, sc.cam0)


def test_reproj_stats_definition():
    """reference :1083-1098: std = sqrt(sum e^2 / N) (not mean-removed), 'mae' = max |e|, inliers |e| <= thr."""
    sc = scene.make_scene(S(6, 40, None, 0xF06, 507, outlier_fraction=0.05))
    pa = problem(sc, initial=False)
    st, err = oracle.reproj_stats(pa, 1.0, want_errors=True)
    assert abs(st.std_x - np.sqrt(np.mean(err[:, 0] ** 2))) < 1e-14
    assert abs(st.std_y - np.sqrt(np.mean(err[:, 1] ** 2))) < 1e-14
    assert st.mae_x == np.abs(err[:, 0]).max() and st.mae_y == np.abs(err[:, 1]).max()
    assert st.num_inliers == int(np.sum(np.sum(err ** 2, 1) <= 1.0)) and st.num_points == sc.n_obs
    # at ground truth the functor residual equals the stats error (no negative parameters -> folding is a no-op)
    assert np.allclose(oracle.residuals(pa), err, rtol=0, atol=1e-9)
