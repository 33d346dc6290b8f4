"""Parity tests proper: the HIP path through the C ABI against the CPU oracle and the golden fixtures.

Bar (BASELINE.json north_star: fp64, "within 1e-6 relative of the reference on identical input"):
  * one sweep: cost <= 1e-12 relative; S / rhs / gradients <= 1e-9 block-scaled (summation order differs:
    the GPU accumulates with f64 atomics, so results are not bitwise reproducible run to run);
  * full solve: same iteration count and termination as the oracle, final cost <= 1e-8 relative, camera
    parameters <= 1e-6 relative, RMS reprojection error <= 1e-8 absolute.
"""
import ctypes as C

import numpy as np
import pytest

import oracle
from lifcal_amd import BundleAdjustment, LifcalError, _capi as capi, scene
from tests.helpers import GOLDEN, SMALL_CASES, S, load_golden, problem, scaled_max_err, vec_err

pytestmark = pytest.mark.gpu


def check_sweep(got, ref, tol_s=1e-9):
    assert got.n_reduced == ref.n_reduced and got.n_promoted == ref.n_promoted
    assert abs(got.cost - ref.cost) <= 1e-12 * abs(ref.cost)
    assert abs(got.gradient_max_norm - ref.gradient_max_norm) <= 1e-10 * ref.gradient_max_norm
    assert scaled_max_err(got.S, ref.S) < tol_s
    assert np.array_equal(got.S, got.S.T)
    assert vec_err(got.rhs, ref.rhs) < tol_s
    assert vec_err(got.gradient_reduced, ref.gradient_reduced) < 1e-10
    if np.abs(ref.point_gradient).max() > 0:
        assert vec_err(got.point_gradient, ref.point_gradient) < 1e-10
        assert vec_err(got.point_hessian_inv, ref.point_hessian_inv) < tol_s
    else:
        assert np.all(got.point_gradient == 0) and np.all(got.point_hessian_inv == 0)


@pytest.mark.parametrize("name,spec", SMALL_CASES, ids=[c[0] for c in SMALL_CASES])
def test_sweep_matches_oracle(built, name, spec):
    sc = scene.make_scene(spec)
    for radius in (1e4, 7.0):
        ref = oracle.sweep(problem(sc), radius=radius, threads=4)
        with BundleAdjustment(problem(sc)) as ba:
            check_sweep(ba.sweep(radius, want_matrices=True), ref)


@pytest.mark.parametrize("name,spec", SMALL_CASES, ids=[c[0] for c in SMALL_CASES])
def test_solve_matches_oracle(built, name, spec):
    sc = scene.make_scene(spec)
    pa, pb = problem(sc), problem(sc)
    with BundleAdjustment(pa) as ba:
        s = ba.performBundleAdjustment()
        st = ba.calcReprojectionError()
    so = oracle.solve(pb, threads=4)
    assert (s.iterations, s.termination) == (so.iterations, so.termination)
    assert (s.successful_steps, s.unsuccessful_steps) == (so.successful_steps, so.unsuccessful_steps)
    assert abs(s.initial_cost - so.initial_cost) <= 1e-12 * so.initial_cost
    assert abs(s.final_cost - so.final_cost) <= 1e-8 * so.final_cost
    live = 5 + (spec.config & 3) + (2 if spec.config & 4 else 0)
    assert np.allclose(pa.cam[:5], pb.cam[:5], rtol=1e-6, atol=0)
    # distortion coefficients: 1e-6 of the vector's scale (single components can sit arbitrarily close to zero)
    assert np.allclose(pa.cam[5:live], pb.cam[5:live], rtol=1e-6, atol=1e-6 * np.abs(pb.cam[5:live]).max() if live > 5 else 0.0)
    assert np.all(pa.cam[live:] == 0.0)
    assert np.allclose(pa.views, pb.views, rtol=0, atol=1e-6 * (1 + np.abs(pb.views).max()))
    assert np.allclose(pa.pts, pb.pts, rtol=0, atol=1e-6 * (1 + np.abs(pb.pts).max()))
    so_st = oracle.reproj_stats(pa)   # a12 on the GPU's own result
    assert abs(st.std_x - so_st.std_x) < 1e-10 and abs(st.std_y - so_st.std_y) < 1e-10
    assert abs(st.mae_x - so_st.mae_x) < 1e-9 and abs(st.mae_y - so_st.mae_y) < 1e-9
    assert (st.num_points, st.num_inliers) == (so_st.num_points, so_st.num_inliers)


@pytest.mark.parametrize("name", GOLDEN)
def test_golden_fixtures(built, name):
    """committed vectors: no oracle call on this path"""
    g, pa = load_golden(name)
    with BundleAdjustment(pa) as ba:
        got = ba.sweep(float(g["radius"]), want_matrices=True)
        assert abs(got.cost - float(g["cost"])) <= 1e-12 * float(g["cost"])
        assert scaled_max_err(got.S, g["S"]) < 1e-9
        assert vec_err(got.rhs, g["rhs"]) < 1e-9
        assert vec_err(got.gradient_reduced, g["gradient_reduced"]) < 1e-10
        s = ba.performBundleAdjustment()
        st = ba.calcReprojectionError()
    assert s.iterations == int(g["solve_iterations"]) and s.termination == int(g["solve_termination"])
    assert abs(s.final_cost - float(g["solve_final_cost"])) <= 1e-8 * float(g["solve_final_cost"])
    assert np.allclose(pa.cam[:5], g["solve_cam"][:5], rtol=1e-6)
    assert np.allclose([st.std_x, st.std_y], g["stats"][:2], rtol=0, atol=1e-8)
    assert (st.num_points, st.num_inliers) == (int(g["stats"][4]), int(g["stats"][5]))


def test_negative_stored_parameters_fold_like_the_functor(built):
    """reference BundleAdjustment.h:123-133 folds signs; calcReprojectionError (:1028-1039) does not."""
    sc = scene.make_scene(S(5, 30, None, 0x506, 801))
    cam = sc.cam0.copy(); cam[2] = -cam[2]
    mk = lambda: capi.ProblemArrays(sc.u, sc.v, sc.mcx, sc.mcy, sc.pt, sc.fr, cam, sc.views0, sc.pts0, sc.spx, sc.scale, sc.config)
    ref = oracle.sweep(mk(), radius=1e4)
    with BundleAdjustment(mk()) as ba:
        check_sweep(ba.sweep(1e4, want_matrices=True), ref)
        st = ba.calcReprojectionError()
    so = oracle.reproj_stats(mk())
    assert abs(st.std_x - so.std_x) <= 1e-9 * so.std_x and abs(st.std_y - so.std_y) <= 1e-9 * so.std_y


def test_ragged_groups_and_single_observation(built):
    """every 7th observation dropped -> ragged group sizes; then a one-observation problem"""
    sc = scene.make_scene(S(6, 40, None, 0xF06, 802))
    keep = np.arange(sc.n_obs) % 7 != 0
    mk = lambda m: capi.ProblemArrays(sc.u[m], sc.v[m], sc.mcx[m], sc.mcy[m], sc.pt[m], sc.fr[m], sc.cam0, sc.views0, sc.pts0, sc.spx, sc.scale, 0xF06)
    ref = oracle.sweep(mk(keep), radius=1e4)
    with BundleAdjustment(mk(keep)) as ba:
        check_sweep(ba.sweep(1e4, want_matrices=True), ref)
    one = np.zeros(sc.n_obs, bool); one[5] = True
    cam_only = lambda: capi.ProblemArrays(sc.u[one], sc.v[one], sc.mcx[one], sc.mcy[one], sc.pt[one], sc.fr[one], sc.cam0, sc.views0, sc.pts0, sc.spx, sc.scale, 0x006)
    ref1 = oracle.sweep(cam_only(), radius=1e4)
    with BundleAdjustment(cam_only()) as ba:
        check_sweep(ba.sweep(1e4, want_matrices=True), ref1)


def test_shuffled_input_order_gives_the_same_system(built):
    """the library re-sorts observations; the reference's frame-major order is not required"""
    sc = scene.make_scene(S(6, 40, None, 0x506, 803))
    perm = np.argsort(scene.Stream(5, 9).uniform(sc.n_obs))
    a = capi.ProblemArrays(sc.u, sc.v, sc.mcx, sc.mcy, sc.pt, sc.fr, sc.cam0, sc.views0, sc.pts0, sc.spx, sc.scale, sc.config)
    b = capi.ProblemArrays(sc.u[perm], sc.v[perm], sc.mcx[perm], sc.mcy[perm], sc.pt[perm], sc.fr[perm], sc.cam0, sc.views0, sc.pts0, sc.spx, sc.scale, sc.config)
    with BundleAdjustment(a) as ba, BundleAdjustment(b) as bb:
        ga, gb = ba.sweep(1e4, want_matrices=True), bb.sweep(1e4, want_matrices=True)
    assert abs(ga.cost - gb.cost) <= 1e-13 * ga.cost and scaled_max_err(ga.S, gb.S) < 1e-11


def test_options_and_reupload(built):
    sc = scene.make_scene(S(6, 40, None, 0x506, 804))
    pa = problem(sc)
    o = capi.default_options_py(); o.max_iterations = 3
    with BundleAdjustment(pa, o) as ba:
        s = ba.performBundleAdjustment()
        assert s.iterations == 3 and s.termination == 4
        cam_after = pa.cam.copy()
        pa.cam[:] = sc.cam0; pa.views[:] = sc.views0; pa.pts[:] = sc.pts0
        ba.upload_parameters()
        s2 = ba.performBundleAdjustment()
        assert s2.iterations == 3 and np.allclose(pa.cam, cam_after, rtol=1e-9)
    po = problem(sc); so = oracle.solve(po, o)
    assert np.allclose(pa.cam[:5], po.cam[:5], rtol=1e-7)
    o2 = capi.default_options_py(); o2.jacobi_scaling = 0
    ref = oracle.sweep(problem(sc), radius=50.0, options=o2)
    with BundleAdjustment(problem(sc), o2) as ba:
        check_sweep(ba.sweep(50.0, want_matrices=True), ref)


def test_invalid_arguments_fail_loudly(built):
    sc = scene.make_scene(S(4, 12, None, 0x506, 805))
    bad = problem(sc); bad.fr = bad.fr.copy(); bad.fr[0] = 999; bad.struct.fr = capi.as_uptr(bad.fr)
    with pytest.raises(LifcalError, match="range"):
        BundleAdjustment(bad)
    o = capi.default_options_py(); o.world_size = 2; o.rank = 0
    with BundleAdjustment(problem(sc), o) as ba:          # two ranks but no collective installed
        with pytest.raises(LifcalError, match="collective"):
            ba.sweep(1e4)
    with BundleAdjustment(problem(sc)) as ba:
        with pytest.raises(LifcalError):
            ba.sweep(-1.0)


@pytest.mark.parametrize("seed", [1311, 1312])
def test_active_bounds_trigger_the_line_search(built, seed):
    """camera-only problem with box bounds that cut off the minimiser: the projected LM step fails ceres' Armijo test,
    so TrustRegionMinimizer::DoLineSearch backtracks (cubic interpolation).  The GPU path must follow the oracle's
    trajectory through those searches (the oracle was checked to backtrack on these seeds: LO_DEBUG_LS=1)."""
    cfg = 0x006
    sc = scene.make_scene(S(6, 60, None, cfg, seed))
    lower = np.full(17, -np.inf); upper = np.full(17, np.inf)
    if sc.cam0[3] < sc.cam_gt[3]: upper[3] = sc.cam0[3] + 0.3
    else: lower[3] = sc.cam0[3] - 0.3
    gap = 0.3 * abs(sc.cam_gt[1] - sc.cam0[1])
    if sc.cam0[1] < sc.cam_gt[1]: upper[1] = sc.cam0[1] + gap
    else: lower[1] = sc.cam0[1] - gap
    mk = lambda: capi.ProblemArrays(sc.u, sc.v, sc.mcx, sc.mcy, sc.pt, sc.fr, sc.cam0, sc.views_gt, sc.pts_gt, sc.spx, sc.scale, cfg, lower=lower, upper=upper)
    pa, pb = mk(), mk()
    with BundleAdjustment(pa) as ba:
        s = ba.performBundleAdjustment()
    so = oracle.solve(pb, threads=4)
    assert (s.iterations, s.successful_steps, s.unsuccessful_steps, s.termination) == (so.iterations, so.successful_steps, so.unsuccessful_steps, so.termination)
    assert abs(s.final_cost - so.final_cost) <= 1e-8 * so.final_cost
    assert np.allclose(pa.cam[:5], pb.cam[:5], rtol=1e-6)
    assert np.all(pa.cam >= lower - 1e-12) and np.all(pa.cam <= upper + 1e-12)


def test_bounded_pose_point_problem_matches_oracle(built):
    """poses and points free, intrinsics boxed (the recalib pattern with a tight box): trial points go through k_sweep2"""
    sc = scene.make_scene(S(8, 60, None, 0xF06, 1320, outlier_fraction=0.02))
    lower = np.full(17, -np.inf); upper = np.full(17, np.inf)
    for k in (1, 3, 4):
        lower[k] = sc.cam0[k] - 0.05 * abs(sc.cam_gt[k] - sc.cam0[k]) - 1e-3
        upper[k] = sc.cam0[k] + 0.05 * abs(sc.cam_gt[k] - sc.cam0[k]) + 1e-3
    mk = lambda: capi.ProblemArrays(sc.u, sc.v, sc.mcx, sc.mcy, sc.pt, sc.fr, sc.cam0, sc.views0, sc.pts0, sc.spx, sc.scale, sc.config,
                                    fixed_mask=0b101, lower=lower, upper=upper)
    pa, pb = mk(), mk()
    with BundleAdjustment(pa) as ba:
        s = ba.performBundleAdjustment()
    so = oracle.solve(pb, threads=4)
    assert (s.iterations, s.termination) == (so.iterations, so.termination)
    assert abs(s.final_cost - so.final_cost) <= 1e-8 * so.final_cost
    assert pa.cam[0] == sc.cam0[0] and pa.cam[2] == sc.cam0[2]
    assert np.allclose(pa.cam[:5], pb.cam[:5], rtol=1e-6)
