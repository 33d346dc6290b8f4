"""options.deterministic = 1: the sweep without order-dependent sums — LDS accumulation in wave order, per-block window slabs
added up in block order by one owner per entry (k_det_reduce) instead of the cross-block f64 atomics, value kernels summed per
workgroup in order.  Done = repeated runs, and separate handles, agree BITWISE; against the oracle the usual tolerances hold
(its summation order is a different one)."""
import numpy as np
import pytest

import oracle
from lifcal_amd import BundleAdjustment, _capi as capi, scene
from tests.helpers import S, problem, scaled_max_err, vec_err

pytestmark = pytest.mark.gpu


def opts(det=1, precision=0):
    o = capi.default_options_py(); o.deterministic = det; o.precision = precision
    return o


CASES = [
    ("full_robust_adj", S(6, 40, None, 0xF06, 3101, outlier_fraction=0.05)),
    ("windowed", S(24, 120, 6, 0xF06, 3102, outlier_fraction=0.02)),
    ("windowed_many_blocks", S(60, 2500, 8, 0x506, 3103)),
]


@pytest.mark.parametrize("name,spec", CASES, ids=[c[0] for c in CASES])
def test_sweeps_are_bitwise_reproducible(built, name, spec):
    sc = scene.make_scene(spec)
    ref = oracle.sweep(problem(sc), radius=1e3, threads=4)
    runs = []
    for handle in range(2):
        with BundleAdjustment(problem(sc), opts()) as ba:
            for rep in range(3):
                g = ba.sweep(1e3, want_matrices=True)
                runs.append((g.cost, g.S.copy(), g.rhs.copy(), g.gradient_reduced.copy(), g.point_gradient.copy(), g.point_hessian_inv.copy(), g.gradient_max_norm))
    c0 = runs[0]
    for r in runs[1:]:
        assert r[0] == c0[0] and r[6] == c0[6]
        for a, b in zip(r[1:6], c0[1:6]):
            assert np.array_equal(a, b)
    assert abs(c0[0] - ref.cost) <= 1e-13 * ref.cost
    assert scaled_max_err(c0[1], ref.S) < 1e-9 and vec_err(c0[2], ref.rhs) < 1e-9
    assert vec_err(c0[4], ref.point_gradient) < 1e-10


@pytest.mark.parametrize("precision", [0, 1])
def test_solves_are_bitwise_reproducible(built, precision):
    sc = scene.make_scene(scene.baseline_spec("cfg2"))
    out = []
    for rep in range(3):
        pa = problem(sc)
        with BundleAdjustment(pa, opts(1, precision)) as ba:
            s = ba.performBundleAdjustment()
            st = ba.calcReprojectionError()
        out.append((pa.cam.copy(), pa.views.copy(), pa.pts.copy(), s.final_cost, s.iterations, s.final_gradient_max_norm, st.std_x, st.std_y, st.mae_x))
    for o in out[1:]:
        for a, b in zip(o, out[0]):
            assert np.array_equal(np.asarray(a), np.asarray(b))
    if precision == 0:
        pb = problem(sc)
        so = oracle.solve(pb, threads=oracle.hardware_threads())
        assert (out[0][4], abs(out[0][3] - so.final_cost) <= 1e-8 * so.final_cost) == (so.iterations, True)
        assert np.allclose(out[0][0][:9], pb.cam[:9], rtol=1e-6)


def test_ordered_sums_equal_atomic_sums_at_the_widest_window(built):
    """k_det_reduce finds the blocks covering a frame inside Plan::NF_MAX frames (a constant shared with the planner
    since round 3, static_assert in lifcal_ba.hip) — a block dropped from the ordered sum would still be bitwise reproducible, so
    the deterministic system is held against the ATOMIC one (and the oracle) on a scene whose frame windows reach NF_MAX = 20"""
    sc = scene.make_scene(S(70, 500, 19, 0xF06, 3120, outlier_fraction=0.02))
    ref = oracle.sweep(problem(sc), radius=1e3, threads=4)
    with BundleAdjustment(problem(sc), opts(0)) as ba:
        at = ba.sweep(1e3, want_matrices=True)
        assert ba.info().max_window_frames >= 19
    with BundleAdjustment(problem(sc), opts(1)) as ba:
        dt = ba.sweep(1e3, want_matrices=True)
    assert abs(dt.cost - at.cost) <= 1e-13 * at.cost
    assert scaled_max_err(dt.S, at.S) < 1e-11 and vec_err(dt.rhs, at.rhs) < 1e-11 and vec_err(dt.gradient_reduced, at.gradient_reduced) < 1e-11
    assert scaled_max_err(dt.S, ref.S) < 1e-9 and vec_err(dt.rhs, ref.rhs) < 1e-9


SPECIAL_CASES = [
    ("constraints", S(6, 40, None, 0x506, 3110, n_constraints=3)),
    ("constraints_adj_robust_windowed", S(24, 300, 6, 0xF06, 3113, n_constraints=12, outlier_fraction=0.03)),   # regular blocks AND special points
    ("long_tracks", S(30, 400, None, 0xF06, 3114, outlier_fraction=0.02)),   # frame span 30 > NF_MAX: every point on the global-atomic kernels, 190 tiles
    ("camera_only", S(6, 300, None, 0x006, 3111)),                           # arity <2,17>
    ("poses_only", S(12, 300, None, 0x306, 3115)),                           # arity <2,17,6>
]


@pytest.mark.parametrize("name,spec", SPECIAL_CASES, ids=[c[0] for c in SPECIAL_CASES])
def test_special_points_and_arities_are_bitwise_reproducible(built, name, spec):
    """round 3: the kernels that sum with GLOBAL atomics (k_sweep, k_schur, k_constraints: points with distance constraints, promoted
    points, tracks longer than the LDS window, the camera-only / pose-only arities) emit in a fixed order in deterministic mode
    (turn counters, kernels.hpp det_turn_wait): bitwise equal sweeps across repetitions and handles, the atomic mode's values up
    to summation order, the oracle's at the usual tolerances"""
    sc = scene.make_scene(spec)
    ref = oracle.sweep(problem(sc), radius=1e3, threads=4)
    runs = []
    for handle in range(2):
        with BundleAdjustment(problem(sc), opts()) as ba:
            for rep in range(3):
                g = ba.sweep(1e3, want_matrices=True)
                runs.append((g.cost, g.S.copy(), g.rhs.copy(), g.gradient_reduced.copy(), g.point_gradient.copy(), g.point_hessian_inv.copy(), g.gradient_max_norm))
    c0 = runs[0]
    for r in runs[1:]:
        assert r[0] == c0[0] and r[6] == c0[6]
        for a, b in zip(r[1:6], c0[1:6]):
            assert np.array_equal(a, b)
    with BundleAdjustment(problem(sc), opts(0)) as ba:
        at = ba.sweep(1e3, want_matrices=True)
    assert abs(c0[0] - at.cost) <= 1e-13 * at.cost
    assert scaled_max_err(c0[1], at.S) < 1e-11 and vec_err(c0[2], at.rhs) < 1e-11 and vec_err(c0[3], at.gradient_reduced) < 1e-11
    assert abs(c0[0] - ref.cost) <= 1e-13 * ref.cost
    assert scaled_max_err(c0[1], ref.S) < 1e-9 and vec_err(c0[2], ref.rhs) < 1e-9


@pytest.mark.parametrize("name", ["constraints_adj_robust_windowed", "long_tracks", "poses_only"])
def test_solves_with_special_points_are_bitwise_reproducible(built, name):
    sc = scene.make_scene(dict(SPECIAL_CASES)[name])
    out = []
    for rep in range(3):
        pa = problem(sc)
        with BundleAdjustment(pa, opts()) as ba:
            s = ba.performBundleAdjustment()
        out.append((pa.cam.copy(), pa.views.copy(), pa.pts.copy(), s.final_cost, s.iterations, s.successful_steps, s.final_gradient_max_norm))
    for o in out[1:]:
        for a, b in zip(o, out[0]):
            assert np.array_equal(np.asarray(a), np.asarray(b))
    pb = problem(sc)
    so = oracle.solve(pb, threads=oracle.hardware_threads())
    assert out[0][4] == so.iterations and abs(out[0][3] - so.final_cost) <= 1e-8 * so.final_cost


def test_bounded_problems_are_bitwise_reproducible_through_the_line_search(built):
    """recalibration pattern (BASELINE configs[4]: slots 0, 2 constant, box bounds) with a box so tight that the projected step fails
    ceres' Armijo test: the line search's sums (step norms, directional derivative, trial costs) are ordered too since round 3 —
    three solves agree bitwise and follow the oracle through the same accept / reject sequence"""
    from tests.helpers import bounded_problem
    sc = scene.make_scene(S(8, 60, None, 0xF06, 3112, recalib=True, outlier_fraction=0.02))
    out = []
    for rep in range(3):
        pa = bounded_problem(sc)
        with BundleAdjustment(pa, opts()) as ba:
            s = ba.performBundleAdjustment()
        out.append((pa.cam.copy(), pa.views.copy(), pa.pts.copy(), s.final_cost, s.iterations, s.successful_steps, s.unsuccessful_steps))
    for o in out[1:]:
        for a, b in zip(o, out[0]):
            assert np.array_equal(np.asarray(a), np.asarray(b))
    pb = bounded_problem(sc)
    so = oracle.solve(pb, threads=4)
    assert (out[0][4], out[0][5], out[0][6]) == (so.iterations, so.successful_steps, so.unsuccessful_steps)
    assert abs(out[0][3] - so.final_cost) <= 1e-8 * so.final_cost
