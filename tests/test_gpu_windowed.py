"""Constant poses (lifcal_ba_set_fixed_frames) and the frame-windowed driver built on them (lifcal_ba_solve_windowed; BASELINE
configs[4] "2000-frame streaming pose + point refine").  Neither has a reference counterpart: LiFCal always hands every pose to
ceres.  The oracle gets the same switch (ceres semantics of SetParameterBlockConstant: the residual blocks stay, the pose block
has no Jacobian), so parity is checked the usual way; the windowed driver is checked against the SAME window schedule driven
through the oracle one window at a time."""
import numpy as np
import pytest

import oracle
from lifcal_amd import BundleAdjustment, _capi as capi, scene, performBundleAdjustmentWindowed
from tests.helpers import S, problem, scaled_max_err, vec_err

pytestmark = pytest.mark.gpu

CASES = [
    ("full", S(6, 40, None, 0xF06, 2101, outlier_fraction=0.03), [1, 4]),
    ("windowed", S(24, 120, 6, 0xF06, 2102, outlier_fraction=0.02), [0, 1, 2, 3, 4]),
    ("constraints_special_path", S(6, 40, None, 0x506, 2103, n_constraints=3), [0, 2]),
    ("poses_only_arity", S(6, 40, None, 0x306, 2104), [3]),
]


@pytest.fixture(autouse=True)
def _clear_oracle_mask():
    yield
    oracle.set_fixed_frames(None)


@pytest.mark.parametrize("name,spec,fixed_ids", CASES, ids=[c[0] for c in CASES])
def test_constant_poses_match_the_oracle(built, name, spec, fixed_ids):
    sc = scene.make_scene(spec)
    mask = np.zeros(spec.n_frames, np.uint8); mask[fixed_ids] = 1
    oracle.set_fixed_frames(mask)
    ref = oracle.sweep(problem(sc), radius=1e3, threads=4)
    pa = problem(sc)
    with BundleAdjustment(pa) as ba:
        ba.set_fixed_frames(mask)
        got = ba.sweep(1e3, want_matrices=True)
        assert abs(got.cost - ref.cost) <= 1e-12 * ref.cost
        assert scaled_max_err(got.S, ref.S) < 1e-9 and vec_err(got.rhs, ref.rhs) < 1e-9
        for f in fixed_ids:   # identity rows, zero right-hand side: the pose is not a column of the problem
            blk = slice(17 + 6 * f, 17 + 6 * f + 6)
            assert np.array_equal(got.S[blk, blk], np.eye(6)) and np.all(got.rhs[blk] == 0)
            off = got.S[blk].copy(); off[:, blk] = 0
            assert np.all(off == 0)
        s = ba.performBundleAdjustment()
        ba.set_fixed_frames(None)          # the switch is reversible
        free = ba.sweep(1e3, want_matrices=True)
        assert not np.array_equal(free.S[17 + 6 * fixed_ids[0], :], got.S[17 + 6 * fixed_ids[0], :])
    pb = problem(sc)
    so = oracle.solve(pb, threads=4)
    assert (s.iterations, s.successful_steps, s.unsuccessful_steps, s.termination) == (so.iterations, so.successful_steps, so.unsuccessful_steps, so.termination)
    assert abs(s.final_cost - so.final_cost) <= 1e-8 * so.final_cost
    v0 = sc.views0.reshape(-1, 6)
    assert np.array_equal(pa.views.reshape(-1, 6)[fixed_ids], v0[fixed_ids])         # bit-identical: never touched
    assert not np.allclose(pa.views.reshape(-1, 6)[[f for f in range(spec.n_frames) if f not in fixed_ids]], np.delete(v0, fixed_ids, 0), atol=1e-9)
    assert np.allclose(pa.views, pb.views, rtol=0, atol=1e-6 * (1 + np.abs(pb.views).max()))
    assert np.allclose(pa.cam[:9], pb.cam[:9], rtol=1e-6, atol=1e-12)


def oracle_windowed(sc, pa, window, overlap):
    """the schedule of lifcal_ba_solve_windowed driven through the oracle, one window at a time"""
    F = pa.struct.n_frames
    step = window - overlap
    a = 0; k = 0; reports = []
    while True:
        b = min(F, a + window)
        sel = np.flatnonzero((pa.fr >= a) & (pa.fr < b))
        ids = np.unique(pa.pt[sel])
        loc = -np.ones(pa.struct.n_points, np.int64); loc[ids] = np.arange(len(ids))
        sub = capi.ProblemArrays(pa.u[sel], pa.v[sel], pa.mcx[sel], pa.mcy[sel], loc[pa.pt[sel]], pa.fr[sel] - a, pa.cam, pa.views[6 * a:6 * b],
                                 pa.pts.reshape(-1, 3)[ids], sc.spx, sc.scale, sc.config, fixed_mask=pa.struct.fixed_mask, lower=pa.lower, upper=pa.upper, use_constraints=0)
        mask = np.zeros(b - a, np.uint8)
        if k > 0:
            mask[:min(b - a, overlap)] = 1
        oracle.set_fixed_frames(mask)
        s = oracle.solve(sub, threads=4)
        oracle.set_fixed_frames(None)
        pa.cam[:] = sub.cam; pa.views[6 * a:6 * b] = sub.views; pa.pts.reshape(-1, 3)[ids] = sub.pts.reshape(-1, 3)
        reports.append((a, b - a, len(ids), len(sel), s.iterations, s.termination))
        if b == F:
            break
        a += step; k += 1
    return reports


def test_windowed_driver_follows_its_schedule_like_the_oracle(built):
    """60 frames, windows of 24 advancing by 16 (overlap 8), intrinsics constant (BASELINE configs[4]: "intrinsics fixed, streaming
    pose + point refine"): window by window the same problems, the same iteration counts and the same result as the oracle driven
    through the same schedule; poses of an overlap are never touched by the later window."""
    spec = S(60, 400, 8, 0xF06, 2110, outlier_fraction=0.02)
    sc = scene.make_scene(spec)
    live = 5 + (sc.config & 3) + (2 if sc.config & 4 else 0)
    mk = lambda: capi.ProblemArrays(sc.u, sc.v, sc.mcx, sc.mcy, sc.pt, sc.fr, sc.cam_gt.copy(), sc.views0.copy(), sc.pts0.copy(), sc.spx, sc.scale, sc.config,
                                    fixed_mask=(1 << live) - 1, use_constraints=0)
    pg = mk()
    reps = performBundleAdjustmentWindowed(pg, 24, 8)
    po = mk()
    oreps = oracle_windowed(sc, po, 24, 8)
    assert [(r.first_frame, r.n_frames, r.n_points, r.n_obs) for r in reps] == [o[:4] for o in oreps]
    assert [r.first_frame for r in reps] == [0, 16, 32, 48] and reps[-1].first_frame + reps[-1].n_frames == 60
    assert [r.n_fixed_frames for r in reps] == [0, 8, 8, 8]
    assert [(r.summary.iterations, r.summary.termination) for r in reps] == [o[4:] for o in oreps]
    assert np.array_equal(pg.cam, sc.cam_gt)                                            # constant intrinsics come back untouched
    assert np.allclose(pg.views, po.views, rtol=0, atol=1e-6 * (1 + np.abs(po.views).max()))
    assert np.allclose(pg.pts, po.pts, rtol=0, atol=1e-6 * (1 + np.abs(po.pts).max()))
    # the windowed result is a good solution of the WHOLE problem: reprojection RMS at the noise level, close to the full solve's
    full = mk()
    with BundleAdjustment(full) as ba:
        ba.performBundleAdjustment()
        st_full = ba.calcReprojectionError()
    with BundleAdjustment(pg) as ba:
        st_win = ba.calcReprojectionError()
    assert st_win.std_x < 1.15 * st_full.std_x and st_win.std_y < 1.15 * st_full.std_y and st_win.num_points == sc.n_obs


def test_windowed_driver_argument_checks(built):
    sc = scene.make_scene(S(6, 40, None, 0x506, 2120))
    pa = problem(sc)
    from lifcal_amd import LifcalError
    with pytest.raises(LifcalError):
        performBundleAdjustmentWindowed(pa, 4, 4)       # overlap must be smaller than the window
    with pytest.raises(LifcalError):
        performBundleAdjustmentWindowed(pa, 0, 0)
    reps = performBundleAdjustmentWindowed(pa, 100, 10)   # one window covers everything: the ordinary solve
    pb = problem(sc)
    so = oracle.solve(pb, threads=4)
    assert len(reps) == 1 and (reps[0].summary.iterations, reps[0].summary.termination) == (so.iterations, so.termination)
    assert np.allclose(pa.cam[:9], pb.cam[:9], rtol=1e-6)
