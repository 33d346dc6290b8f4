"""BASELINE.json configs at FULL size on one GPU (fp64 arms).

configs[0] (5 frames x 100 points, intrinsics only, arity <2,17>) and configs[1] (20 frames x 500 points, 0x506): small enough
for the oracle to form the reduced system and to solve, so they get the full parity treatment (sweep + trajectory).

configs[3]: 1000 frames, 50 k points (~2.4 M micro-image observations with this MLA), 0xF06 — the problem the 8-GPU run
shards; here the whole of it on one device.  configs[4]: recalib mode, 2000 frames, 100 k points (~4.8 M observations):
the reference's recalibration arm (camera slots {0,2} constant + box bounds 0.7/1.3 on slots 1,3,4, reference
src/CameraCalibration.cpp:927-952) and the BASELINE arm "intrinsics fixed, pose+point refine".
The oracle is too slow to form the Schur complement at these sizes, so the comparators are its cheap parts (cost,
reprojection statistics, both O(N) and threaded) and size-independent properties of the solve."""
import numpy as np
import pytest

import oracle
from lifcal_amd import BundleAdjustment, _capi as capi, scene
from tests.helpers import problem

pytestmark = pytest.mark.gpu


def check_solve(pa, fixed=(), lower=None, upper=None):
    cam0 = pa.cam.copy()
    threads = oracle.hardware_threads()
    c0 = oracle.cost(pa, threads=threads)
    with BundleAdjustment(pa) as ba:
        first = ba.sweep(1e4)
        assert abs(first.cost - c0) <= 1e-11 * c0                      # same objective at the start
        again = ba.sweep(1e4)
        assert abs(again.cost - first.cost) <= 1e-11 * first.cost      # idempotent up to summation order
        s = ba.performBundleAdjustment()
        st = ba.calcReprojectionError()
        end = ba.sweep(s.final_radius)
    assert s.termination in (1, 2), s.termination
    assert s.final_cost < 0.2 * s.initial_cost
    assert abs(end.cost - s.final_cost) <= 1e-9 * s.final_cost         # device-resident point == reported point
    c1 = oracle.cost(pa, threads=threads)                              # ... == the point written back to the caller
    assert abs(c1 - s.final_cost) <= 1e-9 * s.final_cost
    so = oracle.reproj_stats(pa)
    assert abs(st.std_x - so.std_x) < 1e-9 and abs(st.std_y - so.std_y) < 1e-9 and st.num_inliers == so.num_inliers
    assert st.std_x < 1.0 and st.std_y < 1.0 and st.num_inliers > 0.95 * st.num_points    # 2 % outliers at +-5 px
    for j in fixed:
        assert pa.cam[j] == cam0[j]                                    # constant slots come back bit-identical
    if lower is not None:
        assert np.all(pa.cam >= lower) and np.all(pa.cam <= upper)
    return s, st


@pytest.mark.parametrize("name", ["cfg1", "cfg2"])
def test_cfg1_cfg2_sweep_and_solve_match_the_oracle(built, name):
    from tests.helpers import scaled_max_err, vec_err
    sc = scene.make_scene(scene.baseline_spec(name))
    if name == "cfg1":
        assert (sc.spec.n_frames, sc.spec.n_points, sc.config) == (5, 100, 0x006)      # BASELINE configs[0]: intrinsics only
    else:
        assert (sc.spec.n_frames, sc.spec.n_points, sc.config) == (20, 500, 0x506)     # BASELINE configs[1]
    threads = oracle.hardware_threads()
    ref = oracle.sweep(problem(sc), radius=1e4, threads=threads)
    pa = problem(sc)
    with BundleAdjustment(pa) as ba:
        got = ba.sweep(1e4, want_matrices=True)
        assert got.n_reduced == ref.S.shape[0]
        assert abs(got.cost - ref.cost) <= 1e-12 * ref.cost
        assert scaled_max_err(got.S, ref.S) < 1e-9 and vec_err(got.rhs, ref.rhs) < 1e-9
        assert vec_err(got.gradient_reduced, ref.gradient_reduced) < 1e-10
        if sc.config & 0x400:
            assert vec_err(got.point_gradient, ref.point_gradient) < 1e-10
        s = ba.performBundleAdjustment()
        st = ba.calcReprojectionError()
    pb = problem(sc)
    so = oracle.solve(pb, threads=threads)
    assert (s.iterations, s.successful_steps, s.unsuccessful_steps, s.termination) == (so.iterations, so.successful_steps, so.unsuccessful_steps, so.termination)
    assert abs(s.final_cost - so.final_cost) <= 1e-8 * so.final_cost
    live = 5 + (sc.config & 3) + (2 if sc.config & 4 else 0)
    assert np.allclose(pa.cam[:live], pb.cam[:live], rtol=1e-6, atol=1e-12)           # north_star: within 1e-6 relative
    assert np.all(pa.cam[live:] == 0.0)
    assert np.allclose(pa.views, pb.views, rtol=0, atol=1e-6 * (1 + np.abs(pb.views).max()))
    assert np.allclose(pa.pts, pb.pts, rtol=0, atol=1e-6 * (1 + np.abs(pb.pts).max()))
    sto = oracle.reproj_stats(pb)
    assert abs(st.std_x - sto.std_x) < 1e-8 and abs(st.std_y - sto.std_y) < 1e-8 and st.num_points == sc.n_obs


def test_cfg4_whole_problem_on_one_gpu(built):
    sc = scene.make_scene(scene.baseline_spec("cfg4"))
    assert sc.spec.n_frames == 1000 and sc.spec.n_points == 50000 and sc.n_obs > 2_000_000
    pa = problem(sc)
    s, st = check_solve(pa)
    # the truth is known: intrinsics come back to the generator's values (gauge-free quantities) within the noise
    assert abs(pa.cam[0] - sc.spec.fL) < 2e-2 * sc.spec.fL and abs(pa.cam[2] - sc.spec.B) < 2e-2 * sc.spec.B


def test_cfg5_recalibration_arms(built):
    sc = scene.make_scene(scene.baseline_spec("cfg5"))
    assert sc.spec.n_frames == 2000 and sc.fixed_mask == 0b101 and sc.lower is not None
    # arm 1: the reference's recalibration set-up (slots 0 and 2 constant, bounds on 1, 3, 4)
    pa = problem(sc)
    check_solve(pa, fixed=(0, 2), lower=sc.lower, upper=sc.upper)
    # arm 2: BASELINE wording — every intrinsic constant (at its calibrated value), poses and points refined
    live = 5 + (sc.config & 3) + (2 if sc.config & 4 else 0)
    pa2 = capi.ProblemArrays(sc.u, sc.v, sc.mcx, sc.mcy, sc.pt, sc.fr, sc.cam_gt.copy(), sc.views0.copy(), sc.pts0.copy(),
                             sc.spx, sc.scale, sc.config, fixed_mask=(1 << live) - 1)
    check_solve(pa2, fixed=tuple(range(live)))


def test_cfg5_streaming_as_baseline_words_it(built):
    """BASELINE configs[4] in its own words: "recalib mode, intrinsics fixed, 2000-frame streaming pose+point refine, fp32 residuals / fp64
    normal-eq accumulate" — the frame-windowed driver (windows of 250 frames advancing by 200) with options.precision = 1, every intrinsic
    constant at its calibrated value.  One window is resident at a time; the result must be a solution of the WHOLE problem at the noise
    level, close to what the all-at-once solve of the same problem reaches."""
    from lifcal_amd import performBundleAdjustmentWindowed
    sc = scene.make_scene(scene.baseline_spec("cfg5"))
    live = 5 + (sc.config & 3) + (2 if sc.config & 4 else 0)
    mk = lambda: capi.ProblemArrays(sc.u, sc.v, sc.mcx, sc.mcy, sc.pt, sc.fr, sc.cam_gt.copy(), sc.views0.copy(), sc.pts0.copy(), sc.spx, sc.scale, sc.config,
                                    fixed_mask=(1 << live) - 1, use_constraints=0)
    o = capi.default_options_py(); o.precision = 1
    pw = mk()
    reps = performBundleAdjustmentWindowed(pw, 250, 50, options=o)
    assert [r.first_frame for r in reps] == list(range(0, 2000, 200)) and reps[-1].first_frame + reps[-1].n_frames == 2000
    assert all(r.summary.termination in (1, 2) for r in reps) and all(r.n_fixed_frames == 50 for r in reps[1:]) and reps[0].n_fixed_frames == 0
    assert max(r.n_obs for r in reps) < 0.2 * sc.n_obs                                   # a window is a fraction of the problem
    assert np.array_equal(pw.cam, sc.cam_gt)
    with BundleAdjustment(pw) as ba:
        st_win = ba.calcReprojectionError()
    pf = mk()
    with BundleAdjustment(pf, o) as ba:
        ba.performBundleAdjustment()
        st_full = ba.calcReprojectionError()
    assert st_win.num_points == sc.n_obs and st_win.std_x < 1.1 * st_full.std_x and st_win.std_y < 1.1 * st_full.std_y
    assert st_win.std_x < 0.6 and st_win.num_inliers > 0.95 * sc.n_obs                   # 0.1 px noise + 2 % outliers at +-5 px
