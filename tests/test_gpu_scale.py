"""Full-size checks on the GPU through size-independent properties (the oracle is too slow / too big to be
the comparator at 1 M observations for every quantity), plus the multi-rank plumbing on one GPU."""
import numpy as np
import pytest

import oracle
from lifcal_amd import BundleAdjustment, _capi as capi, scene, comm_unique_id
from tests.helpers import S, problem, scaled_max_err, vec_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cfg3():
    return scene.make_scene(scene.baseline_spec("cfg3"))


def test_cfg3_sweep_against_oracle(built, cfg3):
    """BASELINE configs[2]: 100 frames, 5k points (~300k obs), mlCenterAdj + robust; oracle still runs in seconds"""
    ref = oracle.sweep(problem(cfg3), radius=1e4, threads=oracle.hardware_threads())
    with BundleAdjustment(problem(cfg3)) as ba:
        got = ba.sweep(1e4, want_matrices=True)
    assert abs(got.cost - ref.cost) <= 1e-12 * ref.cost
    assert scaled_max_err(got.S, ref.S) < 1e-8
    assert vec_err(got.rhs, ref.rhs) < 1e-8 and vec_err(got.point_gradient, ref.point_gradient) < 1e-10


def test_cfg3_solve_properties(built, cfg3):
    pa = problem(cfg3)
    with BundleAdjustment(pa) as ba:
        s = ba.performBundleAdjustment()
        st = ba.calcReprojectionError()
        again = ba.sweep(s.final_radius)
    assert s.termination in (1, 2) and s.final_cost < 0.2 * s.initial_cost
    assert abs(again.cost - s.final_cost) <= 1e-9 * s.final_cost      # device-resident point == reported point
    so = oracle.reproj_stats(pa)                                       # downloaded parameters give the same statistics
    assert abs(st.std_x - so.std_x) < 1e-9 and abs(st.std_y - so.std_y) < 1e-9 and st.num_inliers == so.num_inliers
    assert st.std_x < 1.0 and st.num_inliers > 0.95 * st.num_points    # 2 % outliers at +-5 px


def test_metric_point_properties(built):
    """the 1 M-observation workload of bench.py: invariants that do not need a CPU comparator"""
    sc = scene.make_scene(scene.baseline_spec("metric"))
    assert 0.95e6 < sc.n_obs < 1.05e6
    with BundleAdjustment(problem(sc)) as ba:
        a = ba.sweep(1e4, want_matrices=True)
        b = ba.sweep(1e4)
        assert abs(a.cost - b.cost) <= 1e-11 * a.cost                 # atomics reorder sums, nothing else
        c = oracle.cost(problem(sc), threads=oracle.hardware_threads())
        assert abs(a.cost - c) <= 1e-11 * c                            # cost is cheap enough to check exactly
        w = np.linalg.eigvalsh(a.S)
        assert w.min() > 0                                             # damped reduced system is SPD
        x = np.linalg.solve(a.S, a.rhs)
        assert a.rhs @ x > 0                                           # descent direction: g^T delta < 0
        # trust region: a smaller radius gives a shorter step
        small = ba.sweep(1e-2, want_matrices=True)
        assert np.linalg.norm(np.linalg.solve(small.S, small.rhs)) < np.linalg.norm(x)
    # noise-free copy of a mid-size windowed scene converges to (numerically) zero cost
    nf = scene.make_scene(S(60, 1500, 8, 0xD06, 901, noise_px=0.0))
    o = capi.default_options_py(); o.function_tolerance = 1e-14; o.parameter_tolerance = 1e-14
    with BundleAdjustment(problem(nf), o) as ba:
        s = ba.performBundleAdjustment()
    assert s.final_cost < 1e-10 * s.initial_cost


def _web_scene(name):
    """a *_web workload as bench.py builds it: the lenses of every image point chosen by the library's GPU port of the reference's
    generator (lifcal_mla_project)"""
    from lifcal_amd.mla import MicroLensGrid
    spec = scene.baseline_spec(name)
    grid = MicroLensGrid(spec.raw_width, spec.raw_height, spec.lens_diameter, spec.lens_base_y, spec.grid_rotation, spec.grid_offset, True, device=0)

    def selector(img_x, img_y, img_vd, img_fr, img_pt, scale):
        o = grid.projectPointsToRawImage(img_x, img_y, img_vd, int(scale), fr=img_fr, pt=img_pt)
        return o.src, o.mcx, o.mcy
    try:
        return scene.make_scene(spec, lens_selector=selector)
    finally:
        grid.close()


def test_benchmarked_workload_full_matrix_parity(built):
    """metric_web — the workload BENCH is quoted on — through one sweep against the oracle on every host thread: cost, the whole
    reduced matrix, right-hand side, point gradients (the oracle takes about a second on the GPU box's host)"""
    sc = _web_scene("metric_web")
    assert 0.95e6 < sc.n_obs < 1.05e6
    ref = oracle.sweep(problem(sc), radius=1e4, threads=oracle.hardware_threads())
    with BundleAdjustment(problem(sc)) as ba:
        got = ba.sweep(1e4, want_matrices=True)
    assert abs(got.cost - ref.cost) <= 1e-12 * ref.cost
    assert scaled_max_err(got.S, ref.S) < 1e-8
    assert vec_err(got.rhs, ref.rhs) < 1e-8 and vec_err(got.point_gradient, ref.point_gradient) < 1e-10


def test_cfg4_spot_checks_against_the_oracle(built):
    """BASELINE configs[3] (1000 frames, 50 k points, ~2.4 M observations): the oracle's dense 6017^2 elimination is too slow for a
    unit test, so the comparators are (1) its cost on the whole problem, (2) its sweep on three SUB-PROBLEMS — the observations of a
    40-frame window — whose pose gradients J_f^T r are sums over the frame's own observations, i.e. equal to the full problem's"""
    sc = scene.make_scene(scene.baseline_spec("cfg4"))
    with BundleAdjustment(problem(sc)) as ba:
        full = ba.sweep(1e4, want_matrices=True)
    c = oracle.cost(problem(sc), threads=oracle.hardware_threads())
    assert abs(full.cost - c) <= 1e-11 * c
    F = sc.spec.n_frames
    views = sc.views0.reshape(-1, 6); pts = sc.pts0.reshape(-1, 3)
    for f0 in (0, F // 2 - 20, F - 40):
        sel = (sc.fr >= f0) & (sc.fr < f0 + 40)
        used = np.unique(sc.pt[sel])
        remap = np.zeros(sc.spec.n_points, np.int64); remap[used] = np.arange(len(used))
        sub = capi.ProblemArrays(sc.u[sel], sc.v[sel], sc.mcx[sel], sc.mcy[sel], remap[sc.pt[sel]], sc.fr[sel] - f0, sc.cam0,
                                 views[f0:f0 + 40].reshape(-1), pts[used].reshape(-1), sc.spx, sc.scale, sc.config, use_constraints=0)
        ref = oracle.sweep(sub, radius=1e4, threads=oracle.hardware_threads())
        assert vec_err(full.gradient_reduced[17 + 6 * f0:17 + 6 * (f0 + 40)], ref.gradient_reduced[17:17 + 240]) < 1e-9


def test_rccl_single_rank_and_hook_paths(built):
    """world_size = 1 through a real RCCL communicator must not change anything; a summing hook doubles as a
    2-rank rehearsal on one GPU: rank r owns half the points, the hook adds the other rank's partial block."""
    import ctypes as C
    sc = scene.make_scene(S(24, 160, 6, 0xF06, 902, outlier_fraction=0.02))
    ref = oracle.sweep(problem(sc), radius=1e4, threads=4)
    o = capi.default_options_py()
    with BundleAdjustment(problem(sc), o) as ba:
        ba.comm_init_rccl(comm_unique_id())
        got = ba.sweep(1e4, want_matrices=True)
        assert scaled_max_err(got.S, ref.S) < 1e-9
    # the slab exchange (pack -> ncclAllGather -> unpack) through the real RCCL communicator, forced at world size 1
    import os
    os.environ["LIFCAL_FORCE_EXCHANGE"] = "1"
    try:
        with BundleAdjustment(problem(sc), o) as ba:
            ba.comm_init_rccl(comm_unique_id())
            got = ba.sweep(1e4, want_matrices=True)
            assert abs(got.cost - ref.cost) <= 1e-12 * ref.cost
            assert scaled_max_err(got.S, ref.S) < 1e-9 and vec_err(got.rhs, ref.rhs) < 1e-9
            s = ba.performBundleAdjustment()
            assert s.termination in (1, 2)
    finally:
        del os.environ["LIFCAL_FORCE_EXCHANGE"]
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    # sequential two-rank emulation: run rank 1 first and record every buffer it would contribute,
    # then run rank 0 with a hook that adds rank 1's recorded partials (and vice versa for the 1st pass)
    rec = {0: [], 1: []}

    def make(rank, other, replay):
        oo = capi.default_options_py(); oo.world_size = 2; oo.rank = rank
        ba = BundleAdjustment(problem(sc), oo)
        state = {"k": 0}

        def hook(ptr, count, stream):
            hip.hipStreamSynchronize(stream)
            buf = np.zeros(count)
            hip.hipMemcpy(buf.ctypes.data, ptr, count * 8, 2)
            rec[rank].append(buf.copy())
            if replay:
                buf = buf + rec[other][state["k"]]
                hip.hipMemcpy(ptr, buf.ctypes.data, count * 8, 1)
            state["k"] += 1
            return 0
        ba.set_allreduce(hook)
        return ba
    # the Jacobi scaling needs the all-reduced diagonal, so iterate once to fill the recordings consistently
    with make(1, 0, False) as b1:
        b1.sweep(1e4)
    with make(0, 1, True) as b0:
        g0 = b0.sweep(1e4, want_matrices=True)
    assert abs(g0.cost - ref.cost) <= 1e-12 * ref.cost
    assert scaled_max_err(g0.S, ref.S) < 1e-9 and vec_err(g0.rhs, ref.rhs) < 1e-9
