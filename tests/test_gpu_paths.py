"""The two sweep implementations against each other and against the oracle on problems that exercise the
routing: v2 (LDS-window kernel, regular points) vs v1 (global-atomic kernels, the general fallback)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from lifcal_amd import BundleAdjustment, _capi as capi, scene
from tests.helpers import S, problem, scaled_max_err, vec_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import sys, json, numpy as np
sys.path.insert(0, %r)
from lifcal_amd import BundleAdjustment, _capi as capi, scene
from tests.helpers import S
spec = S(%s)
sc = scene.make_scene(spec)
with BundleAdjustment(capi.ProblemArrays.from_scene(sc)) as ba:
    r = ba.sweep(123.0, want_matrices=True)
    info = ba.info()
np.savez(sys.argv[1], S=r.S, rhs=r.rhs, cost=r.cost, pg=r.point_gradient, ui=r.point_hessian_inv, chunks=info.n_chunks)
"""


def run_child(tmp_path, spec_args, env_extra, tag):
    out = os.path.join(str(tmp_path), tag + ".npz")
    env = dict(os.environ); env.update(env_extra)
    subprocess.check_call([sys.executable, "-c", _CHILD % (ROOT, spec_args), out], env=env, cwd=ROOT)
    return np.load(out)


@pytest.mark.parametrize("spec_args", [
    "24, 160, 6, 0xF06, 1201, outlier_fraction=0.02",
    "8, 60, None, 0x506, 1202",
    "40, 400, 9, 0xD05, 1203",
])
def test_v1_and_v2_kernels_agree(built, tmp_path, spec_args):
    """same library, LIFCAL_DISABLE_V2 routes everything through the global-atomic kernels"""
    a = run_child(tmp_path, spec_args, {}, "v2")
    b = run_child(tmp_path, spec_args, {"LIFCAL_DISABLE_V2": "1"}, "v1")
    assert int(a["chunks"]) > 0 and int(b["chunks"]) == 0
    assert abs(float(a["cost"]) - float(b["cost"])) <= 1e-13 * float(b["cost"])
    assert scaled_max_err(a["S"], b["S"]) < 1e-10 and vec_err(a["rhs"], b["rhs"]) < 1e-10
    assert vec_err(a["pg"], b["pg"]) < 1e-11 and vec_err(a["ui"], b["ui"]) < 1e-10


@pytest.mark.parametrize("blocks", ["1", "3", "7", "64"])
def test_block_count_does_not_change_the_result(built, tmp_path, blocks):
    spec_args = "30, 300, 8, 0xF06, 1204, outlier_fraction=0.02"
    a = run_child(tmp_path, spec_args, {"LIFCAL_V2_BLOCKS": blocks}, "b" + blocks)
    b = run_child(tmp_path, spec_args, {"LIFCAL_DISABLE_V2": "1"}, "ref")
    assert int(a["chunks"]) >= 1
    assert scaled_max_err(a["S"], b["S"]) < 1e-10 and vec_err(a["rhs"], b["rhs"]) < 1e-10


@pytest.mark.parametrize("split", ["0", "1", "2", "3", "6"])
def test_group_splitting_does_not_change_the_result(built, tmp_path, split):
    """LIFCAL_GROUP_SPLIT cuts (point, frame) groups into several lanes of at most that many observations; the per-lane
    blocks add up, so the reduced system, the point gradients and U^-1 must not move (1 = one observation per lane)"""
    spec_args = "30, 300, 8, 0xF06, 1206, outlier_fraction=0.02"
    a = run_child(tmp_path, spec_args, {"LIFCAL_GROUP_SPLIT": split}, "s" + split)
    b = run_child(tmp_path, spec_args, {"LIFCAL_DISABLE_V2": "1"}, "ref")
    assert int(a["chunks"]) >= 1
    assert abs(float(a["cost"]) - float(b["cost"])) <= 1e-13 * float(b["cost"])
    assert scaled_max_err(a["S"], b["S"]) < 1e-10 and vec_err(a["rhs"], b["rhs"]) < 1e-10
    assert vec_err(a["pg"], b["pg"]) < 1e-11 and vec_err(a["ui"], b["ui"]) < 1e-10


@pytest.mark.parametrize("spec_args", [
    "24, 160, 6, 0xF06, 1211, outlier_fraction=0.02",
    "12, 90, None, 0x501, 1212",
    "30, 250, 9, 0xD04, 1213",
    "40, 400, 12, 0x506, 1214",
    "60, 700, 8, 0xF06, 1215, outlier_fraction=0.02",
])
def test_two_kernel_sweep_agrees(built, tmp_path, spec_args):
    """k_front4 + k_back4 (LIFCAL_SWEEP_KERNEL=4: tiles of whole points worked by evaluator / accumulator / emitter waves, the point
    elimination in its own 1024-thread kernel; sweep4.hpp) against the global-atomic kernels and against the default k_sweep3"""
    a = run_child(tmp_path, spec_args, {"LIFCAL_SWEEP_KERNEL": "4"}, "k4")
    b = run_child(tmp_path, spec_args, {"LIFCAL_DISABLE_V2": "1"}, "ref")
    c = run_child(tmp_path, spec_args, {}, "k3")
    assert int(a["chunks"]) >= 1
    for o in (b, c):
        assert abs(float(a["cost"]) - float(o["cost"])) <= 1e-13 * float(o["cost"])
        assert scaled_max_err(a["S"], o["S"]) < 1e-10 and vec_err(a["rhs"], o["rhs"]) < 1e-10
        assert vec_err(a["pg"], o["pg"]) < 1e-11 and vec_err(a["ui"], o["ui"]) < 1e-10


def test_two_kernel_sweep_follows_the_oracle_through_a_solve(built, monkeypatch):
    """the same LM trajectory as the oracle with LIFCAL_SWEEP_KERNEL=4 (diagonal-only pass, Jacobi scaling, candidate evaluation)"""
    import oracle
    monkeypatch.setenv("LIFCAL_SWEEP_KERNEL", "4")
    sc = scene.make_scene(S(30, 300, 8, 0xF06, 1216, outlier_fraction=0.02))
    pa = problem(sc)
    with BundleAdjustment(pa) as ba:
        s = ba.performBundleAdjustment()
    so = oracle.solve(problem(sc), threads=4)
    assert (s.iterations, s.successful_steps, s.termination) == (so.iterations, so.successful_steps, so.termination)
    assert abs(s.final_cost - so.final_cost) <= 1e-8 * so.final_cost


@pytest.mark.parametrize("spec_args", [
    "24, 160, 6, 0xF06, 1207, outlier_fraction=0.02",
    "12, 90, None, 0x501, 1208",
    "30, 250, 9, 0xD04, 1209",
])
def test_wave_specialised_and_single_role_window_kernels_agree(built, tmp_path, spec_args):
    """k_sweep3 (512 threads, evaluator / accumulator waves, default) against k_sweep2 (LIFCAL_SWEEP_KERNEL=2)"""
    a = run_child(tmp_path, spec_args, {}, "k3")
    b = run_child(tmp_path, spec_args, {"LIFCAL_SWEEP_KERNEL": "2"}, "k2")
    assert int(a["chunks"]) > 0 and int(b["chunks"]) > 0
    assert abs(float(a["cost"]) - float(b["cost"])) <= 1e-13 * float(b["cost"])
    assert scaled_max_err(a["S"], b["S"]) < 1e-11 and vec_err(a["rhs"], b["rhs"]) < 1e-11
    assert vec_err(a["pg"], b["pg"]) < 1e-12 and vec_err(a["ui"], b["ui"]) < 1e-11


def test_mixed_regular_and_oversized_points(built):
    """a few points are also seen 25 frames later (span > 20 frame window): they take the fallback path, the rest v2"""
    sc = scene.make_scene(S(40, 300, 8, 0xF06, 1205, outlier_fraction=0.02))
    fr = sc.fr.copy()
    moved = 0
    for p in (5, 17, 40):
        idx = np.flatnonzero(sc.pt == p)
        f0 = sc.fr[idx].min()
        if f0 + 27 < 40:
            sel = idx[sc.fr[idx] == f0]
            fr[sel] = f0 + 27; moved += len(sel)
    assert moved > 0
    mk = lambda: capi.ProblemArrays(sc.u, sc.v, sc.mcx, sc.mcy, sc.pt, fr, sc.cam0, sc.views0, sc.pts0, sc.spx, sc.scale, sc.config)
    ref = oracle.sweep(mk(), radius=1e4, threads=4)
    with BundleAdjustment(mk()) as ba:
        got = ba.sweep(1e4, want_matrices=True)
        info = ba.info()
        s = ba.performBundleAdjustment()
    assert info.n_chunks >= 1
    assert abs(got.cost - ref.cost) <= 1e-12 * ref.cost
    assert scaled_max_err(got.S, ref.S) < 1e-9 and vec_err(got.rhs, ref.rhs) < 1e-9
    assert vec_err(got.point_gradient, ref.point_gradient) < 1e-10
    so = oracle.solve(mk(), threads=4)
    assert s.iterations == so.iterations and abs(s.final_cost - so.final_cost) <= 1e-8 * so.final_cost


def test_window_wider_than_lds_limit_falls_back(built):
    """most points span more than the 20-frame LDS window: they are routed to the fallback kernels"""
    sc = scene.make_scene(S(40, 60, 30, 0x506, 1206))
    ref = oracle.sweep(problem(sc), radius=1e4, threads=4)
    with BundleAdjustment(problem(sc)) as ba:
        got = ba.sweep(1e4, want_matrices=True)
    assert scaled_max_err(got.S, ref.S) < 1e-9 and vec_err(got.rhs, ref.rhs) < 1e-9


def test_huge_group_is_not_truncated(built):
    """300 observations of one (point, frame) pair: more than the 8-bit group size of the v2 slot word"""
    sc = scene.make_scene(S(6, 30, None, 0x506, 1207))
    i0 = 10
    rep = np.full(300, i0)
    cat = lambda a: np.concatenate([a, a[rep]])
    noise = scene.Stream(3, 3).normal(300, 0.05)
    u = cat(sc.u); u[-300:] += noise
    mk = lambda: capi.ProblemArrays(u, cat(sc.v), cat(sc.mcx), cat(sc.mcy), cat(sc.pt), cat(sc.fr), sc.cam0, sc.views0, sc.pts0, sc.spx, sc.scale, sc.config)
    ref = oracle.sweep(mk(), radius=1e4)
    with BundleAdjustment(mk()) as ba:
        got = ba.sweep(1e4, want_matrices=True)
    assert abs(got.cost - ref.cost) <= 1e-12 * ref.cost and scaled_max_err(got.S, ref.S) < 1e-9


def ordered():
    """options.deterministic = 1 for the comparisons of two reduced solvers: the sweeps of both arms then sum in one fixed order, so what
    the comparison sees is the round-off of the two ELIMINATION orders alone (with atomic sums the arms also differ by the summation
    order of every sweep, amplified along the gauge directions over ten iterations: the camera assertion failed once in eight runs)"""
    o = capi.default_options_py(); o.deterministic = 1
    return o


@pytest.mark.parametrize("spec", [S(24, 120, 6, 0xF06, 6101, outlier_fraction=0.02), S(60, 400, 8, 0x506, 6102), S(41, 300, 10, 0xF06, 6103, recalib=True, outlier_fraction=0.02),
                                  S(23, 100, 6, 0xF06, 6104)],
                         ids=["w6_f24", "w8_f60", "w10_f41_recalib", "w6_f23_single_chain_limit"])
def test_twisted_band_factorisation_equals_the_single_chain(built, monkeypatch, spec):
    """bandchol2.hpp: eliminating the pose chain from both ends on two workgroups (+ the middle frames and the arrow last) solves the
    same reduced system as the single chain of bandchol.hpp: same LM trajectory, same result (to the round-off of a different
    elimination order), and both follow the oracle"""
    import oracle
    sc = scene.make_scene(spec)
    res = {}
    for tw in ("1", "0"):
        monkeypatch.setenv("LIFCAL_TWISTED", tw)
        pa = problem(sc)
        with BundleAdjustment(pa, ordered()) as ba:
            s = ba.performBundleAdjustment()
        res[tw] = (pa, s)
    (p1, s1), (p0, s0) = res["1"], res["0"]
    assert (s1.iterations, s1.successful_steps, s1.unsuccessful_steps, s1.termination) == (s0.iterations, s0.successful_steps, s0.unsuccessful_steps, s0.termination)
    assert abs(s1.final_cost - s0.final_cost) <= 1e-11 * s0.final_cost
    # (poses and points move along the weakly damped gauge directions: round-off of the two elimination orders shows there first)
    assert np.allclose(p1.cam, p0.cam, rtol=1e-8, atol=1e-13)
    assert np.allclose(p1.views, p0.views, rtol=0, atol=1e-6 * (1 + np.abs(p0.views).max())) and np.allclose(p1.pts, p0.pts, rtol=0, atol=1e-6 * (1 + np.abs(p0.pts).max()))
    pb = problem(sc)
    so = oracle.solve(pb, threads=4)
    assert (s1.iterations, s1.termination) == (so.iterations, so.termination)
    assert abs(s1.final_cost - so.final_cost) <= 1e-8 * so.final_cost


@pytest.mark.parametrize("spec", [S(24, 120, 6, 0xF06, 6201, outlier_fraction=0.02), S(60, 400, 8, 0x506, 6202), S(41, 300, 10, 0xF06, 6203, recalib=True, outlier_fraction=0.02),
                                  S(97, 500, 4, 0xF06, 6204), S(50, 150, 3, 0x706, 6205), S(40, 220, 4, 0xF06, 6206, n_constraints=2, outlier_fraction=0.02),
                                  S(36, 200, 3, 0x506, 6207, n_constraints=4)],
                         ids=["w6_f24", "w8_f60", "w10_f41_recalib", "w4_f97", "w3_f50", "w4_f40_two_constraints_23_arrow_rows", "w3_f36_four_constraints_29_arrow_rows"])
def test_block_odd_even_reduction_equals_the_chain(built, monkeypatch, spec):
    """bandchol3.hpp: the block odd-even reduction of the band + arrow system (log2(F / bw) levels, one workgroup per eliminated
    super-block) solves the same reduced system as the chain factorisations: same LM trajectory, same result to the round-off of
    another elimination order, and both follow the oracle"""
    import oracle
    sc = scene.make_scene(spec)
    res = {}
    for cr in ("1", "0"):
        monkeypatch.setenv("LIFCAL_CR", cr)
        pa = problem(sc)
        with BundleAdjustment(pa, ordered()) as ba:
            s = ba.performBundleAdjustment()
        res[cr] = (pa, s)
    (p1, s1), (p0, s0) = res["1"], res["0"]
    assert (s1.iterations, s1.successful_steps, s1.unsuccessful_steps, s1.termination) == (s0.iterations, s0.successful_steps, s0.unsuccessful_steps, s0.termination)
    assert abs(s1.final_cost - s0.final_cost) <= 1e-11 * s0.final_cost
    assert np.allclose(p1.cam, p0.cam, rtol=1e-8, atol=1e-13)
    assert np.allclose(p1.views, p0.views, rtol=0, atol=1e-6 * (1 + np.abs(p0.views).max())) and np.allclose(p1.pts, p0.pts, rtol=0, atol=1e-6 * (1 + np.abs(p0.pts).max()))
    pb = problem(sc)
    so = oracle.solve(pb, threads=4)
    assert (s1.iterations, s1.termination) == (so.iterations, so.termination)
    assert abs(s1.final_cost - so.final_cost) <= 1e-8 * so.final_cost
