"""Randomised structure stress test of the sweep kernels through the C ABI: many small scenes of different shapes
(frame counts, visibility windows, camera configurations, constraints, recalibration masks), further deformed by dropping and
duplicating observations, compared with the oracle.  Meant to shake out layout / synchronisation corner cases of the planner
and of the LDS-window kernels (pass boundaries, idle waves, split groups, wide windows without the K-split region, special
points mixed with regular ones) — the full-size configs found one such case (a Z matrix rounded past its LDS region)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from lifcal_amd import BundleAdjustment, _capi as capi, scene
from tests.helpers import S, scaled_max_err, vec_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONFIGS = [0x506, 0xF06, 0xD01, 0x505, 0x500, 0xD04, 0xF02, 0x006, 0x306]


def deformed_problem(k):
    """scene k of the family + a deterministic deformation of its observation list"""
    rs = scene.Stream(9000 + k, 5)
    F = int(rs.integers(1, 60 if k % 5 else 140)[0]) + 1
    P = int(rs.integers(1, 300 if k % 5 else 2500)[0]) + 1
    window = [None, 2, 3, 6, 12, 19, 25][int(rs.integers(1, 7)[0])]
    if window is not None and window > F:
        window = None
    cfg = CONFIGS[int(rs.integers(1, len(CONFIGS))[0])]
    ncon = int(rs.integers(1, 4)[0]) if (cfg & 0x500) == 0x500 and P >= 8 and k % 3 == 0 else 0
    spec = S(F, P, window, cfg, 9100 + k, outlier_fraction=0.03 if cfg & 0x200 else 0.0, n_constraints=ncon, recalib=(k % 7 == 3))
    sc = scene.make_scene(spec)
    n = sc.n_obs
    keep = np.ones(n, bool)
    mode = k % 4
    if mode == 1 and n > 20:                      # ragged: drop a third of the observations at random
        keep = rs.uniform(n) > 0.33
    sel = np.flatnonzero(keep)
    if mode == 2 and n > 20:                      # a few very long groups: repeat the observations of two (point, frame) pairs
        key = sc.pt[sel].astype(np.int64) * 100000 + sc.fr[sel]
        big = sel[key == key[0]]
        sel = np.concatenate([sel, np.repeat(big, 90 if k % 8 == 2 else 12)])
    if mode == 3 and n > 20:                      # shuffled input order
        sel = sel[np.argsort(rs.uniform(len(sel)))]
    pa = lambda: capi.ProblemArrays(sc.u[sel], sc.v[sel], sc.mcx[sel], sc.mcy[sel], sc.pt[sel], sc.fr[sel], sc.cam0.copy(), sc.views0.copy(),
                                    sc.pts0.copy(), sc.spx, sc.scale, sc.config, fixed_mask=sc.fixed_mask, lower=sc.lower, upper=sc.upper,
                                    c_i=sc.c_i, c_j=sc.c_j, c_dist=sc.c_dist, c_sigma=sc.c_sigma, use_constraints=sc.use_constraints)
    return spec, pa, len(sel)


@pytest.mark.parametrize("chunk", range(30))
def test_random_structures_match_the_oracle(built, chunk):
    for k in range(chunk * 8, chunk * 8 + 8):
        spec, mk, n = deformed_problem(k)
        if n == 0:
            continue
        ref = oracle.sweep(mk(), radius=1e3, threads=4)
        with BundleAdjustment(mk()) as ba:
            got = ba.sweep(1e3, want_matrices=True)
        tag = f"case {k}: F={spec.n_frames} P={spec.n_points} window={spec.window} cfg={spec.config:#x} n={n}"
        assert abs(got.cost - ref.cost) <= 1e-11 * max(ref.cost, 1e-300), tag
        assert scaled_max_err(got.S, ref.S) < 1e-8, tag
        assert vec_err(got.rhs, ref.rhs) < 1e-8, tag
        assert vec_err(got.point_gradient, ref.point_gradient) < 1e-9, tag


@pytest.mark.parametrize("k", [1, 4, 6, 9, 12, 16, 18, 21, 27, 33, 36, 46])   # small members of the family (k % 5 != 0): the oracle solves them in well under a second
def test_random_structures_solve_like_the_oracle(built, k):
    spec, mk, n = deformed_problem(k)
    if n < 30:
        pytest.skip("too few observations for a meaningful solve")
    pb = mk()
    so = oracle.solve(pb, threads=4)
    pg = mk()
    with BundleAdjustment(pg) as ba:
        sg = ba.performBundleAdjustment()
    tag = f"case {k}: F={spec.n_frames} P={spec.n_points} window={spec.window} cfg={spec.config:#x} n={n}"
    assert (sg.iterations, sg.termination) == (so.iterations, so.termination), tag
    assert abs(sg.final_cost - so.final_cost) <= 1e-7 * max(so.final_cost, 1e-300), tag


@pytest.mark.parametrize("k", [3, 7, 17, 19, 27, 33, 42, 44, 54, 58])   # the members of the family the reduction applies to (poses, >= 4 super-blocks)
def test_random_structures_solve_like_the_oracle_with_the_block_reduction(built, monkeypatch, k):
    """the same deformed scenes with the reduced solve forced onto the block odd-even reduction (bandchol3.hpp) wherever it is
    applicable (ragged frame counts, padded last super-block, band widths 1 .. 10; the others fall back to the chains)"""
    monkeypatch.setenv("LIFCAL_CR", "1")
    spec, mk, n = deformed_problem(k)
    if n < 30:
        pytest.skip("too few observations for a meaningful solve")
    so = oracle.solve(mk(), threads=4)
    with BundleAdjustment(mk()) as ba:
        sg = ba.performBundleAdjustment()
    tag = f"case {k}: F={spec.n_frames} P={spec.n_points} window={spec.window} cfg={spec.config:#x} n={n}"
    assert (sg.iterations, sg.termination) == (so.iterations, so.termination), tag
    assert abs(sg.final_cost - so.final_cost) <= 1e-7 * max(so.final_cost, 1e-300), tag


_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from lifcal_amd import BundleAdjustment
from tests.test_gpu_stress import deformed_problem
out = []
for k in (%s):
    spec, mk, n = deformed_problem(k)
    with BundleAdjustment(mk()) as ba:
        r = ba.sweep(1e3, want_matrices=True)
    out.append(np.concatenate([[r.cost], r.S.ravel(), r.rhs]))
np.save(sys.argv[1], np.concatenate(out))
"""


def test_one_block_many_passes_and_both_kernels_agree(built, tmp_path):
    """the same cases with ONE workgroup for all regular points (many passes per block) and through k_sweep2"""
    ks = "3, 5, 10, 17, 22, 30, 41"
    res = {}
    for tag, env in (("default", {}), ("oneblock", {"LIFCAL_V2_BLOCKS": "1"}), ("k2", {"LIFCAL_SWEEP_KERNEL": "2"}), ("k2one", {"LIFCAL_SWEEP_KERNEL": "2", "LIFCAL_V2_BLOCKS": "1"})):
        out = os.path.join(str(tmp_path), tag + ".npy")
        e = dict(os.environ); e.update(env)
        subprocess.check_call([sys.executable, "-c", _CHILD % (ROOT, ks), out], env=e, cwd=ROOT)
        res[tag] = np.load(out)
    for tag in ("oneblock", "k2", "k2one"):
        assert vec_err(res[tag], res["default"]) < 1e-10, tag
