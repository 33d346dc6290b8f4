"""SURVEY.md 8(f) f2 — start values of the plenoptic parameters (reference src/CameraCalibration.cpp:456-499).
CPU: the oracle's restatement (one-sided Jacobi SVD of the n x 2 system) against numpy's SVD-based lstsq, which has
the same minimum-norm semantics as Eigen::JacobiSVD::solve, on full-rank, masked and rank-deficient inputs."""
import numpy as np
import pytest

import oracle
from lifcal_amd import _capi as capi, scene


def init_inputs(spec, seed=0, noise=0.0):
    """image points of a synthetic scene: one per (point, frame) pair, virtual depth from the generator's camera"""
    sc = scene.make_scene(spec)
    key = np.unique(sc.pt.astype(np.int64) * 1000003 + sc.fr)
    pt = (key // 1000003).astype(np.uint32); fr = (key % 1000003).astype(np.uint32)
    R = scene.euler_xyz(sc.views_gt.reshape(-1, 6)[:, :3])
    w2c = np.zeros((len(R), 4, 4)); w2c[:, :3, :3] = R; w2c[:, :3, 3] = sc.views_gt.reshape(-1, 6)[:, 3:]; w2c[:, 3, 3] = 1.0
    P = sc.pts_gt.reshape(-1, 3)
    z = np.einsum("nj,nj->n", w2c[fr, 2, :3], P[pt]) + w2c[fr, 2, 3]
    fL, B, bL0 = sc.spec.fL, sc.spec.B, sc.spec.bL0
    bL = fL * z / (z - fL)
    v = (bL - bL0) / B
    if noise:
        v = v + noise * scene.Stream(seed, 77).normal(len(v))
    return sc, capi.InitArrays(v, fr, pt, w2c, P, fL), (v, bL)


def lstsq_reference(v, bL):
    a = np.stack([v, np.ones_like(v)], 1); b = bL.copy()
    bad = (v < 2) | (bL < 0)
    a[bad] = 0; b[bad] = 0
    x, _, rank, _ = np.linalg.lstsq(a, b, rcond=2 * np.finfo(float).eps)
    return x, int((~bad).sum()), rank


def test_noise_free_points_recover_the_generator(built):
    sc, arrs, (v, bL) = init_inputs(scene.SceneSpec(6, 80, None, 0x506, 501))
    r = oracle.init_plenoptic(arrs)
    assert r.rank == 2 and r.n_used == len(v)
    assert abs(r.B_init - sc.spec.B) < 1e-9 and abs(r.bL0_init - sc.spec.bL0) < 1e-8


def test_matches_svd_least_squares_with_noise_and_masked_rows(built):
    sc, arrs, (v, bL) = init_inputs(scene.SceneSpec(8, 120, None, 0x506, 502), seed=502, noise=0.05)
    arrs.vdepth[::7] = 1.5          # reference :485: rows with v < 2 are zeroed
    x, used, rank = lstsq_reference(arrs.vdepth, bL)
    r = oracle.init_plenoptic(arrs)
    assert r.n_used == used and r.rank == rank == 2
    assert abs(r.B_init - x[0]) <= 1e-10 * abs(x[0]) and abs(r.bL0_init - x[1]) <= 1e-10 * abs(x[1])


def test_rank_deficient_input_gives_the_minimum_norm_solution(built):
    sc, arrs, (v, bL) = init_inputs(scene.SceneSpec(4, 30, None, 0x506, 503))
    arrs.vdepth[:] = 2.5            # all rows identical in v: columns [2.5, 1] are parallel
    x, used, rank = lstsq_reference(arrs.vdepth, bL)
    r = oracle.init_plenoptic(arrs)
    assert rank == 1 and r.rank == 1 and r.n_used == used
    assert abs(r.B_init - x[0]) <= 1e-10 * abs(x[0]) and abs(r.bL0_init - x[1]) <= 1e-10 * abs(x[1])
    arrs.vdepth[:] = 1.0            # every row masked: zero matrix, zero solution
    r = oracle.init_plenoptic(arrs)
    assert r.rank == 0 and r.n_used == 0 and r.B_init == 0.0 and r.bL0_init == 0.0
