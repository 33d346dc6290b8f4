"""The oracle's residual + autodiff Jacobian against an independent 40-digit restatement (tests/mp_model.py),
central finite differences, and the structural facts of SURVEY.md §8 (dead columns, sign folding)."""
import numpy as np
import pytest

import oracle
from lifcal_amd import scene
from tests import mp_model
from tests.helpers import S

CONFIGS = [0x506, 0xF06, 0xD05, 0x500, 0xD00, 0x504, 0xD02, 0x501]


@pytest.fixture(scope="module")
def sc():
    return scene.make_scene(S(6, 40, None, 0x506, 301))


@pytest.mark.parametrize("config", CONFIGS)
def test_oracle_matches_arbitrary_precision(sc, config):
    n_rad = config & 3
    tan = bool(config & 4)
    cam = sc.cam_gt.copy()
    cam[5:] = 0.0
    for i in range(n_rad):
        cam[5 + i] = sc.spec.k[i]
    if tan:
        cam[5 + n_rad], cam[6 + n_rad] = sc.spec.p
    cam[:5] *= 1.003  # away from the generating point so residuals are not tiny
    live = 5 + n_rad + (2 if tan else 0)
    for i in (3, 400, 1100):
        f, p = sc.fr[i], sc.pt[i]
        view = sc.views0[6 * f:6 * f + 6]
        P = sc.pts0[3 * p:3 * p + 3]
        r, J = oracle.residual_block(config, 3, cam, view, P, sc.u[i], sc.v[i], sc.mcx[i], sc.mcy[i], sc.spx, sc.scale)
        x = np.concatenate([cam, view, P])
        cols = list(range(live)) + list(range(17, 26))
        rm, Jm = mp_model.residual_and_jacobian(x, config, sc.u[i], sc.v[i], sc.mcx[i], sc.mcy[i], sc.spx, sc.scale, cols)
        assert abs(r[0] - float(rm[0])) <= 1e-10 * max(1.0, abs(r[0]))
        assert abs(r[1] - float(rm[1])) <= 1e-10 * max(1.0, abs(r[1]))
        for a in range(2):
            row_scale = max(abs(float(Jm[(a, c)])) for c in cols)
            for c in cols:
                # column-wise relative agreement (tolerance 1e-11 of the column value or 1e-13 of the row scale)
                ref = float(Jm[(a, c)])
                assert abs(J[a, c] - ref) <= 1e-11 * abs(ref) + 1e-13 * row_scale, (hex(config), i, a, c, J[a, c], ref)
        # structurally dead camera slots: exactly zero
        assert np.all(J[:, live:17] == 0.0)


def test_sign_folding_flips_derivative(sc):
    """reference BundleAdjustment.h:123-133: |camera[0..2]| and |c_raw| fold value and derivative sign together."""
    i = 10
    f, p = sc.fr[i], sc.pt[i]
    view, P = sc.views_gt[6 * f:6 * f + 6], sc.pts_gt[3 * p:3 * p + 3]
    cam = sc.cam_gt.copy()
    r0, J0 = oracle.residual_block(0x506, 3, cam, view, P, sc.u[i], sc.v[i], sc.mcx[i], sc.mcy[i], sc.spx, sc.scale)
    for k in range(3):
        c2 = cam.copy()
        c2[k] = -c2[k]
        r1, J1 = oracle.residual_block(0x506, 3, c2, view, P, sc.u[i], sc.v[i], sc.mcx[i], sc.mcy[i], sc.spx, sc.scale)
        assert np.array_equal(r0, r1)
        assert np.array_equal(J1[:, k], -J0[:, k])
        mask = np.ones(26, bool); mask[k] = False
        assert np.array_equal(J1[:, mask], J0[:, mask])


def test_arities_agree(sc):
    """<2,17,6,3>, <2,17,6> and <2,17> (reference Create(), :199-222) give the same residual and shared columns."""
    i = 77
    f, p = sc.fr[i], sc.pt[i]
    view, P = sc.views0[6 * f:6 * f + 6], sc.pts0[3 * p:3 * p + 3]
    args = (sc.u[i], sc.v[i], sc.mcx[i], sc.mcy[i], sc.spx, sc.scale)
    r3, J3 = oracle.residual_block(0xF06, 3, sc.cam0 + 1e-3, view, P, *args)
    r2, J2 = oracle.residual_block(0xF06, 2, sc.cam0 + 1e-3, view, P, *args)
    r1, J1 = oracle.residual_block(0xF06, 1, sc.cam0 + 1e-3, view, P, *args)
    assert np.allclose(r3, r2, rtol=0, atol=1e-12) and np.allclose(r3, r1, rtol=0, atol=1e-9)
    assert np.allclose(J3[:, :23], J2[:, :23], rtol=1e-13, atol=0)
    assert np.all(J2[:, 23:] == 0) and np.all(J1[:, 17:] == 0)
    assert np.allclose(J3[:, :17], J1[:, :17], rtol=1e-9, atol=1e-9 * np.abs(J3).max())


def test_projection_matches_scene_generator(sc):
    """scene.project (numpy, used only to synthesise data) and the oracle agree on the forward model."""
    for config in (0x506, 0xF06):
        idx = np.arange(0, sc.n_obs, 97)
        R = scene.euler_xyz(sc.views_gt.reshape(-1, 6)[:, :3])
        pc = np.einsum("nij,nj->ni", R[sc.fr[idx]], sc.pts_gt.reshape(-1, 3)[sc.pt[idx]]) + sc.views_gt.reshape(-1, 6)[sc.fr[idx], 3:]
        ml = np.stack([sc.mcx[idx], sc.mcy[idx]], -1)
        got = scene.project(pc, ml, sc.cam_gt, config, sc.spx, sc.scale)
        sp = sc.spx / sc.scale
        craw = np.abs((sc.cam_gt[3:5] + 0.5) * sc.scale - 0.5)
        for n, i in enumerate(idx):
            exp = oracle.project_point(pc[n], sp, sp, sc.cam_gt[0], sc.cam_gt[1], sc.cam_gt[2], craw, ml[n],
                                       sc.cam_gt[5:7], sc.cam_gt[7:9], bool(config & 0x800))
            assert np.allclose(got[n], exp, rtol=0, atol=1e-9)


def test_rigid_transform_is_euler_xyz():
    view = np.array([0.11, -0.07, 0.23, 1.0, -2.0, 3.0])
    RT = oracle.rigid_transform(view)
    assert np.allclose(RT[:, :3], scene.euler_xyz(view[:3]), rtol=0, atol=1e-15)
    assert np.array_equal(RT[:, 3], view[3:])


def test_constraint_block():
    p1 = np.array([1.0, 2.0, 3.0]); p2 = np.array([-2.0, 0.5, 7.0])
    r, J = oracle.constraint_block(p1, p2, 4.0, 0.5)
    d = p1 - p2; n = np.linalg.norm(d)
    assert abs(r - (n - 4.0) / (0.5 + 1e-6)) < 1e-14
    assert np.allclose(J[:3], d / n / (0.5 + 1e-6), rtol=1e-14) and np.allclose(J[3:], -J[:3], rtol=1e-14)
