"""The oracle's Schur-reduced system against the full dense normal equations assembled with numpy from
per-block autodiff Jacobians, including robust loss, constraints and fixed columns."""
import numpy as np
import pytest

import oracle
from lifcal_amd import _capi as capi, scene
from tests.helpers import S, problem


def dense_system(sc, pa, radius, loss_scale=0.5):
    F, P, N = sc.spec.n_frames, sc.spec.n_points, sc.n_obs
    robust = bool(sc.config & 0x200)
    ncol = 17 + 6 * F + 3 * P
    rows, res = [], []
    cost = 0.0
    for i in range(N):
        f, p = sc.fr[i], sc.pt[i]
        r, J = oracle.residual_block(sc.config, 3, pa.cam, pa.views[6 * f:6 * f + 6], pa.pts[3 * p:3 * p + 3],
                                     sc.u[i], sc.v[i], sc.mcx[i], sc.mcy[i], sc.spx, sc.scale)
        s = float(r @ r)
        if robust:
            b = loss_scale ** 2
            cost += 0.5 * b * np.log1p(s / b)
            w = np.sqrt(1.0 / (1.0 + s / b))
            r, J = r * w, J * w
        else:
            cost += 0.5 * s
        for a in range(2):
            row = np.zeros(ncol)
            row[:17] = J[a, :17]; row[17 + 6 * f:23 + 6 * f] = J[a, 17:23]; row[17 + 6 * F + 3 * p:20 + 6 * F + 3 * p] = J[a, 23:]
            rows.append(row); res.append(r[a])
    if pa.struct.use_constraints:
        for c in range(len(sc.c_i)):
            i, j = int(sc.c_i[c]), int(sc.c_j[c])
            r, J = oracle.constraint_block(pa.pts[3 * i:3 * i + 3], pa.pts[3 * j:3 * j + 3], sc.c_dist[c], sc.c_sigma[c])
            row = np.zeros(ncol); row[17 + 6 * F + 3 * i:20 + 6 * F + 3 * i] = J[:3]; row[17 + 6 * F + 3 * j:20 + 6 * F + 3 * j] = J[3:]
            rows.append(row); res.append(r); cost += 0.5 * r * r
    J = np.array(rows); r = np.array(res)
    for k in range(17):
        if (pa.struct.fixed_mask >> k) & 1:
            J[:, k] = 0.0
    H = J.T @ J
    g = J.T @ r
    h = np.diag(H).copy()
    sig = 1.0 / (1.0 + np.sqrt(h))
    lam = np.clip(h * sig * sig, 1e-6, 1e32) / (radius * sig * sig)
    live = h > 0
    Hd = H + np.diag(np.where(live, lam, 1.0))
    return cost, H, g, np.linalg.solve(Hd, -g), live


@pytest.mark.parametrize("spec", [
    S(5, 25, None, 0x506, 401),
    S(5, 25, None, 0xF06, 402, outlier_fraction=0.05),
    S(5, 25, None, 0x506, 403, n_constraints=3),
    S(6, 30, None, 0xF06, 404, recalib=True),
], ids=["plain", "robust_adj", "constraints", "recalib_fixed"])
def test_reduced_system_equals_full_normal_equations(spec):
    sc = scene.make_scene(spec)
    pa = problem(sc)
    F, P = spec.n_frames, spec.n_points
    radius = 3e3
    cost, H, g, delta, live = dense_system(sc, pa, radius)
    ref = oracle.sweep(pa, radius=radius)
    assert ref.rc == 0
    assert abs(ref.cost - cost) <= 1e-12 * cost
    x = np.linalg.solve(ref.S, ref.rhs)
    # camera + poses: canonical ordering = dense ordering for the first 17 + 6F unknowns
    nb = 17 + 6 * F
    scale = np.abs(delta[:nb]).max()
    assert np.abs(x[:nb] - delta[:nb]).max() <= 1e-7 * scale
    # promoted points (constraints): ascending point id after the poses
    prom = sorted(set(int(j) for j in sc.c_j)) if pa.struct.use_constraints and len(sc.c_j) else []
    assert ref.n_promoted == len(prom)
    for k, q in enumerate(prom):
        assert np.allclose(x[nb + 3 * k:nb + 3 * k + 3], delta[nb + 3 * q:nb + 3 * q + 3], rtol=1e-6, atol=1e-9 * scale)
    # gradient of the reduced block and of the points
    assert np.allclose(ref.gradient_reduced[:nb], g[:nb], rtol=1e-10, atol=1e-12 * np.abs(g).max())
    assert np.allclose(ref.point_gradient, g[nb:], rtol=1e-10, atol=1e-12 * np.abs(g).max())
    assert abs(ref.gradient_max_norm - np.abs(g).max()) <= 1e-12 * np.abs(g).max()
    # back-substitution: delta_P = -Uinv (g_P + W delta_B) reproduces the dense solution for eliminated points
    W = H[nb:, :nb]
    for q in range(P):
        if q in prom or not live[nb + 3 * q]:
            continue
        Ui = ref.point_hessian_inv[9 * q:9 * q + 9].reshape(3, 3)
        rhs_p = g[nb + 3 * q:nb + 3 * q + 3] + W[3 * q:3 * q + 3] @ delta[:nb]
        for qq in prom:
            rhs_p = rhs_p + H[nb + 3 * q:nb + 3 * q + 3, nb + 3 * qq:nb + 3 * qq + 3] @ delta[nb + 3 * qq:nb + 3 * qq + 3]
        assert np.allclose(-Ui @ rhs_p, delta[nb + 3 * q:nb + 3 * q + 3], rtol=1e-6, atol=1e-9 * scale)


def test_fixed_columns_are_identity_rows():
    sc = scene.make_scene(S(6, 30, None, 0xF06, 405, recalib=True))
    ref = oracle.sweep(problem(sc), radius=1e4)
    for k in (0, 2, 9, 16):   # fL and B fixed (reference :935-936); slots >= 9 are structurally dead
        e = np.zeros(ref.n_reduced); e[k] = 1.0
        assert np.array_equal(ref.S[k], e) and np.array_equal(ref.S[:, k], e) and ref.rhs[k] == 0.0
