"""Properties of the CPU restatement of the micro-lens grid / projectPointsToRawImage step (oracle/lifcal_mla.cpp).

The reference holds no fixtures for this step, so the restatement is checked against what the reference's
comments and geometry promise: a hexagonal lattice of lens centres, maps that name the lens a pixel belongs to / is nearest
to, a web of lattice base lines sorted by length, and observations that are the central projection of the virtual-image
point through each lens centre (src/CameraCalibration.cpp:748-749).
"""
import numpy as np
import pytest

from oracle.mla import MicroLensGrid

D = 23.2


@pytest.fixture(scope="module")
def grid():
    return MicroLensGrid(640, 480, D, rotation=0.0, offset=(1.7, -3.2))


@pytest.fixture(scope="module")
def grid_rot():
    return MicroLensGrid(640, 480, D, rotation=0.004, offset=(-2.1, 0.6))


def test_lens_centres_form_a_hexagonal_lattice(grid):
    cx, cy, t = grid.lenses()
    assert grid.n_lenses == len(cx) > 500
    c = np.stack([cx, cy], 1).astype(np.float64)
    inside = (cx > 3 * D) & (cx < 640 - 3 * D) & (cy > 3 * D) & (cy < 480 - 3 * D)
    d2 = ((c[inside, None, :] - c[None, :, :]) ** 2).sum(-1)
    d2.sort(axis=1)
    assert np.allclose(d2[:, 0], 0)
    assert np.allclose(np.sqrt(d2[:, 1:7]), D, atol=2e-3)               # six neighbours one diameter away
    assert np.all(np.sqrt(d2[:, 7]) > 1.7 * D)
    assert len(np.unique(np.round(c, 2), axis=0)) == len(c)              # no lens twice
    # the image is covered: centres up to half a diameter beyond each border are part of the grid (MicroLensGrid.cpp:189-194)
    assert cx.min() < 0.5 * D and cx.max() > 639 - 0.5 * D and cy.min() < 0.5 * D and cy.max() > 479 - 0.5 * D
    assert cx.min() >= -0.5 * D - 0.5 * D and cx.max() <= 639 + D and cy.min() >= -D and cy.max() <= 479 + D
    assert set(np.unique(t)) == {0, 1, 2}


def test_lens_types_differ_between_neighbours(grid):
    cx, cy, t = grid.lenses()
    c = np.stack([cx, cy], 1).astype(np.float64)
    inside = np.flatnonzero((cx > 2 * D) & (cx < 640 - 2 * D) & (cy > 2 * D) & (cy < 480 - 2 * D))
    d2 = ((c[inside, None, :] - c[None, :, :]) ** 2).sum(-1)
    nb = np.argsort(d2, axis=1)[:, 1:7]
    # along a lattice row the type advances by one per lens (x % 3, MicroLensGrid.cpp:231): the two row neighbours differ
    same_row = np.abs(c[nb, 1] - c[inside, None, 1]) < 1e-3
    assert np.all(same_row.sum(1) == 2)
    assert np.all(t[nb][same_row].reshape(-1, 2) != t[inside, None])


def test_rotation_turns_the_lattice_about_the_offset(grid_rot):
    cx, cy, _ = grid_rot.lenses()
    # row direction: neighbours one diameter apart along (cos a, -sin a) in image coordinates (y down)
    c = np.stack([cx, cy], 1).astype(np.float64)
    i = int(np.argmin((cx - 320) ** 2 + (cy - 240) ** 2))
    d = c - c[i]
    row = d[(np.abs(np.hypot(d[:, 0], d[:, 1]) - D) < 1e-2) & (d[:, 0] > 0.9 * D)]
    assert len(row) == 1
    assert np.arctan2(-row[0, 1], row[0, 0]) == pytest.approx(0.004, abs=2e-4)


def test_maps(grid):
    cx, cy, _ = grid.lenses()
    ml, nxt = grid.maps()
    r = np.float32(D) * np.float32(0.5) - np.float32(1.0)
    yy, xx = np.mgrid[0:480, 0:640]
    own = ml >= 0
    assert 0.6 < own.mean() < 0.8                                         # pi (d/2-1)^2 / (sqrt(3)/2 d^2) = 0.757
    dist_own = np.hypot(xx[own] - cx[ml[own]], yy[own] - cy[ml[own]])
    assert np.all(dist_own <= r + 1e-4)
    assert np.all(nxt >= 0)
    assert np.array_equal(nxt[own], ml[own])
    # every pixel within the validity radius of a lens centre belongs to that lens
    c = np.stack([cx, cy], 1).astype(np.float64)
    pix = np.stack([xx.ravel(), yy.ravel()], 1).astype(np.float64)
    step = 7                                                              # a sample keeps the brute force small
    pix = pix[::step]
    d = np.hypot(pix[:, None, 0] - c[None, :, 0], pix[:, None, 1] - c[None, :, 1])
    near = d.argmin(1)
    dn = d[np.arange(len(pix)), near]
    flat_ml = ml.ravel()[::step]; flat_nx = nxt.ravel()[::step]
    sure_in = dn < r - 1e-3
    assert np.array_equal(flat_ml[sure_in], near[sure_in])
    sure_out = dn > r + 1e-3
    assert np.all(flat_ml[sure_out] == -1)
    # border pixels point at a lens with a valid pixel in the ring where the search first hit one: never farther than
    # the true nearest such lens by more than the ring geometry allows
    dn_next = d[np.arange(len(pix)), flat_nx]
    assert np.all(dn_next <= D / np.sqrt(3) + 2.0)


def test_web_is_sorted_unique_lattice_vectors(grid):
    dist, ex, ey, grp = grid.web()
    assert grid.n_web_lines == len(dist)
    assert np.all(np.diff(dist) >= -1e-9) and np.all(np.diff(grp) >= 0) and grp[0] == 0 and grp[-1] == grid.n_web_groups - 1
    assert np.allclose(np.hypot(ex, ey), 1.0, atol=1e-12)
    assert dist.max() <= 10 * D * (1 + 1e-6)
    for g in range(grid.n_web_groups):
        assert np.ptp(dist[grp == g].astype(np.float32)) == 0
    # lattice coordinates (a, b): v = a (1,0) d + b (1/2, sqrt(3)/2) d
    v = np.stack([ex * dist, ey * dist], 1) / D
    b = v[:, 1] / np.sqrt(0.75)
    a = v[:, 0] - 0.5 * b
    assert np.allclose(a, np.round(a), atol=1e-5) and np.allclose(b, np.round(b), atol=1e-5)
    ab = np.stack([np.round(a), np.round(b)], 1).astype(int)
    assert len(np.unique(ab, axis=0)) == len(ab)
    s = set(map(tuple, ab))
    assert not any((-p, -q) in s for p, q in s)                           # one of each +-pair ("-pi/2 < phi <= pi/2")
    # all lattice vectors up to ten diameters, one per pair; the three of exactly ten diameters stand or fall with the
    # rounding of `baseLineDist > maxDist` (double against float, CameraCalibration.cpp:605)
    n_in = n_edge = 0
    for p in range(-12, 13):
        for q in range(-12, 13):
            r = np.hypot(p + 0.5 * q, np.sqrt(0.75) * q)
            if (p, q) != (0, 0) and r <= 10 + 1e-9:
                n_in += r < 10 - 1e-9
                n_edge += r >= 10 - 1e-9
    assert n_edge == 6 and n_in // 2 <= len(ab) <= (n_in + n_edge) // 2


def test_rotated_web_keeps_both_vertical_lines(grid_rot):
    """With rotation on the grid the test `epiLine[1] == -1.0f` (CameraCalibration.cpp:605) no longer matches the rotated
    vertical line, so both (0, k sqrt(3) d) and its negative stay in the web: the reference then lists those lenses twice."""
    dist, ex, ey, _ = grid_rot.web()
    v = np.round(np.stack([ex * dist, ey * dist], 1), 3)
    s = set(map(tuple, v))
    opp = [p for p in s if (-p[0], -p[1]) in s]
    assert len(opp) == 2 * 5                                               # k = 1..5: k sqrt(3) <= 10


def _brute(grid, px, py, vd, margin):
    """lenses the reference must (margin < 0) / may (margin > 0) report for one point, given the nearest lens from the map"""
    if not (2.0 < np.float32(vd) < 20.0):
        return set()
    cx, cy, _ = grid.lenses()
    _, nxt = grid.maps()
    xi = min(int(np.float32(px) + np.float32(0.5)), grid.width - 1); yi = min(int(np.float32(py) + np.float32(0.5)), grid.height - 1)
    n0 = nxt[yi, xi]
    radius = D * 0.5 * vd + 2.0
    dp = np.hypot(cx - px, cy - py)
    dn = np.hypot(cx - cx[n0], cy - cy[n0])
    valid = D * 0.5 - 1.0
    ok = (dp <= radius + margin) & (dn <= min(radius, 10 * D) + margin) & (dp / vd < valid + margin / vd)
    if dp[n0] > radius - margin and margin < 0:
        return set()
    xr = (px - cx) / vd + cx; yr = (py - cy) / vd + cy
    ok &= (xr >= -margin) & (xr <= grid.width - 1 + margin) & (yr >= -margin) & (yr <= grid.height - 1 + margin)
    ok &= (cx > 1) & (cx < grid.width - 2) & (cy > 1) & (cy < grid.height - 2) if margin < 0 else True
    return set(np.flatnonzero(ok))


def test_projection_against_brute_force(grid):
    rng = np.random.default_rng(5)
    n = 400
    # the search radius d/2 v + 2 stays inside the image: at the borders the reference clamps the predicted lens centre into
    # the image (:729-732) and may then list a border lens more than once
    px = rng.uniform(115, 525, n); py = rng.uniform(115, 365, n); vd = rng.uniform(1.5, 7.0, n)
    vd[:5] = [1.0, 2.0, 20.0, 25.0, np.nan]                               # outside (2, 20): no observation (:655)
    o = grid.project_frame(px, py, vd, 1)
    assert not np.isin(o.point, [0, 1, 2, 3, 4]).any()
    assert np.all(np.diff(o.point) >= 0)                                  # frame order is point order
    cx, cy, _ = grid.lenses()
    key = {(np.float32(a).item(), np.float32(b).item()): i for i, (a, b) in enumerate(zip(cx, cy))}
    found = [set() for _ in range(n)]
    for k in range(len(o.point)):
        li = key[(o.cX[k], o.cY[k])]
        assert li not in found[o.point[k]]                                # no duplicates without rotation
        found[o.point[k]].add(li)
        # central projection through the lens centre: (xR - c) vd + c = upsampled point (:748-749)
        assert (o.xR[k] - o.cX[k]) * np.float32(vd[o.point[k]]) + o.cX[k] == pytest.approx(px[o.point[k]], abs=2e-3)
        assert (o.yR[k] - o.cY[k]) * np.float32(vd[o.point[k]]) + o.cY[k] == pytest.approx(py[o.point[k]], abs=2e-3)
        assert np.hypot(o.xR[k] - o.cX[k], o.yR[k] - o.cY[k]) < D * 0.5 - 1.0
    checked = 0
    for p in range(5, n):
        must = _brute(grid, px[p], py[p], vd[p], -2e-2)
        may = _brute(grid, px[p], py[p], vd[p], +2e-2)
        assert must <= found[p] <= may, (p, must - found[p], found[p] - may)
        checked += len(found[p])
    assert checked > 2000
    # the number of micro images a point shows up in grows with the virtual depth (about pi/ (2 sqrt 3) v^2 ... )
    cnt = np.array([len(f) for f in found])
    assert cnt[vd > 6].mean() > 3 * cnt[(vd > 2) & (vd < 3)].mean()


def test_upsampling_scale(grid):
    """depth_to_raw_im_scale s maps virtual-image pixel centres to raw pixel centres: s (x + 1/2) - 1/2 (:665-666)"""
    rng = np.random.default_rng(6)
    px = rng.uniform(40, 280, 50); py = rng.uniform(40, 200, 50); vd = rng.uniform(3, 6, 50)
    a = grid.project_frame(px, py, vd, 2)
    b = grid.project_frame(2 * (px + 0.5) - 0.5, 2 * (py + 0.5) - 0.5, vd, 1)
    assert len(a.xR) == len(b.xR) > 0
    assert np.array_equal(a.point, b.point) and np.array_equal(a.cX, b.cX)
    assert np.allclose(a.xR, b.xR, atol=1e-3) and np.allclose(a.yR, b.yR, atol=1e-3)


def test_rotated_grid_reports_vertical_neighbours_twice(grid_rot):
    px = np.array([320.3]); py = np.array([241.1]); vd = np.array([8.0])
    o = grid_rot.project_frame(px, py, vd, 1)
    c = np.round(np.stack([o.cX, o.cY], 1), 3)
    u, cnt = np.unique(c, axis=0, return_counts=True)
    assert cnt.max() == 2 and (cnt == 2).sum() >= 2 and len(u) > 30


def test_capacity_protocol(grid):
    import ctypes as C
    from oracle import lib
    px = np.array([300.0]); py = np.array([200.0]); vd = np.array([6.0])
    full = grid.project_frame(px, py, vd, 1)
    dp = C.POINTER(C.c_double)
    out = [np.zeros(2) for _ in range(4)]; pt = np.zeros(2, np.int64)
    m = lib().lo_mla_project_frame(grid._h, 1, 1, px.ctypes.data_as(dp), py.ctypes.data_as(dp), vd.ctypes.data_as(dp), 2,
                                   *[a.ctypes.data_as(dp) for a in out], pt.ctypes.data_as(C.POINTER(C.c_int64)))
    assert m == -len(full.xR)
    assert np.array_equal(out[0], full.xR[:2])
