"""Parity of the GPU micro-lens maps / projectPointsToRawImage (include/lifcal_mla.h) with the CPU restatement
(oracle/lifcal_mla.cpp): bit-exact — lens list, both per-pixel maps, the epipolar web and the observation lists.

Reference: src/MicroLensGrid/MicroLensGrid.cpp:186-270, :338-421; src/CameraCalibration.cpp:521-632, :637-769.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GRIDS = {
    "plain": dict(width=640, height=480, lens_diameter=23.2, rotation=0.0, offset=(1.7, -3.2)),
    "rotated": dict(width=640, height=480, lens_diameter=23.2, rotation=0.004, offset=(-2.1, 0.6)),
    "rotated_neg": dict(width=701, height=533, lens_diameter=17.35, rotation=-0.021, offset=(5.25, 7.5), lens_base_y=(0.5, 0.86)),
    "no_rotation_on_grid": dict(width=512, height=384, lens_diameter=34.97, rotation=0.01, offset=(0.0, 0.0), rotation_on_grid=False),
    "raytrix_like": dict(width=2048, height=2048, lens_diameter=23.202295, rotation=0.00121, offset=(12.4, -7.9)),
    "small_lenses": dict(width=300, height=200, lens_diameter=7.5, rotation=0.3, offset=(0.5, 0.25)),
}


@pytest.fixture(scope="module", params=sorted(GRIDS))
def pair(request):
    from lifcal_amd.mla import MicroLensGrid
    from oracle.mla import MicroLensGrid as OracleGrid
    kw = GRIDS[request.param]
    g = MicroLensGrid(**kw)
    o = OracleGrid(**kw)
    yield request.param, g, o
    g.close()


def test_lens_list_is_identical(pair):
    _, g, o = pair
    assert g.n_lenses == o.n_lenses and g.n_lenses > 0
    for a, b in zip(g.lenses(), o.lenses()):
        assert np.array_equal(a, b)


def test_web_is_identical(pair):
    _, g, o = pair
    assert (g.n_web_groups, g.n_web_lines) == (o.n_web_groups, o.n_web_lines)
    for a, b in zip(g.web(), o.web()):
        assert np.array_equal(a, b)


def test_maps_are_identical(pair):
    _, g, o = pair
    ml, nxt = g.maps()
    oml, onxt = o.maps()
    assert np.array_equal(ml, oml), np.argwhere(ml != oml)[:5]
    assert np.array_equal(nxt, onxt), np.argwhere(nxt != onxt)[:5]


def _points(g, n, seed, border=False):
    rng = np.random.default_rng(seed)
    lo = -0.02 if border else 0.1
    x = rng.uniform(lo * g.width, (1 - lo) * g.width, n)
    y = rng.uniform(lo * g.height, (1 - lo) * g.height, n)
    vd = rng.uniform(1.2, 21.0, n) if border else rng.uniform(1.8, 9.0, n)
    return x, y, vd


def _same(a, b):
    assert len(a.u) == len(b.xR), (len(a.u), len(b.xR))
    assert np.array_equal(a.u, b.xR) and np.array_equal(a.v, b.yR)
    assert np.array_equal(a.mcx, b.cX) and np.array_equal(a.mcy, b.cY)
    assert np.array_equal(a.src.astype(np.int64), b.point)


@pytest.mark.parametrize("border", [False, True])
@pytest.mark.parametrize("scale", [1, 2])
def test_projection_is_identical(pair, border, scale):
    name, g, o = pair
    n = 3000 if g.width < 2000 else 20000
    x, y, vd = _points(g, n, 17 + scale, border)
    x /= scale; y /= scale
    a = g.projectPointsToRawImage(x, y, vd, scale)
    b = o.project_frame(x, y, vd, scale)
    assert len(a.u) > n // 4
    _same(a, b)


def test_out_of_contract_points_yield_nothing(pair):
    _, g, o = pair
    x = np.array([np.nan, 1e30, -1e30, -40.0, 100.0, 100.0, 100.0, 100.0, np.inf])
    y = np.array([100.0, 100.0, 100.0, 100.0, np.nan, -1e12, 100.0, 100.0, 100.0])
    vd = np.array([5.0, 5.0, 5.0, 5.0, 5.0, 5.0, np.nan, np.inf, 5.0])
    a = g.projectPointsToRawImage(x, y, vd, 1)
    b = o.project_frame(x, y, vd, 1)
    _same(a, b)
    assert len(a.u) == 0


def test_frames_concatenate_and_indices_are_gathered(pair):
    """all frames in one launch = the per-frame lists one after the other (reference loop over frames, :645)"""
    _, g, o = pair
    sizes = [700, 1, 0, 1300]
    xs, ys, vs, frs, pts = [], [], [], [], []
    for f, m in enumerate(sizes):
        x, y, vd = _points(g, m, 100 + f)
        xs.append(x); ys.append(y); vs.append(vd); frs.append(np.full(m, f, np.uint32)); pts.append(np.random.default_rng(f).integers(0, 50, m).astype(np.uint32))
    X, Y, V, FR, PT = map(np.concatenate, (xs, ys, vs, frs, pts))
    a = g.projectPointsToRawImage(X, Y, V, 1, fr=FR, pt=PT)
    start = 0; k = 0
    for f, m in enumerate(sizes):
        b = o.project_frame(xs[f], ys[f], vs[f], 1)
        sl = slice(k, k + len(b.xR))
        assert np.array_equal(a.u[sl], b.xR) and np.array_equal(a.mcy[sl], b.cY)
        assert np.array_equal(a.src[sl].astype(np.int64), b.point + start)
        assert np.all(a.fr[sl] == f) and np.array_equal(a.pt[sl], pts[f][b.point])
        start += m; k += len(b.xR)
    assert k == len(a.u)


def test_capacity_protocol_and_errors():
    from lifcal_amd import _capi as capi
    from lifcal_amd.mla import MicroLensGrid
    from lifcal_amd.bundle_adjustment import LifcalError
    lib = capi.load_library()
    g = MicroLensGrid(**GRIDS["plain"])
    x = np.array([300.0, 310.0]); y = np.array([200.0, 220.0]); vd = np.array([6.0, 4.0])
    full = g.projectPointsToRawImage(x, y, vd, 1)
    pts = capi.MlaPoints(2, capi.as_dptr(x), capi.as_dptr(y), capi.as_dptr(vd), None, None)
    bufs = [np.full(3, -1.0) for _ in range(4)]
    obs = capi.MlaObservations(3, 0, *[capi.as_dptr(b) for b in bufs], None, None, None)
    assert lib.lifcal_mla_project(g._h, 1, C.byref(pts), C.byref(obs)) == capi.MLA_MORE
    assert obs.n_obs == len(full.u) > 3 and np.all(bufs[0] == -1.0)          # nothing written
    obs = capi.MlaObservations(0, 0, None, None, None, None, None, None, None)
    assert lib.lifcal_mla_project(g._h, 0, C.byref(pts), C.byref(obs)) == -1     # scale < 1
    fr_out = np.zeros(64, np.uint32)
    big = [np.zeros(64) for _ in range(4)]
    obs = capi.MlaObservations(64, 0, *[capi.as_dptr(b) for b in big], None, capi.as_uptr(fr_out), None)
    assert lib.lifcal_mla_project(g._h, 1, C.byref(pts), C.byref(obs)) == -1     # fr wanted, not supplied
    g.close()
    with pytest.raises(LifcalError):
        MicroLensGrid(640, 480, 1.0)                                             # lens diameter must exceed the 1-pixel border twice
    with pytest.raises(LifcalError):
        MicroLensGrid(0, 480, 23.0)


def test_observations_feed_the_bundle_adjustment_layout():
    """the projection's output arrays are lifcal_ba_problem's u, v, mcx, mcy, fr, pt (include/lifcal_ba.h)"""
    from lifcal_amd.mla import MicroLensGrid
    g = MicroLensGrid(**GRIDS["plain"])
    x, y, vd = _points(g, 500, 3)
    fr = np.repeat(np.arange(5, dtype=np.uint32), 100); pt = np.tile(np.arange(100, dtype=np.uint32), 5)
    a = g.projectPointsToRawImage(x, y, vd, 1, fr=fr, pt=pt)
    assert a.u.dtype == np.float64 and a.fr.dtype == np.uint32 and len(a.fr) == len(a.u) == len(a.pt)
    assert np.all(np.diff(a.fr.astype(np.int64)) >= 0)                            # frame-major, as the reference pushes them
    # central projection (:748-749), in double from the float results
    assert np.allclose((a.u - a.mcx) * np.float32(vd[a.src]) + a.mcx, np.float32(x[a.src]), atol=2e-3)
    g.close()


def test_random_grids_and_points_stay_identical():
    """120 random lens grids (diameter 5..45 px, any rotation, offsets, squeezed lattices, rotation on or off the grid) with
    3000 random image points each, over the borders and across the whole virtual-depth range: every list bit for bit."""
    from lifcal_amd.mla import MicroLensGrid
    from oracle.mla import MicroLensGrid as OracleGrid
    rng = np.random.default_rng(20241022)
    n_obs = 0
    for case in range(120):
        kw = dict(width=int(rng.integers(48, 420)), height=int(rng.integers(48, 420)), lens_diameter=float(rng.uniform(5.0, 45.0)),
                  lens_base_y=(0.5, float(rng.uniform(0.80, 0.92))), rotation=float(rng.uniform(-0.6, 0.6)) if case % 3 else float(rng.uniform(-0.01, 0.01)),
                  offset=(float(rng.uniform(-25, 25)), float(rng.uniform(-25, 25))), rotation_on_grid=bool(case % 4))
        g = MicroLensGrid(**kw); o = OracleGrid(**kw)
        for a, b in zip(g.lenses() + g.web() + g.maps(), o.lenses() + o.web() + o.maps()):
            assert np.array_equal(a, b), (case, kw)
        scale = int(rng.integers(1, 4))
        n = 3000
        x = rng.uniform(-0.05 * g.width, 1.05 * g.width, n) / scale
        y = rng.uniform(-0.05 * g.height, 1.05 * g.height, n) / scale
        vd = rng.uniform(1.5, 21.0, n)
        a = g.projectPointsToRawImage(x, y, vd, scale)
        b = o.project_frame(x, y, vd, scale)
        _same(a, b)
        n_obs += len(a.u)
        g.close()
    assert n_obs > 500_000
