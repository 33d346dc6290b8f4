"""The three rows built so far, chained on the GPU as LiFCal's calibration loop chains them (src/CameraCalibration.cpp):
projectPointsToRawImage (:637-769) makes the micro-image observations from virtual-image points, initPlenopticParameters
(:456-499) gives B and bL0 their start values, performBundleAdjustment (:774-992) refines everything.  The same chain on the
CPU restatements gives the same calibration."""
import dataclasses

import numpy as np
import pytest

import oracle
from oracle.mla import MicroLensGrid as OracleGrid
from lifcal_amd import BundleAdjustment, MicroLensGrid, initPlenopticParameters, _capi as capi, scene

pytestmark = pytest.mark.gpu


def _chain(sc, grid_cls, project, init, solve):
    sp = sc.spec
    g = grid_cls(sp.raw_width, sp.raw_height, sp.lens_diameter, sp.lens_base_y, sp.grid_rotation, sp.grid_offset)
    obs = project(g, sc)
    F = sp.n_frames
    R = scene.euler_xyz(sc.views0.reshape(-1, 6)[:, :3])
    w2c = np.zeros((F, 4, 4)); w2c[:, :3, :3] = R; w2c[:, :3, 3] = sc.views0.reshape(-1, 6)[:, 3:]; w2c[:, 3, 3] = 1.0
    B0, bL00 = init(sc, w2c)
    cam0 = sc.cam0.copy(); cam0[1] = bL00; cam0[2] = B0
    pa = capi.ProblemArrays(obs["u"], obs["v"], obs["mcx"], obs["mcy"], obs["pt"], obs["fr"], cam0, sc.views0, sc.pts0, sc.spx, sc.scale, sc.config)
    summary, stats = solve(pa)
    return obs, (B0, bL00), pa, summary, stats


def _gpu_project(g, sc):
    o = g.projectPointsToRawImage(sc.img_x, sc.img_y, sc.img_vd, sc.spec.scale, fr=sc.img_fr, pt=sc.img_pt)
    return dict(u=o.u, v=o.v, mcx=o.mcx, mcy=o.mcy, fr=o.fr, pt=o.pt)


def _cpu_project(g, sc):
    parts = []
    for f in range(sc.spec.n_frames):
        m = np.flatnonzero(sc.img_fr == f)
        o = g.project_frame(sc.img_x[m], sc.img_y[m], sc.img_vd[m], sc.spec.scale)
        parts.append((o.xR, o.yR, o.cX, o.cY, np.full(len(o.xR), f, np.uint32), sc.img_pt[m][o.point]))
    u, v, mcx, mcy, fr, pt = (np.concatenate(c) for c in zip(*parts))
    return dict(u=u, v=v, mcx=mcx, mcy=mcy, fr=fr, pt=pt)


def _gpu_init(sc, w2c):
    r = initPlenopticParameters(sc.img_vd, sc.img_fr, sc.img_pt, w2c, sc.pts0.reshape(-1, 3), sc.cam0[0])
    return r.B_init, r.bL0_init


def _cpu_init(sc, w2c):
    arrs = capi.InitArrays(sc.img_vd, sc.img_fr, sc.img_pt, w2c, sc.pts0.reshape(-1, 3), sc.cam0[0])
    r = oracle.init_plenoptic(arrs)
    return r.B_init, r.bL0_init


def _gpu_solve(pa):
    ba = BundleAdjustment(pa)
    s = ba.performBundleAdjustment()
    st = ba.calcReprojectionError(1.0)
    ba.close()
    return s, st


def _cpu_solve(pa):
    s = oracle.solve(pa)
    return s, oracle.reproj_stats(pa, 1.0)


@pytest.mark.parametrize("config", [0x500, 0x506])
def test_chain_calibrates_and_agrees_with_the_cpu_chain(built, config):
    # distortion-free generator camera: the central projection of projectPointsToRawImage is then the exact forward model
    sp = dataclasses.replace(scene.SceneSpec(8, 90, None, config, 9100 + config), k=(0.0, 0.0), p=(0.0, 0.0), noise_px=0.0)
    sc = scene.make_scene(sp)
    obs, init_g, pa_g, s_g, st_g = _chain(sc, MicroLensGrid, _gpu_project, _gpu_init, _gpu_solve)
    obs_c, init_c, pa_c, s_c, st_c = _chain(sc, OracleGrid, _cpu_project, _cpu_init, _cpu_solve)
    for k in obs:                                                   # row f1: identical observation lists
        assert np.array_equal(obs[k], obs_c[k]), k
    assert len(obs["u"]) > 3000
    assert init_g == pytest.approx(init_c, rel=1e-11)               # row f2
    # the hot path: both chains end at the same calibration (1e-6 relative, BASELINE north_star) ...
    assert s_g.final_cost == pytest.approx(s_c.final_cost, rel=1e-5, abs=1e-9)
    assert np.allclose(pa_g.cam[:5], pa_c.cam[:5], rtol=1e-6)
    # ... which explains the observations down to the float rounding of the micro-image coordinates
    assert max(st_g.std_x, st_g.std_y) < 2e-3
    assert st_g.num_inliers == st_g.num_points == len(obs["u"])
    # gauge-free quantities of the generator are recovered: B and bL0 enter only through the virtual depth
    assert pa_g.cam[2] / sc.cam_gt[2] == pytest.approx(1.0, abs=0.05)
