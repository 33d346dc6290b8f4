"""SURVEY.md 8f rank f4 inside the GPU chain: a COLMAP sparse model on disk -> ColmapModel (include/lifcal_colmap.h; reference
CalibrationData::readDataFromFirstCalibration + getCalibDataCV, src/CalibrationData/CalibrationData.cpp:56-127, :492-538) ->
projectPointsToRawImage (:637-769) -> initPlenopticParameters (:456-499) -> performBundleAdjustment (:774-992), against the same
chain on the CPU restatements (oracle/colmap.py reads the same files).  The model files are written here from a synthetic scene in
COLMAP's published binary layout (cameras.bin / images.bin / points3D.bin): poses as unit quaternion + translation, the virtual-image
points of every frame as its 2D points, sparse unordered point ids, a few outlier 2D points.  Virtual depths are not part of a
COLMAP model (the reference reads them from depth maps, out of scope): they travel beside the model, keyed by (frame, point)."""
import dataclasses
import os
import struct

import numpy as np
import pytest

import oracle
from oracle import colmap as oc
from oracle.mla import MicroLensGrid as OracleGrid
from lifcal_amd import BundleAdjustment, MicroLensGrid, initPlenopticParameters, _capi as capi, scene
from lifcal_amd.colmap import ColmapModel

pytestmark = pytest.mark.gpu
INVALID = 2 ** 64 - 1


def _quat_xyz(a):
    """unit quaternion (w, x, y, z) of Rx(a0) Ry(a1) Rz(a2) (reference CameraModel.h:251-254)"""
    h = 0.5 * np.asarray(a)
    qx = np.array([np.cos(h[0]), np.sin(h[0]), 0, 0]); qy = np.array([np.cos(h[1]), 0, np.sin(h[1]), 0]); qz = np.array([np.cos(h[2]), 0, 0, np.sin(h[2])])

    def mul(p, q):
        return np.array([p[0] * q[0] - p[1] * q[1] - p[2] * q[2] - p[3] * q[3], p[0] * q[1] + p[1] * q[0] + p[2] * q[3] - p[3] * q[2],
                         p[0] * q[2] + p[2] * q[0] + p[3] * q[1] - p[1] * q[3], p[0] * q[3] + p[3] * q[0] + p[1] * q[2] - p[2] * q[1]])
    return mul(mul(qx, qy), qz)


def write_colmap_model(folder, sc, point_ids, image_ids):
    """the scene's start values as a COLMAP model (binary layout)"""
    os.makedirs(folder, exist_ok=True)
    views = sc.views0.reshape(-1, 6); pts = sc.pts0.reshape(-1, 3)
    rs = np.random.default_rng(7)
    with open(os.path.join(folder, "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", 1))
        params = [3181.25, 3184.75, 511.3, 513.9, 0.0, 0.0, 0.0, 0.0]   # OPENCV: fx fy cx cy k1 k2 p1 p2
        f.write(struct.pack("<IiQQ", 1, 4, 1024, 1024)); f.write(struct.pack("<8d", *params))
    with open(os.path.join(folder, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(image_ids)))
        for k in rs.permutation(len(image_ids)):                         # file order is not id order
            q = _quat_xyz(views[k, :3]); t = views[k, 3:]
            f.write(struct.pack("<I4d3dI", int(image_ids[k]), *q, *t, 1)); f.write(f"frame_{image_ids[k]:04d}.png".encode() + b"\x00")
            m = np.flatnonzero(sc.img_fr == k)
            p2 = [(float(sc.img_x[i]), float(sc.img_y[i]), int(point_ids[sc.img_pt[i]])) for i in m]
            p2.insert(len(p2) // 2, (12.5, 800.25, INVALID))             # an outlier without a 3D point
            f.write(struct.pack("<Q", len(p2)))
            for x, y, pid in p2:
                f.write(struct.pack("<ddQ", x, y, pid))
    with open(os.path.join(folder, "points3D.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(point_ids)))
        for j in rs.permutation(len(point_ids)):
            f.write(struct.pack("<Q3d3BdQ", int(point_ids[j]), *pts[j], 10, 20, 30, 0.5, 0))


def _chain_from_model(model, sc, vd_of, grid_cls, project, init, solve):
    """model: frames ascending by image id, points ascending by COLMAP id (both readers); vd_of: (frame, point) -> virtual depth"""
    sp = sc.spec
    vd = np.array([vd_of[(int(f), int(p))] for f, p in zip(model["fr"], model["pt"])])
    g = grid_cls(sp.raw_width, sp.raw_height, sp.lens_diameter, sp.lens_base_y, sp.grid_rotation, sp.grid_offset)
    obs = project(g, model, vd, sp.scale)
    B0, bL00 = init(model, vd, sc.cam0[0])
    cam0 = sc.cam0.copy(); cam0[1] = bL00; cam0[2] = B0
    pa = capi.ProblemArrays(obs["u"], obs["v"], obs["mcx"], obs["mcy"], obs["pt"], obs["fr"], cam0, model["views"], model["pts"], sc.spx, sc.scale, sc.config)
    return obs, (B0, bL00), pa, solve(pa)


def test_colmap_model_through_the_gpu_chain(built, tmp_path):
    sp = dataclasses.replace(scene.SceneSpec(8, 90, None, 0x506, 9201), k=(0.0, 0.0), p=(0.0, 0.0), noise_px=0.0)
    sc = scene.make_scene(sp)
    rs = np.random.default_rng(11)
    point_ids = rs.permutation(np.arange(3, 3 + 7 * sp.n_points, 7))     # sparse, unordered COLMAP ids
    image_ids = np.array([4, 9, 2, 17, 11, 30, 6, 8])                    # frame k of the scene is image image_ids[k]
    folder = str(tmp_path / "sparse")
    write_colmap_model(folder, sc, point_ids, image_ids)
    # both readers order frames by image id and points by COLMAP id: the scene's indices in that order
    f_rank = np.argsort(np.argsort(image_ids)); p_rank = np.argsort(np.argsort(point_ids))
    vd_of = {(int(f_rank[f]), int(p_rank[p])): float(v) for f, p, v in zip(sc.img_fr, sc.img_pt, sc.img_vd)}

    m = ColmapModel(folder)
    gpu_model = dict(x=m.x, y=m.y, fr=m.fr, pt=m.pt, views=m.views, pts=m.pts, w2c=m.world_to_cam_matrices())
    o = oc.calib_data(folder)
    cpu_model = dict(x=np.asarray(o["x"]), y=np.asarray(o["y"]), fr=np.asarray(o["fr"], np.uint32), pt=np.asarray(o["pt"], np.uint32),
                     views=np.asarray(o["views"]), pts=np.asarray(o["pts"]), w2c=np.asarray(o["world_to_cam"]))
    assert m.info.n_frames == sp.n_frames and m.info.n_points == sp.n_points and m.info.n_image_points == len(sc.img_x)   # the outliers are gone
    # the ingestion returns the scene: points bit for bit, rotations to round-off (Euler angles may sit on Eigen's other branch)
    assert np.array_equal(m.pts.reshape(-1, 3)[p_rank], sc.pts0.reshape(-1, 3))
    R_model = scene.euler_xyz(m.views.reshape(-1, 6)[:, :3])[f_rank]
    assert np.max(np.abs(R_model - scene.euler_xyz(sc.views0.reshape(-1, 6)[:, :3]))) < 1e-14

    def gpu_project(g, mod, vd, scale):
        r = g.projectPointsToRawImage(mod["x"], mod["y"], vd, scale, fr=mod["fr"], pt=mod["pt"])
        return dict(u=r.u, v=r.v, mcx=r.mcx, mcy=r.mcy, fr=r.fr, pt=r.pt)

    def cpu_project(g, mod, vd, scale):
        parts = []
        for f in range(sp.n_frames):
            sel = np.flatnonzero(mod["fr"] == f)
            r = g.project_frame(mod["x"][sel], mod["y"][sel], vd[sel], scale)
            parts.append((r.xR, r.yR, r.cX, r.cY, np.full(len(r.xR), f, np.uint32), mod["pt"][sel][r.point]))
        u, v, mcx, mcy, fr, pt = (np.concatenate(c) for c in zip(*parts))
        return dict(u=u, v=v, mcx=mcx, mcy=mcy, fr=fr, pt=pt)

    def gpu_init(mod, vd, fL):
        r = initPlenopticParameters(vd, mod["fr"], mod["pt"], mod["w2c"], mod["pts"].reshape(-1, 3), fL)
        return r.B_init, r.bL0_init

    def cpu_init(mod, vd, fL):
        r = oracle.init_plenoptic(capi.InitArrays(vd, mod["fr"], mod["pt"], mod["w2c"], mod["pts"].reshape(-1, 3), fL))
        return r.B_init, r.bL0_init

    def gpu_solve(pa):
        with BundleAdjustment(pa) as ba:
            s = ba.performBundleAdjustment()
            st = ba.calcReprojectionError(1.0)
        return s, st

    def cpu_solve(pa):
        return oracle.solve(pa), oracle.reproj_stats(pa, 1.0)

    obs_g, init_g, pa_g, (s_g, st_g) = _chain_from_model(gpu_model, sc, vd_of, MicroLensGrid, gpu_project, gpu_init, gpu_solve)
    obs_c, init_c, pa_c, (s_c, st_c) = _chain_from_model(cpu_model, sc, vd_of, OracleGrid, cpu_project, cpu_init, cpu_solve)
    for k in obs_g:                                                      # identical observation lists from identical ingested models
        assert np.array_equal(obs_g[k], obs_c[k]), k
    assert len(obs_g["u"]) > 3000
    assert init_g == pytest.approx(init_c, rel=1e-11)
    assert (s_g.iterations, s_g.termination) == (s_c.iterations, s_c.termination)
    assert s_g.final_cost == pytest.approx(s_c.final_cost, rel=1e-5, abs=1e-9)
    assert np.allclose(pa_g.cam[:5], pa_c.cam[:5], rtol=1e-6)
    assert max(st_g.std_x, st_g.std_y) < 2e-3 and st_g.num_inliers == st_g.num_points == len(obs_g["u"])
