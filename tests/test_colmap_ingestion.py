"""COLMAP model ingestion (include/lifcal_colmap.h): the native reader against the oracle's restatement
(oracle/colmap.py) and the committed fixture (tests/golden/colmap_small, written by tools/make_colmap_fixture.py), both file
layouts, plus the properties the reference's getCalibDataCV promises (src/CalibrationData/CalibrationData.cpp:492-538).
Host code: runs without a GPU."""
import os
import shutil

import numpy as np
import pytest

from lifcal_amd import scene
from lifcal_amd.bundle_adjustment import LifcalError
from lifcal_amd.colmap import ColmapModel
from oracle import colmap as oc

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "colmap_small")


@pytest.mark.parametrize("layout", ["bin", "txt"])
def test_reader_matches_oracle_and_fixture(built, layout):
    m = ColmapModel(os.path.join(FIX, layout))
    e = np.load(os.path.join(FIX, "expected.npz"))
    o = oc.calib_data(os.path.join(FIX, layout))
    assert m.info.binary == (1 if layout == "bin" else 0)
    for ref in (e, o):
        assert (m.info.camera_model_id, m.info.width, m.info.height) == (int(ref["camera_model_id"]), int(ref["width"]), int(ref["height"]))
        got = np.array([m.info.fx, m.info.fy, m.info.cx, m.info.cy, m.info.k1, m.info.k2, m.info.p1, m.info.p2])
        assert np.array_equal(got, ref["params"]) and m.info.f == float(ref["f"])
        # everything that is copied or parsed is bit-exact (strtod and Python's float() both round correctly)
        assert np.array_equal(m.frame_ids, ref["frame_ids"]) and np.array_equal(m.colmap_point_ids, ref["colmap_point_ids"])
        assert np.array_equal(m.pts, ref["pts"]) and np.array_equal(m.x, ref["x"]) and np.array_equal(m.y, ref["y"])
        assert np.array_equal(m.fr, ref["fr"]) and np.array_equal(m.pt, ref["pt"])
        assert np.array_equal(m.views.reshape(-1, 6)[:, 3:], ref["views"].reshape(-1, 6)[:, 3:])
        # quaternion normalisation, rotation matrix and Euler angles: same formulas, libm vs Python's math -> round-off
        assert np.max(np.abs(m.quat - ref["quat"])) < 1e-15
        assert np.max(np.abs(m.world_to_cam_matrices() - ref["world_to_cam"])) < 1e-14
        assert np.max(np.abs(m.views - ref["views"])) < 1e-14


def test_ingestion_properties(built):
    m = ColmapModel(os.path.join(FIX, "bin"))
    F = m.info.n_frames
    assert list(m.frame_ids) == sorted(m.frame_ids) and list(m.colmap_point_ids) == sorted(m.colmap_point_ids)
    assert m.info.n_image_points == len(m.x) and np.all(np.diff(m.fr.astype(int)) >= 0)           # frame-major
    assert 5 in m.frame_ids and not np.any(m.fr == list(m.frame_ids).index(5))                     # the image without points is a frame
    for f in range(F):                                                                              # one observation per (frame, point)
        sel = m.pt[m.fr == f]
        assert len(sel) == len(set(sel.tolist()))
    views = m.views.reshape(F, 6)
    assert np.all(views[:, 0] >= 0.0) and np.all(views[:, 0] <= np.pi)                              # Eigen: first angle in [0, pi]
    # the angles reproduce the rotation of the quaternion: Rx Ry Rz (CameraModel.h:251-254) == toRotationMatrix()
    R = scene.euler_xyz(views[:, :3])
    W = m.world_to_cam_matrices()
    assert np.max(np.abs(R - W[:, :3, :3])) < 1e-14
    assert np.array_equal(W[:, :3, 3], views[:, 3:]) and np.array_equal(W[:, 3], np.tile([0, 0, 0, 1.0], (F, 1)))
    # unit quaternions, orthonormal rotations
    assert np.max(np.abs(np.linalg.norm(m.quat.reshape(F, 4), axis=1) - 1)) < 1e-15
    assert np.max(np.abs(np.einsum("fij,fkj->fik", R, R) - np.eye(3))) < 1e-14
    f, size, c, k, p = m.getIntrinsicParamCV()
    assert f == (m.info.fx + m.info.fy) / 2 and size == (1024, 1024) and c == (511.3, 513.9)


def test_errors(built, tmp_path):
    with pytest.raises(LifcalError):                       # files missing (reference: "some data files ... are missing", returns false)
        ColmapModel(str(tmp_path))
    d = tmp_path / "mixed"; d.mkdir()                      # an incomplete binary triple falls back to a complete text triple
    shutil.copy(os.path.join(FIX, "bin", "cameras.bin"), d)
    for n in ("cameras.txt", "images.txt", "points3D.txt"):
        shutil.copy(os.path.join(FIX, "txt", n), d)
    assert ColmapModel(str(d)).info.binary == 0
    t = tmp_path / "trunc"; t.mkdir()                      # truncated binary file
    for n in ("cameras.bin", "points3D.bin"):
        shutil.copy(os.path.join(FIX, "bin", n), t)
    raw = open(os.path.join(FIX, "bin", "images.bin"), "rb").read()
    (t / "images.bin").write_bytes(raw[: len(raw) // 2])
    with pytest.raises(LifcalError, match="truncated"):
        ColmapModel(str(t))
    u = tmp_path / "unknown"; u.mkdir()                    # a 2D point referencing a 3D point that does not exist
    for n in ("cameras.txt", "images.txt"):
        shutil.copy(os.path.join(FIX, "txt", n), u)
    lines = [ln for ln in open(os.path.join(FIX, "txt", "points3D.txt")) if not ln.startswith("101 ")]
    (u / "points3D.txt").write_text("".join(lines))
    with pytest.raises(LifcalError, match="not in points3D"):
        ColmapModel(str(u))
    c = tmp_path / "nocam1"; c.mkdir()                     # LoadIntrinsicOrientation wants camera id 1
    for n in ("images.txt", "points3D.txt"):
        shutil.copy(os.path.join(FIX, "txt", n), c)
    (c / "cameras.txt").write_text("2 PINHOLE 640 480 500.0 505.0 320.0 240.0\n")
    with pytest.raises(LifcalError, match="camera"):
        ColmapModel(str(c))


def test_ingested_model_feeds_the_init_step(built):
    """the flattened arrays are what lifcal_init_plenoptic / lifcal_mla_points take: shapes, index ranges, camera-frame depths positive"""
    m = ColmapModel(os.path.join(FIX, "txt"))
    W = m.world_to_cam_matrices()
    P = m.pts.reshape(-1, 3)
    z = np.einsum("nj,nj->n", W[m.fr, 2, :3], P[m.pt]) + W[m.fr, 2, 3]
    assert m.fr.max() < m.info.n_frames and m.pt.max() < m.info.n_points and np.all(np.isfinite(z))
