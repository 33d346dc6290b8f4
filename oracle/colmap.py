"""oracle/colmap.py — TEST INFRASTRUCTURE ONLY: restatement of the reference's COLMAP ingestion in plain Python.

    read_model          <- colmap::Reconstruction::Read as CalibrationData::readDataFromFirstCalibration calls it
                           (reference src/CalibrationData/CalibrationData.cpp:56-127): .bin triple if complete, else .txt triple
    intrinsics          <- IntrinsicOrientation::LoadIntrinsicOrientation / getIntrinsicParam (IntrinsicOrientation.cpp:40-71)
    inlier_points       <- Images::LoadImageCoordinates (ImagePoints/Images.cpp:29-101)
    calib_data          <- CalibrationData::getCalibDataCV (:492-538), frames in ascending image id, points in ascending id
    euler_angles_012    <- Eigen::MatrixBase::eulerAngles(0, 1, 2) (:531; Eigen is not in this image: its published algorithm)
COLMAP (third-party, commit 1f69517d per reference installation/Dockerfile) is absent; the file layouts are COLMAP's
published model format.  PARITY UNPINNED: the reference holds no model files or tests for this path.
"""
from __future__ import annotations

import math
import os
import struct

import numpy as np

MODEL_PARAMS = {0: 3, 1: 4, 2: 4, 3: 5, 4: 8, 5: 8, 6: 12, 7: 5, 8: 4, 9: 5, 10: 12}
MODEL_NAMES = ["SIMPLE_PINHOLE", "PINHOLE", "SIMPLE_RADIAL", "RADIAL", "OPENCV", "OPENCV_FISHEYE", "FULL_OPENCV", "FOV",
               "SIMPLE_RADIAL_FISHEYE", "RADIAL_FISHEYE", "THIN_PRISM_FISHEYE"]
INVALID = 2 ** 64 - 1


def _read_bin(d):
    cams, imgs, pts = {}, [], []
    with open(os.path.join(d, "cameras.bin"), "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n):
            cid, model, w, h = struct.unpack("<IiQQ", f.read(24))
            params = struct.unpack("<%dd" % MODEL_PARAMS[model], f.read(8 * MODEL_PARAMS[model]))
            cams[cid] = (model, w, h, list(params))
    with open(os.path.join(d, "images.bin"), "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n):
            (iid,) = struct.unpack("<I", f.read(4))
            q = struct.unpack("<4d", f.read(32)); t = struct.unpack("<3d", f.read(24))
            (cid,) = struct.unpack("<I", f.read(4))
            name = b""
            while True:
                c = f.read(1)
                if c == b"\x00":
                    break
                name += c
            (m,) = struct.unpack("<Q", f.read(8))
            raw = np.frombuffer(f.read(24 * m), dtype=np.dtype([("x", "<f8"), ("y", "<f8"), ("id", "<u8")]))
            imgs.append((iid, q, t, cid, [(float(a), float(b), int(c)) for a, b, c in raw]))
    with open(os.path.join(d, "points3D.bin"), "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n):
            pid, x, y, z = struct.unpack("<Q3d", f.read(32))
            f.read(3); f.read(8)
            (tl,) = struct.unpack("<Q", f.read(8))
            f.read(8 * tl)
            pts.append((pid, (x, y, z)))
    return cams, imgs, pts


def _lines(path):
    with open(path) as f:
        return [ln.strip() for ln in f.read().split("\n")]


def _read_txt(d):
    cams, imgs, pts = {}, [], []
    for ln in _lines(os.path.join(d, "cameras.txt")):
        if not ln or ln[0] == "#":
            continue
        t = ln.split()
        cams[int(t[0])] = (MODEL_NAMES.index(t[1]), int(t[2]), int(t[3]), [float(v) for v in t[4:]])
    L = _lines(os.path.join(d, "images.txt"))
    i = 0
    while i < len(L):
        ln = L[i]; i += 1
        if not ln or ln[0] == "#":
            continue
        t = ln.split()
        iid = int(t[0]); q = tuple(float(v) for v in t[1:5]); tr = tuple(float(v) for v in t[5:8]); cid = int(t[8])
        p = L[i].split() if i < len(L) else []
        i += 1
        pts2 = [(float(p[3 * k]), float(p[3 * k + 1]), int(p[3 * k + 2]) % 2 ** 64) for k in range(len(p) // 3)]
        imgs.append((iid, q, tr, cid, pts2))
    for ln in _lines(os.path.join(d, "points3D.txt")):
        if not ln or ln[0] == "#":
            continue
        t = ln.split()
        pts.append((int(t[0]), (float(t[1]), float(t[2]), float(t[3]))))
    return cams, imgs, pts


def read_model(folder):
    have = lambda ext: all(os.path.exists(os.path.join(folder, n + ext)) for n in ("cameras", "images", "points3D"))
    if have(".bin"):
        return _read_bin(folder) + (True,)
    if have(".txt"):
        return _read_txt(folder) + (False,)
    raise FileNotFoundError("some data files of the COLMAP model are missing")


def quat_to_matrix(q):
    w, x, y, z = q
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1.0 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1.0 - (txx + tzz), tyz - twx], [txz - twy, tyz + twx, 1.0 - (txx + tyy)]])


def euler_angles_012(m):
    """Eigen 3.3/3.4 EulerAngles.h with (a0, a1, a2) = (0, 1, 2): i, j, k = 0, 1, 2 and odd = 0"""
    r0 = math.atan2(m[1, 2], m[2, 2])
    c2 = math.sqrt(m[0, 0] * m[0, 0] + m[0, 1] * m[0, 1])
    if r0 > 0.0:
        r0 -= math.pi
        r1 = math.atan2(-m[0, 2], -c2)
    else:
        r1 = math.atan2(-m[0, 2], c2)
    s1, c1 = math.sin(r0), math.cos(r0)
    r2 = math.atan2(s1 * m[2, 0] - c1 * m[1, 0], c1 * m[1, 1] - s1 * m[2, 1])
    return np.array([-r0, -r1, -r2])


def calib_data(folder):
    """dict with the arrays include/lifcal_colmap.h hands out"""
    cams, imgs, pts, binary = read_model(folder)
    model, w, h, params = cams[1]
    assert len(params) == MODEL_PARAMS[model] and len(params) >= 8
    out = dict(binary=int(binary), camera_model_id=model, width=w, height=h, params=np.array(params[:8]), f=(params[0] + params[1]) / 2)
    ids = []
    xyz = []
    for pid, c in sorted(pts, key=lambda p: p[0]):
        if pid in ids:
            continue
        ids.append(pid); xyz.append(c)
    dense = {pid: i for i, pid in enumerate(ids)}
    frame_ids, views, w2c, quat, X, Y, FR, PT = [], [], [], [], [], [], [], []
    seen_img = set()
    for iid, q, t, cid, p2 in sorted(imgs, key=lambda im: im[0]):
        if iid in seen_img:
            continue
        seen_img.add(iid)
        f = len(frame_ids)
        frame_ids.append(iid)
        qa = np.array(q, float); qa = qa / math.sqrt(float(qa[0] * qa[0] + qa[1] * qa[1] + qa[2] * qa[2] + qa[3] * qa[3]))
        R = quat_to_matrix(qa)
        views.append(np.concatenate([euler_angles_012(R), t]))
        M = np.eye(4); M[:3, :3] = R; M[:3, 3] = t
        w2c.append(M); quat.append(qa)
        seen = set()
        for x, y, pid in p2:
            if pid == INVALID or pid in seen:
                continue
            seen.add(pid)
            X.append(x); Y.append(y); FR.append(f); PT.append(dense[pid])
    out.update(frame_ids=np.array(frame_ids, np.int32), views=np.array(views).reshape(-1), world_to_cam=np.array(w2c), quat=np.array(quat).reshape(-1),
               colmap_point_ids=np.array(ids, np.uint64), pts=np.array(xyz).reshape(-1),
               x=np.array(X), y=np.array(Y), fr=np.array(FR, np.uint32), pt=np.array(PT, np.uint32))
    return out
