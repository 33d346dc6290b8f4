"""Host mirror of the reference's result writers (CameraCalibration::store*, src/CameraCalibration.cpp:1296-1617) over
include/lifcal_io.h.  Host code inside liblifcal_ba.so; no GPU needed."""
from __future__ import annotations

import ctypes as C
import os
from typing import Sequence

import numpy as np

from . import _capi as capi
from .bundle_adjustment import LifcalError


def camera_model(image_size: Sequence[int], pixel_size: float, camera: Sequence[float], config: int) -> capi.CameraModel:
    """camera[] as performBundleAdjustment leaves it (fL, bL0, B, cx, cy, radial.., tangential..) -> the members the
    reference copies it into (:965-988)."""
    n_rad = config & 3
    tan = bool(config & 0x004)
    m = capi.CameraModel()
    m.image_width, m.image_height, m.pixel_size = int(image_size[0]), int(image_size[1]), float(pixel_size)
    m.fL, m.bL0, m.B, m.cx, m.cy = (float(camera[i]) for i in range(5))
    m.n_radial = n_rad
    for i in range(n_rad):
        m.radial[i] = float(camera[5 + i])
    m.tangential = 1 if tan else 0
    if tan:
        m.tangential_dist[0], m.tangential_dist[1] = float(camera[5 + n_rad]), float(camera[6 + n_rad])
    m.ml_center_adjustment = 1 if config & 0x800 else 0
    return m


def _ok(rc, what):
    if rc != 0:
        raise LifcalError(f"{what}: {capi.load_library().lifcal_ba_last_error().decode() or 'cannot write'}")


def storeCameraModel(dir_results: str, model: capi.CameraModel):
    _ok(capi.load_library().lifcal_write_camera_model(os.path.join(dir_results, "CameraModel.xml").encode(), C.byref(model)), "storeCameraModel")


def storeExtrinsicOrientations(dir_results: str, frame_ids, views):
    ids = np.ascontiguousarray(frame_ids, np.int32); v = np.ascontiguousarray(views, np.float64).reshape(-1)
    _ok(capi.load_library().lifcal_write_extrinsic_orientations_xml(os.path.join(dir_results, "extrinsicOrientations.xml").encode(), len(ids),
                                                                    ids.ctypes.data_as(capi._iptr), capi.as_dptr(v)), "storeExtrinsicOrientations")


def storeExtrinsicOrientationsTxt(dir_results: str, frame_ids, views):
    ids = np.ascontiguousarray(frame_ids, np.int32); v = np.ascontiguousarray(views, np.float64).reshape(-1)
    _ok(capi.load_library().lifcal_write_extrinsic_orientations_txt(os.path.join(dir_results, "ExtrinsicOrientations.txt").encode(), len(ids),
                                                                    ids.ctypes.data_as(capi._iptr), capi.as_dptr(v)), "storeExtrinsicOrientationsTxt")


def storeRawImagePointsCsv(dir_results: str, frame_ids, fr, u, v, x_proj, y_proj, pt):
    ids = np.ascontiguousarray(frame_ids, np.int32)
    fr = np.ascontiguousarray(fr, np.uint32); pt = np.ascontiguousarray(pt, np.uint32)
    arrs = [np.ascontiguousarray(a, np.float64) for a in (u, v, x_proj, y_proj)]
    _ok(capi.load_library().lifcal_write_raw_image_points_csv(os.path.join(dir_results, "rawImagePoints.csv").encode(), len(fr), len(ids), ids.ctypes.data_as(capi._iptr),
                                                              capi.as_uptr(fr), *[capi.as_dptr(a) for a in arrs], capi.as_uptr(pt)), "storeRawImagePointsCsv")


def storeProtocol(dir_results: str, model: capi.CameraModel, config: int, stats):
    p = capi.Protocol()
    p.model = model
    p.refine_poses = 1 if config & 0x100 else 0
    p.refine_points = 1 if config & 0x400 else 0   # the flag as set (:1595), whether or not poses were refined
    p.robust_cost = 1 if config & 0x200 else 0
    p.std_x, p.std_y, p.mae_x, p.mae_y = stats.std_x, stats.std_y, stats.mae_x, stats.mae_y
    _ok(capi.load_library().lifcal_write_protocol(os.path.join(dir_results, "calibrationProtocol.txt").encode(), C.byref(p)), "storeProtocol")
