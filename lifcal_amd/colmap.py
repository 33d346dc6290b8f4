"""Host-side mirror of the reference's COLMAP ingestion (include/lifcal_colmap.h): CalibrationData::readDataFromFirstCalibration
+ getCalibDataCV + getIntrinsicParamCV (reference src/CalibrationData/CalibrationData.cpp:56-127, :492-538, :561-571).
The files are read by the native library (host C++); this file only hands the arrays over as numpy."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as capi
from .bundle_adjustment import _check


class ColmapModel:
    """The flattened content of a COLMAP sparse model folder (cameras / images / points3D as .bin or .txt)."""

    def __init__(self, folder: str):
        lib = capi.load_library()
        h = C.c_void_p()
        _check(lib, lib.lifcal_colmap_read(str(folder).encode(), C.byref(h)), "lifcal_colmap_read")
        try:
            info = capi.ColmapInfo()
            _check(lib, lib.lifcal_colmap_get_info(h, C.byref(info)), "lifcal_colmap_get_info")
            self.info = info
            F, P, n = info.n_frames, info.n_points, info.n_image_points
            self.frame_ids = np.zeros(F, np.int32); self.views = np.zeros(6 * F); self.world_to_cam = np.zeros(16 * F); self.quat = np.zeros(4 * F)
            _check(lib, lib.lifcal_colmap_get_frames(h, self.frame_ids.ctypes.data_as(C.POINTER(C.c_int32)), capi.as_dptr(self.views),
                                                     capi.as_dptr(self.world_to_cam), capi.as_dptr(self.quat)), "lifcal_colmap_get_frames")
            self.colmap_point_ids = np.zeros(P, np.uint64); self.pts = np.zeros(3 * P)
            _check(lib, lib.lifcal_colmap_get_points(h, self.colmap_point_ids.ctypes.data_as(C.POINTER(C.c_uint64)), capi.as_dptr(self.pts)), "lifcal_colmap_get_points")
            self.x = np.zeros(n); self.y = np.zeros(n); self.fr = np.zeros(n, np.uint32); self.pt = np.zeros(n, np.uint32)
            _check(lib, lib.lifcal_colmap_get_image_points(h, capi.as_dptr(self.x), capi.as_dptr(self.y), capi.as_uptr(self.fr), capi.as_uptr(self.pt)), "lifcal_colmap_get_image_points")
        finally:
            lib.lifcal_colmap_free(h)

    # reference CalibrationData::getIntrinsicParamCV (:561-571): f, imageSize, c, k, p
    def getIntrinsicParamCV(self):
        i = self.info
        return i.f, (i.width, i.height), (i.cx, i.cy), (i.k1, i.k2), (i.p1, i.p2)

    def world_to_cam_matrices(self) -> np.ndarray:
        """(F, 4, 4) in mathematical (row, column) indexing (the ABI stores Eigen's column-major layout)"""
        return np.transpose(self.world_to_cam.reshape(-1, 4, 4), (0, 2, 1)).copy()
