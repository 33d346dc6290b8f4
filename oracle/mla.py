"""oracle.mla — ctypes access to oracle/lifcal_mla.cpp (micro-lens grid, maps, epipolar web, projectPointsToRawImage).

TEST INFRASTRUCTURE ONLY, parity unpinned (see oracle/README.md).  Reference: src/MicroLensGrid/MicroLensGrid.cpp:186-270,
:338-421; src/CameraCalibration.cpp:521-632, :637-769.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import lib as _lib


class MlaParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("lens_diameter", C.c_float), ("lens_base_y", C.c_float * 2),
                ("rotation", C.c_float), ("offset", C.c_float * 2), ("rotation_on_grid", C.c_int32)]


@dataclass
class Observations:
    xR: np.ndarray
    yR: np.ndarray
    cX: np.ndarray
    cY: np.ndarray
    point: np.ndarray


_bound = False


def _L():
    global _bound
    L = _lib()
    if not _bound:
        vp, i32p, fp, dp, i64p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int64)
        L.lo_mla_create.argtypes = [C.POINTER(MlaParams)]; L.lo_mla_create.restype = vp
        L.lo_mla_destroy.argtypes = [vp]; L.lo_mla_destroy.restype = None
        L.lo_mla_counts.argtypes = [vp, i32p, i32p, i32p]
        L.lo_mla_lenses.argtypes = [vp, fp, fp, i32p]
        L.lo_mla_maps.argtypes = [vp, i32p, i32p]
        L.lo_mla_web.argtypes = [vp, dp, dp, dp, i32p]
        L.lo_mla_project_frame.argtypes = [vp, C.c_int32, C.c_int64, dp, dp, dp, C.c_int64, dp, dp, dp, dp, i64p]
        L.lo_mla_project_frame.restype = C.c_int64
        _bound = True
    return L


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class MicroLensGrid:
    """MicroLensGrid::readInGrid (derived values) + createGrid + defineMlMaps, and CameraCalibration::defineEpiPolarLines."""

    def __init__(self, width, height, lens_diameter, lens_base_y=(0.5, 0.8660254), rotation=0.0, offset=(0.0, 0.0), rotation_on_grid=True):
        self.params = MlaParams(int(width), int(height), float(lens_diameter), (C.c_float * 2)(*lens_base_y), float(rotation),
                                (C.c_float * 2)(*offset), 1 if rotation_on_grid else 0)
        self._h = _L().lo_mla_create(C.byref(self.params))
        if not self._h:
            raise ValueError("bad micro-lens grid parameters")
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        _L().lo_mla_counts(self._h, C.byref(a), C.byref(b), C.byref(c))
        self.n_lenses, self.n_web_groups, self.n_web_lines = a.value, b.value, c.value
        self.width, self.height = int(width), int(height)

    def __del__(self):
        if getattr(self, "_h", None):
            _L().lo_mla_destroy(self._h)
            self._h = None

    def lenses(self):
        cx = np.zeros(self.n_lenses, np.float32); cy = np.zeros(self.n_lenses, np.float32); t = np.zeros(self.n_lenses, np.int32)
        _L().lo_mla_lenses(self._h, _p(cx, C.c_float), _p(cy, C.c_float), _p(t, C.c_int32))
        return cx, cy, t

    def maps(self):
        a = np.zeros((self.height, self.width), np.int32); b = np.zeros((self.height, self.width), np.int32)
        _L().lo_mla_maps(self._h, _p(a, C.c_int32), _p(b, C.c_int32))
        return a, b

    def web(self):
        n = self.n_web_lines
        d = np.zeros(n); ex = np.zeros(n); ey = np.zeros(n); g = np.zeros(n, np.int32)
        _L().lo_mla_web(self._h, _p(d, C.c_double), _p(ex, C.c_double), _p(ey, C.c_double), _p(g, C.c_int32))
        return d, ex, ey, g

    def project_frame(self, px, py, vdepth, depth_to_raw_im_scale=1) -> Observations:
        px = np.ascontiguousarray(px, np.float64); py = np.ascontiguousarray(py, np.float64); vd = np.ascontiguousarray(vdepth, np.float64)
        n = len(px)
        cap = max(64, 64 * n)
        while True:
            xR = np.zeros(cap); yR = np.zeros(cap); cX = np.zeros(cap); cY = np.zeros(cap); pt = np.zeros(cap, np.int64)
            m = _L().lo_mla_project_frame(self._h, int(depth_to_raw_im_scale), n, _p(px, C.c_double), _p(py, C.c_double), _p(vd, C.c_double),
                                          cap, _p(xR, C.c_double), _p(yR, C.c_double), _p(cX, C.c_double), _p(cY, C.c_double), _p(pt, C.c_int64))
            if m >= 0:
                return Observations(xR[:m].copy(), yR[:m].copy(), cX[:m].copy(), cY[:m].copy(), pt[:m].copy())
            cap = -m
