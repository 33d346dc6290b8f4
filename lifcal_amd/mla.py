"""Host mirror of the reference's MicroLensGrid + CameraCalibration::projectPointsToRawImage over include/lifcal_mla.h.

Reference: src/MicroLensGrid/MicroLensGrid.cpp:56-270, :338-421; src/CameraCalibration.cpp:521-632, :637-769.  The lens maps
and the projection run on the GPU inside liblifcal_ba.so; there is no Python or CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _capi as capi
from .bundle_adjustment import LifcalError


@dataclass
class RawObservations:
    """rawImageCoordinates / microLensCenter / objectCoordinatesByRawID of all frames, concatenated in frame order"""
    u: np.ndarray
    v: np.ndarray
    mcx: np.ndarray
    mcy: np.ndarray
    src: np.ndarray                    # index of the image point each observation came from
    fr: Optional[np.ndarray] = None
    pt: Optional[np.ndarray] = None


def _check(rc: int, what: str):
    if rc < 0:
        lib = capi.load_library()
        raise LifcalError(f"{what}: {lib.lifcal_ba_strerror(rc).decode()} ({lib.lifcal_ba_last_error().decode()})")


class MicroLensGrid:
    """MicroLensGrid::readInGrid (values of the MLA file passed in) + createGrid + defineMlMaps, with the epipolar web of
    CameraCalibration::defineEpiPolarLines attached, as projectPointsToRawImage needs all of them."""

    def __init__(self, width: int, height: int, lens_diameter: float, lens_base_y: Sequence[float] = (0.5, 0.8660254),
                 rotation: float = 0.0, offset: Sequence[float] = (0.0, 0.0), rotation_on_grid: bool = True, device: int = 0):
        self._lib = capi.load_library()
        self.params = capi.MlaParams(int(width), int(height), float(lens_diameter), (C.c_float * 2)(*lens_base_y), float(rotation),
                                     (C.c_float * 2)(*offset), 1 if rotation_on_grid else 0)
        self._h = C.c_void_p()
        _check(self._lib.lifcal_mla_create(C.byref(self.params), int(device), C.byref(self._h)), "lifcal_mla_create")
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        _check(self._lib.lifcal_mla_info(self._h, C.byref(a), C.byref(b), C.byref(c)), "lifcal_mla_info")
        self.n_lenses, self.n_web_groups, self.n_web_lines = a.value, b.value, c.value
        self.width, self.height = int(width), int(height)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lifcal_mla_destroy(self._h)
            self._h = None

    __del__ = close

    def lenses(self):
        cx = np.zeros(self.n_lenses, np.float32); cy = np.zeros(self.n_lenses, np.float32); t = np.zeros(self.n_lenses, np.int32)
        _check(self._lib.lifcal_mla_get_lenses(self._h, cx.ctypes.data_as(capi._fptr), cy.ctypes.data_as(capi._fptr), t.ctypes.data_as(capi._iptr)), "lenses")
        return cx, cy, t

    def maps(self):
        a = np.zeros((self.height, self.width), np.int32); b = np.zeros((self.height, self.width), np.int32)
        _check(self._lib.lifcal_mla_get_maps(self._h, a.ctypes.data_as(capi._iptr), b.ctypes.data_as(capi._iptr)), "maps")
        return a, b

    def web(self):
        n = self.n_web_lines
        d = np.zeros(n); ex = np.zeros(n); ey = np.zeros(n); g = np.zeros(n, np.int32)
        _check(self._lib.lifcal_mla_get_web(self._h, capi.as_dptr(d), capi.as_dptr(ex), capi.as_dptr(ey), g.ctypes.data_as(capi._iptr)), "web")
        return d, ex, ey, g

    def projectPointsToRawImage(self, x, y, vdepth, depth_to_raw_im_scale: int = 1, fr=None, pt=None) -> RawObservations:
        """All frames at once: x, y, vdepth (and optionally fr, pt) per image point, frames concatenated in order."""
        x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64); vd = np.ascontiguousarray(vdepth, np.float64)
        n = len(x)
        assert len(y) == n and len(vd) == n
        fr_a = np.ascontiguousarray(fr, np.uint32) if fr is not None else None
        pt_a = np.ascontiguousarray(pt, np.uint32) if pt is not None else None
        pts = capi.MlaPoints(n, capi.as_dptr(x), capi.as_dptr(y), capi.as_dptr(vd), capi.as_uptr(fr_a), capi.as_uptr(pt_a))
        obs = capi.MlaObservations()
        rc = self._lib.lifcal_mla_project(self._h, int(depth_to_raw_im_scale), C.byref(pts), C.byref(obs))   # count
        _check(rc, "lifcal_mla_project")
        m = int(obs.n_obs)
        out = RawObservations(np.zeros(m), np.zeros(m), np.zeros(m), np.zeros(m), np.zeros(m, np.uint32),
                              np.zeros(m, np.uint32) if fr_a is not None else None, np.zeros(m, np.uint32) if pt_a is not None else None)
        if m == 0:
            return out
        obs = capi.MlaObservations(m, 0, capi.as_dptr(out.u), capi.as_dptr(out.v), capi.as_dptr(out.mcx), capi.as_dptr(out.mcy),
                                   capi.as_uptr(out.src), capi.as_uptr(out.fr), capi.as_uptr(out.pt))
        rc = self._lib.lifcal_mla_project(self._h, int(depth_to_raw_im_scale), C.byref(pts), C.byref(obs))
        _check(rc, "lifcal_mla_project")
        if rc != 0 or int(obs.n_obs) != m:
            raise LifcalError("lifcal_mla_project: observation count changed between the count and the fill call")
        return out
