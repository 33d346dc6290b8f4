"""oracle — CPU restatement of LiFCal's bundle-adjustment hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
lifcal_amd (the product) never does.  PARITY UNPINNED: the reference has no fixtures for this
path and cannot be built in this image (see oracle/README.md).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from lifcal_amd import _capi as capi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liblifcal_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("lifcal_oracle.cpp", "lifcal_mla.cpp", "model.hpp", "analytic.hpp", "jet.hpp", "lifcal_oracle.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "lifcal_ba.h"))
    stale = force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liblifcal_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        d = C.c_double
        dp = capi.dptr
        L.lo_project_point.argtypes = [dp, d, d, d, d, d, dp, dp, dp, C.c_int, dp, C.c_int, dp]
        L.lo_rigid_transform.argtypes = [dp, dp]
        L.lo_residual_block.argtypes = [C.c_uint32, C.c_int, dp, dp, dp, d, d, d, d, d, d, d, dp, dp]
        L.lo_constraint_block.argtypes = [dp, dp, d, d, dp, dp]
        L.lo_cost.argtypes = [C.POINTER(capi.Problem), d, C.c_int, dp]
        L.lo_residuals.argtypes = [C.POINTER(capi.Problem), dp]
        L.lo_reduced_size.argtypes = [C.POINTER(capi.Problem), capi.uptr, capi.uptr]
        L.lo_sweep.argtypes = [C.POINTER(capi.Problem), C.POINTER(capi.Options), d, C.c_int, C.POINTER(capi.SweepOut), dp, dp]
        L.lo_sweep_analytic.argtypes = L.lo_sweep.argtypes
        L.lo_solve.argtypes = [C.POINTER(capi.Problem), C.POINTER(capi.Options), C.c_int, C.POINTER(capi.Summary)]
        L.lo_reproj_stats.argtypes = [C.POINTER(capi.Problem), d, C.POINTER(capi.Stats), dp]
        L.lo_hardware_threads.restype = C.c_int
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(capi.dptr) if a is not None else None


def init_plenoptic(arrs: capi.InitArrays) -> capi.InitResult:
    """reference CameraCalibration::initPlenopticParameters (src/CameraCalibration.cpp:456-499) on the CPU"""
    res = capi.InitResult()
    L = lib()
    L.lo_init_plenoptic.restype = C.c_int
    L.lo_init_plenoptic.argtypes = [C.POINTER(capi.InitProblem), C.POINTER(capi.InitResult)]
    rc = L.lo_init_plenoptic(C.byref(arrs.struct), C.byref(res))
    assert rc == 0, rc
    return res


def set_fixed_frames(mask=None):
    """poses held constant in the following sweeps / solves (None clears); mirrors lifcal_ba_set_fixed_frames"""
    L = lib()
    L.lo_set_fixed_frames.argtypes = [C.POINTER(C.c_uint8), C.c_uint32]
    if mask is None:
        L.lo_set_fixed_frames(None, 0)
    else:
        m = np.ascontiguousarray(mask, np.uint8)
        L.lo_set_fixed_frames(m.ctypes.data_as(C.POINTER(C.c_uint8)), len(m))


def hardware_threads() -> int:
    return int(lib().lo_hardware_threads())


def residual_block(config, arity, cam, view, point, u, v, mcx, mcy, spx, scale, spy=None, jacobian=True):
    """One residual block: r[2] and the autodiff Jacobian [2, 26] = camera 17 | view 6 | point 3."""
    cam = np.ascontiguousarray(cam, np.float64); view = np.ascontiguousarray(view, np.float64)
    point = np.ascontiguousarray(point, np.float64)
    r = np.zeros(2); J = np.zeros((2, 26)) if jacobian else None
    rc = lib().lo_residual_block(int(config), int(arity), _dp(cam), _dp(view), _dp(point), u, v, mcx, mcy,
                                 spx, spx if spy is None else spy, scale, _dp(r), _dp(J))
    assert rc == 0
    return r, J


def constraint_block(p1, p2, distance, sigma):
    p1 = np.ascontiguousarray(p1, np.float64); p2 = np.ascontiguousarray(p2, np.float64)
    r = np.zeros(1); J = np.zeros(6)
    lib().lo_constraint_block(_dp(p1), _dp(p2), distance, sigma, _dp(r), _dp(J))
    return r[0], J


def project_point(pc, spx, spy, fL, bL0, B, c_raw, ml, radial, tangential, adj):
    pc = np.ascontiguousarray(pc, np.float64); c_raw = np.ascontiguousarray(c_raw, np.float64)
    ml = np.ascontiguousarray(ml, np.float64)
    rad = np.ascontiguousarray(radial, np.float64) if radial is not None and len(radial) else None
    tan = np.ascontiguousarray(tangential, np.float64) if tangential is not None else None
    out = np.zeros(2)
    lib().lo_project_point(_dp(pc), spx, spy, fL, bL0, B, _dp(c_raw), _dp(ml), _dp(rad), 0 if rad is None else len(rad),
                           _dp(tan), int(bool(adj)), _dp(out))
    return out


def rigid_transform(view):
    view = np.ascontiguousarray(view, np.float64); RT = np.zeros(12)
    lib().lo_rigid_transform(_dp(view), _dp(RT))
    return RT.reshape(3, 4)


def cost(pa: capi.ProblemArrays, loss_scale=0.5, threads=1) -> float:
    c = np.zeros(1)
    rc = lib().lo_cost(C.byref(pa.struct), loss_scale, threads, _dp(c))
    assert rc == 0, rc
    return float(c[0])


def residuals(pa: capi.ProblemArrays) -> np.ndarray:
    r = np.zeros(2 * pa.struct.n_obs)
    rc = lib().lo_residuals(C.byref(pa.struct), _dp(r))
    assert rc == 0, rc
    return r.reshape(-1, 2)


def reduced_size(pa: capi.ProblemArrays):
    n = np.zeros(1, np.uint32); m = np.zeros(1, np.uint32)
    rc = lib().lo_reduced_size(C.byref(pa.struct), n.ctypes.data_as(capi.uptr), m.ctypes.data_as(capi.uptr))
    assert rc == 0, rc
    return int(n[0]), int(m[0])


class SweepResult:
    pass


def sweep(pa: capi.ProblemArrays, radius=1e4, options=None, threads=1, want_matrices=True, analytic=False) -> SweepResult:
    """One Jacobian + Schur sweep.  analytic=False: dual numbers through the functor, as ceres::AutoDiffCostFunction evaluates the
    reference's residual blocks; analytic=True: hand-derived Jacobian with per-lens / per-frame tables (oracle/analytic.hpp)."""
    o = options if options is not None else capi.default_options_py()
    n, m = reduced_size(pa)
    res = SweepResult()
    out = capi.SweepOut()
    if want_matrices:
        res.S = np.zeros((n, n)); res.rhs = np.zeros(n); res.gradient_reduced = np.zeros(n)
        res.point_gradient = np.zeros(3 * pa.struct.n_points); res.point_hessian_inv = np.zeros(9 * pa.struct.n_points)
        out.S, out.rhs, out.gradient_reduced = _dp(res.S), _dp(res.rhs), _dp(res.gradient_reduced)
        out.point_gradient, out.point_hessian_inv = _dp(res.point_gradient), _dp(res.point_hessian_inv)
    te = np.zeros(1); ts = np.zeros(1)
    fn = lib().lo_sweep_analytic if analytic else lib().lo_sweep
    rc = fn(C.byref(pa.struct), C.byref(o), radius, threads, C.byref(out), _dp(te), _dp(ts))
    res.rc = rc
    res.cost = out.cost; res.gradient_max_norm = out.gradient_max_norm
    res.n_reduced = n; res.n_promoted = m
    res.seconds = out.seconds; res.seconds_eval = float(te[0]); res.seconds_schur = float(ts[0])
    return res


def solve(pa: capi.ProblemArrays, options=None, threads=1) -> capi.Summary:
    o = options if options is not None else capi.default_options_py()
    s = capi.Summary()
    rc = lib().lo_solve(C.byref(pa.struct), C.byref(o), threads, C.byref(s))
    assert rc == 0, rc
    return s


def reproj_stats(pa: capi.ProblemArrays, thr=1.0, want_errors=False):
    st = capi.Stats()
    err = np.zeros(2 * pa.struct.n_obs) if want_errors else None
    rc = lib().lo_reproj_stats(C.byref(pa.struct), thr, C.byref(st), _dp(err))
    assert rc == 0, rc
    return (st, err.reshape(-1, 2)) if want_errors else st
