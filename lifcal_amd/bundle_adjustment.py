"""Host-side mirror of LiFCal's bundle-adjustment seam on top of the C ABI (include/lifcal_ba.h).

Names follow the reference (src/CameraCalibration.{h,cpp}):
    performBundleAdjustment()  <- CameraCalibration::performBundleAdjustment  (:774-992)
    calcReprojectionError()    <- CameraCalibration::calcReprojectionError    (:1026-1103)
    make_config()              <- the config bitmask assembly                  (:778-814)
The arithmetic lives in the HIP library; this file only flattens arguments and forwards them.
There is no Python/CPU fallback: if the library is missing, loading raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _capi as capi


def make_config(nRadialDistParam=2, tangentialDistParam=True, refinePoses=True, useRobustCostFunction=True,
                refine3Dpoints=True, mlCenterAdjustment=True) -> int:
    """reference src/CameraCalibration.cpp:778-814 (nRadialDistParam is clamped to 2 at :786)."""
    cfg = min(int(nRadialDistParam), 2) & 0x3
    if tangentialDistParam:
        cfg |= 0x004
    if refinePoses:
        cfg |= 0x100
    if useRobustCostFunction:
        cfg |= 0x200
    if refine3Dpoints:
        cfg |= 0x400
    if mlCenterAdjustment:
        cfg |= 0x800
    return cfg


class LifcalError(RuntimeError):
    pass


def _check(lib, rc, what):
    if rc != 0:
        raise LifcalError(f"{what}: {lib.lifcal_ba_strerror(rc).decode()} ({rc}) {lib.lifcal_ba_last_error().decode()}")


class BundleAdjustment:
    """One bundle-adjustment problem resident on one MI355X (one rank of a point-sharded job)."""

    def __init__(self, problem: capi.ProblemArrays, options: Optional[capi.Options] = None, partition: Optional["capi.PartitionArrays"] = None):
        """partition: `problem` is the rank's SHARD (only the observations of the points it owns, see capi.PartitionArrays.shard_of)
        and the handle is made with lifcal_ba_create_shard; None: the whole problem, lifcal_ba_create."""
        self.lib = capi.load_library()
        self.problem = problem
        if options is None:
            options = capi.Options()
            self.lib.lifcal_ba_default_options(C.byref(options))
        self.options = options
        self._h = C.c_void_p()
        self._hook = None
        self._partition = partition
        if partition is not None:
            _check(self.lib, self.lib.lifcal_ba_create_shard(C.byref(problem.struct), C.byref(partition.struct), C.byref(options), C.byref(self._h)), "lifcal_ba_create_shard")
        else:
            _check(self.lib, self.lib.lifcal_ba_create(C.byref(problem.struct), C.byref(options), C.byref(self._h)), "lifcal_ba_create")

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if self._h:
            self.lib.lifcal_ba_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- the reference's two entry points ------------------------------------------------------
    def performBundleAdjustment(self) -> capi.Summary:
        """Runs LM to termination; cam/views/pts of `problem` are updated in place (reference :965-988)."""
        s = capi.Summary()
        _check(self.lib, self.lib.lifcal_ba_solve(self._h, C.byref(s)), "lifcal_ba_solve")
        return s

    def calcReprojectionError(self, inlierThreshold: float = 1.0) -> capi.Stats:
        st = capi.Stats()
        _check(self.lib, self.lib.lifcal_ba_reproj_stats(self._h, inlierThreshold, C.byref(st)), "lifcal_ba_reproj_stats")
        return st

    def projectObservations(self):
        """(x_proj, y_proj) per observation at the stored parameters, in the caller's observation order — the projected
        columns of reference storeRawImagePointsCsv (src/CameraCalibration.cpp:1504-1538)."""
        n = self.problem.struct.n_obs
        x = np.zeros(n); y = np.zeros(n)
        _check(self.lib, self.lib.lifcal_ba_project_observations(self._h, capi.as_dptr(x), capi.as_dptr(y)), "lifcal_ba_project_observations")
        return x, y

    def set_fixed_frames(self, fixed=None):
        """hold the poses of the frames with fixed[f] != 0 constant in the following sweeps / solves (None frees all)"""
        if fixed is None:
            _check(self.lib, self.lib.lifcal_ba_set_fixed_frames(self._h, None), "lifcal_ba_set_fixed_frames")
        else:
            m = np.ascontiguousarray(fixed, np.uint8)
            assert len(m) == self.problem.struct.n_frames
            _check(self.lib, self.lib.lifcal_ba_set_fixed_frames(self._h, m.ctypes.data_as(C.POINTER(C.c_uint8))), "lifcal_ba_set_fixed_frames")

    # -- the benchmarked unit ------------------------------------------------------------------
    def sweep(self, radius: float = 1e4, want_matrices: bool = False):
        """One Jacobian+Schur sweep; returns a namespace with cost, gradient_max_norm, seconds and,
        if requested, S / rhs / gradient_reduced / point_gradient / point_hessian_inv (canonical order)."""
        info = self.info()
        out = capi.SweepOut()
        res = type("Sweep", (), {})()
        if want_matrices:
            n = info.n_reduced
            P = self.problem.struct.n_points
            res.S = np.zeros((n, n)); res.rhs = np.zeros(n); res.gradient_reduced = np.zeros(n)
            res.point_gradient = np.zeros(3 * P); res.point_hessian_inv = np.zeros(9 * P)
            out.S, out.rhs, out.gradient_reduced = capi.as_dptr(res.S), capi.as_dptr(res.rhs), capi.as_dptr(res.gradient_reduced)
            out.point_gradient, out.point_hessian_inv = capi.as_dptr(res.point_gradient), capi.as_dptr(res.point_hessian_inv)
        _check(self.lib, self.lib.lifcal_ba_sweep(self._h, float(radius), C.byref(out)), "lifcal_ba_sweep")
        res.cost = out.cost; res.gradient_max_norm = out.gradient_max_norm; res.seconds = out.seconds
        res.n_reduced = out.n_reduced; res.n_promoted = out.n_promoted
        return res

    def sweep_enqueue(self, radius: float = 1e4):
        """Enqueue one sweep on the handle's stream without a host round trip (timing loops)."""
        rc = self.lib.lifcal_ba_sweep_enqueue(self._h, float(radius))
        if rc:
            _check(self.lib, rc, "lifcal_ba_sweep_enqueue")

    def profile_begin(self, max_sweeps: int, stride: int = 1):
        """The next max_sweeps sweeps form a profiled span; every stride-th one carries the dominant kernel's own time stamps
        (a stamped launch costs ~5 us of queue time: timing loops sample, lifcal_ba_profile_begin_sampled)."""
        _check(self.lib, self.lib.lifcal_ba_profile_begin_sampled(self._h, int(max_sweeps), int(stride)), "lifcal_ba_profile_begin_sampled")

    def profile_end(self) -> capi.Profile:
        p = capi.Profile()
        _check(self.lib, self.lib.lifcal_ba_profile_end(self._h, C.byref(p)), "lifcal_ba_profile_end")
        return p

    # -- plumbing --------------------------------------------------------------------------------
    def upload_parameters(self):
        _check(self.lib, self.lib.lifcal_ba_upload_parameters(self._h), "lifcal_ba_upload_parameters")

    def download_parameters(self):
        _check(self.lib, self.lib.lifcal_ba_download_parameters(self._h), "lifcal_ba_download_parameters")

    def info(self) -> capi.Info:
        i = capi.Info()
        _check(self.lib, self.lib.lifcal_ba_get_info(self._h, C.byref(i)), "lifcal_ba_get_info")
        return i

    def set_allreduce(self, fn):
        """fn(device_ptr:int, count:int, stream:int) -> 0 on success; must sum-reduce in place."""
        def tramp(ctx, buf, count, stream):
            try:
                return int(fn(int(buf or 0), int(count), int(stream or 0)))
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        self._hook = capi.ALLREDUCE_FN(tramp)
        _check(self.lib, self.lib.lifcal_ba_set_allreduce(self._h, self._hook, None), "lifcal_ba_set_allreduce")

    def set_allgather(self, fn):
        """fn(send_ptr:int, recv_ptr:int, count_per_rank:int, stream:int) -> 0 on success; recv is rank-major."""
        def tramp(ctx, send, recv, count, stream):
            try:
                return int(fn(int(send or 0), int(recv or 0), int(count), int(stream or 0)))
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        self._ghook = capi.ALLGATHER_FN(tramp)
        _check(self.lib, self.lib.lifcal_ba_set_allgather(self._h, self._ghook, None), "lifcal_ba_set_allgather")

    def comm_init_rccl(self, unique_id: bytes):
        buf = C.create_string_buffer(unique_id, 128)
        _check(self.lib, self.lib.lifcal_ba_comm_init_rccl(self._h, buf), "lifcal_ba_comm_init_rccl")


def performBundleAdjustmentWindowed(problem: capi.ProblemArrays, window_frames: int, overlap_frames: int, options: Optional[capi.Options] = None,
                                    comm_template: Optional[BundleAdjustment] = None):
    """Frame-windowed ("streaming") solve of a long sequence (lifcal_ba_solve_windowed): parameters of `problem` are updated in
    place, one window resident on the device at a time.  Returns the list of per-window reports."""
    lib = capi.load_library()
    if options is None:
        options = capi.Options(); lib.lifcal_ba_default_options(C.byref(options))
    step = window_frames - overlap_frames
    cap = max(1, (problem.struct.n_frames + step - 1) // max(step, 1) + 1)
    reps = (capi.WindowReport * cap)()
    n = C.c_uint32(cap)
    _check(lib, lib.lifcal_ba_solve_windowed(C.byref(problem.struct), C.byref(options), int(window_frames), int(overlap_frames),
                                             comm_template._h if comm_template is not None else None, reps, C.byref(n)), "lifcal_ba_solve_windowed")
    return [reps[i] for i in range(n.value)]


def comm_unique_id() -> bytes:
    lib = capi.load_library()
    buf = C.create_string_buffer(128)
    _check(lib, lib.lifcal_ba_comm_unique_id(buf), "lifcal_ba_comm_unique_id")
    return buf.raw


def initPlenopticParameters(vdepth, fr, pt, world_to_cam, pts, fL_init, device: int = 0):
    """Start values (B_init, bL0_init) of the plenoptic parameters — reference CameraCalibration::initPlenopticParameters
    (src/CameraCalibration.cpp:456-499).  vdepth / fr / pt: one entry per image point (virtual depth, frame, object point);
    world_to_cam: (F, 4, 4) matrices; pts: (P, 3); fL_init = fPH_init * pixelSize_totFoc.  Returns capi.InitResult."""
    lib = capi.load_library()
    arrs = capi.InitArrays(vdepth, fr, pt, world_to_cam, pts, fL_init)
    res = capi.InitResult()
    _check(lib, lib.lifcal_init_plenoptic(C.byref(arrs.struct), int(device), C.byref(res)), "lifcal_init_plenoptic")
    return res


def plan(problem: capi.ProblemArrays, rank: int = 0, world_size: int = 1):
    """Host-only layout planning (no GPU needed): returns (PlanInfo, obs_order, point_owner)."""
    lib = capi.load_library()
    info = capi.PlanInfo()
    order = np.zeros(max(problem.struct.n_obs, 1), np.uint32)
    owner = np.zeros(max(problem.struct.n_points, 1), np.uint32)
    _check(lib, lib.lifcal_ba_plan(C.byref(problem.struct), rank, world_size, C.byref(info), capi.as_uptr(order), capi.as_uptr(owner)), "lifcal_ba_plan")
    return info, order[: problem.struct.n_obs], owner[: problem.struct.n_points].astype(np.int32)
