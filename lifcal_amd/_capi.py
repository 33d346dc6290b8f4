"""ctypes view of include/lifcal_ba.h (struct layouts + prototypes).  Plumbing only."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LIFCAL_BA_LIB") or os.path.join(_HERE, "csrc", "liblifcal_ba.so")   # (LIFCAL_BA_LIB: a development build, e.g. make dev / make stamps)

dptr = C.POINTER(C.c_double)
uptr = C.POINTER(C.c_uint32)


class Problem(C.Structure):
    _fields_ = [
        ("n_obs", C.c_uint32), ("n_frames", C.c_uint32), ("n_points", C.c_uint32), ("n_constraints", C.c_uint32),
        ("u", dptr), ("v", dptr), ("mcx", dptr), ("mcy", dptr), ("pt", uptr), ("fr", uptr),
        ("cam", dptr), ("views", dptr), ("pts", dptr),
        ("spx", C.c_double), ("spy", C.c_double), ("scale", C.c_double),
        ("config", C.c_uint32), ("fixed_mask", C.c_uint32),
        ("lower", dptr), ("upper", dptr),
        ("c_i", uptr), ("c_j", uptr), ("c_dist", dptr), ("c_sigma", dptr),
        ("use_constraints", C.c_uint32), ("reserved", C.c_uint32),
    ]


class Options(C.Structure):
    _fields_ = [
        ("function_tolerance", C.c_double), ("parameter_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
        ("initial_radius", C.c_double), ("max_radius", C.c_double), ("min_radius", C.c_double),
        ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
        ("loss_scale", C.c_double),
        ("max_iterations", C.c_int32), ("jacobi_scaling", C.c_int32), ("precision", C.c_int32), ("device", C.c_int32),
        ("rank", C.c_int32), ("world_size", C.c_int32), ("verbose", C.c_int32), ("deterministic", C.c_int32),
    ]


class Summary(C.Structure):
    _fields_ = [
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("final_radius", C.c_double),
        ("final_gradient_max_norm", C.c_double),
        ("iterations", C.c_int32), ("successful_steps", C.c_int32), ("unsuccessful_steps", C.c_int32),
        ("termination", C.c_int32),
        ("seconds_total", C.c_double), ("seconds_sweep", C.c_double), ("seconds_linear_solve", C.c_double),
    ]


class SweepOut(C.Structure):
    _fields_ = [
        ("cost", C.c_double), ("gradient_max_norm", C.c_double),
        ("n_reduced", C.c_uint32), ("n_promoted", C.c_uint32),
        ("S", dptr), ("rhs", dptr), ("gradient_reduced", dptr), ("point_gradient", dptr), ("point_hessian_inv", dptr),
        ("seconds", C.c_double),
    ]


class Stats(C.Structure):
    _fields_ = [("std_x", C.c_double), ("std_y", C.c_double), ("mae_x", C.c_double), ("mae_y", C.c_double),
                ("num_points", C.c_uint32), ("num_inliers", C.c_uint32)]


class Info(C.Structure):
    _fields_ = [("n_obs_local", C.c_uint32), ("n_points_local", C.c_uint32), ("n_groups", C.c_uint32),
                ("n_tiles", C.c_uint32), ("n_lenses", C.c_uint32), ("n_reduced", C.c_uint32), ("n_promoted", C.c_uint32),
                ("n_chunks", C.c_uint32), ("max_window_frames", C.c_uint32),
                ("device_bytes", C.c_uint64), ("stream", C.c_void_p)]


class Profile(C.Structure):
    _fields_ = [("n_sweeps", C.c_uint32), ("special_points", C.c_double), ("ms_accumulate", C.c_double),
                ("ms_schur", C.c_double), ("ms_total", C.c_double), ("ms_exchange", C.c_double), ("n_sampled", C.c_uint32)]


class WindowReport(C.Structure):   # lifcal_ba_window_report
    _fields_ = [("first_frame", C.c_uint32), ("n_frames", C.c_uint32), ("n_fixed_frames", C.c_uint32), ("n_points", C.c_uint32), ("n_obs", C.c_uint32),
                ("n_dropped_constraints", C.c_uint32), ("summary", Summary)]


class Partition(C.Structure):      # lifcal_ba_partition
    _fields_ = [("world_size", C.c_uint32), ("n_frames", C.c_uint32), ("n_points", C.c_uint32), ("band_width", C.c_uint32), ("n_obs", C.c_uint64),
                ("point_owner", C.POINTER(C.c_int32)), ("rank_first", uptr), ("rank_frames", uptr), ("rank_obs", C.POINTER(C.c_uint64)), ("frame_used", C.POINTER(C.c_uint8))]


class PartitionArrays:
    """Owns the arrays of a lifcal_ba_partition filled by lifcal_ba_partition_points."""

    def __init__(self, problem: "ProblemArrays", world_size: int):
        lib = load_library()
        F, P = problem.struct.n_frames, problem.struct.n_points
        self.point_owner = np.zeros(max(P, 1), np.int32); self.rank_first = np.zeros(world_size, np.uint32); self.rank_frames = np.zeros(world_size, np.uint32)
        self.rank_obs = np.zeros(world_size, np.uint64); self.frame_used = np.zeros(max(F, 1), np.uint8)
        self.struct = Partition(world_size, 0, 0, 0, 0, self.point_owner.ctypes.data_as(C.POINTER(C.c_int32)), as_uptr(self.rank_first), as_uptr(self.rank_frames),
                                self.rank_obs.ctypes.data_as(C.POINTER(C.c_uint64)), self.frame_used.ctypes.data_as(C.POINTER(C.c_uint8)))
        rc = lib.lifcal_ba_partition_points(C.byref(problem.struct), C.byref(self.struct))
        if rc:
            raise RuntimeError(f"lifcal_ba_partition_points: {rc}")

    def shard_of(self, problem: "ProblemArrays", rank: int) -> "ProblemArrays":
        """the rank's local problem: only the observations of the points it owns (full parameter arrays)"""
        sel = np.flatnonzero(self.point_owner[problem.pt] == rank)
        return ProblemArrays(problem.u[sel], problem.v[sel], problem.mcx[sel], problem.mcy[sel], problem.pt[sel], problem.fr[sel], problem.cam, problem.views, problem.pts,
                             problem.struct.spx, problem.struct.scale, problem.struct.config, spy=problem.struct.spy, fixed_mask=problem.struct.fixed_mask,
                             lower=problem.lower, upper=problem.upper, use_constraints=0)


class PlanInfo(C.Structure):
    _fields_ = [("n_groups", C.c_uint32), ("n_tiles", C.c_uint32), ("n_lenses", C.c_uint32), ("n_promoted", C.c_uint32),
                ("n_reduced", C.c_uint32), ("max_group_obs", C.c_uint32), ("n_chunks", C.c_uint32),
                ("max_window_frames", C.c_uint32)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

class InitProblem(C.Structure):
    _fields_ = [("n", C.c_uint64), ("vdepth", C.POINTER(C.c_double)), ("fr", C.POINTER(C.c_uint32)), ("pt", C.POINTER(C.c_uint32)),
                ("n_frames", C.c_uint32), ("n_points", C.c_uint32), ("world_to_cam", C.POINTER(C.c_double)), ("pts", C.POINTER(C.c_double)),
                ("fL_init", C.c_double)]


class InitResult(C.Structure):
    _fields_ = [("B_init", C.c_double), ("bL0_init", C.c_double), ("n_used", C.c_uint64), ("rank", C.c_int32), ("reserved", C.c_int32)]


class InitArrays:
    """Owns contiguous copies of the inputs of lifcal_init_plenoptic / lo_init_plenoptic."""

    def __init__(self, vdepth, fr, pt, world_to_cam, pts, fL_init):
        self.vdepth = np.ascontiguousarray(vdepth, np.float64)
        self.fr = np.ascontiguousarray(fr, np.uint32)
        self.pt = np.ascontiguousarray(pt, np.uint32)
        # (F, 4, 4) matrices in mathematical (row, column) indexing -> Eigen's column-major storage
        w = np.asarray(world_to_cam, np.float64).reshape(-1, 4, 4)
        self.w2c = np.ascontiguousarray(np.transpose(w, (0, 2, 1)).reshape(-1, 16))
        self.pts = np.ascontiguousarray(np.asarray(pts, np.float64).reshape(-1, 3))
        assert len(self.vdepth) == len(self.fr) == len(self.pt)
        self.struct = InitProblem(len(self.vdepth), as_dptr(self.vdepth), as_uptr(self.fr), as_uptr(self.pt), len(self.w2c), len(self.pts),
                                  as_dptr(self.w2c), as_dptr(self.pts), float(fL_init))


class MlaParams(C.Structure):      # include/lifcal_mla.h lifcal_mla_params
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("lens_diameter", C.c_float), ("lens_base_y", C.c_float * 2),
                ("rotation", C.c_float), ("offset", C.c_float * 2), ("rotation_on_grid", C.c_int32)]


class MlaPoints(C.Structure):      # lifcal_mla_points
    _fields_ = [("n", C.c_uint64), ("x", dptr), ("y", dptr), ("vdepth", dptr), ("fr", uptr), ("pt", uptr)]


class MlaObservations(C.Structure):  # lifcal_mla_observations
    _fields_ = [("capacity", C.c_uint64), ("n_obs", C.c_uint64), ("u", dptr), ("v", dptr), ("mcx", dptr), ("mcy", dptr),
                ("src", uptr), ("fr", uptr), ("pt", uptr)]


MLA_MORE = 1
_fptr = C.POINTER(C.c_float)
_iptr = C.POINTER(C.c_int32)

# every symbol include/lifcal_mla.h declares
MLA_PROTOTYPES = {
    "lifcal_mla_create": (C.c_int, [C.POINTER(MlaParams), C.c_int32, C.POINTER(C.c_void_p)]),
    "lifcal_mla_destroy": (None, [C.c_void_p]),
    "lifcal_mla_info": (C.c_int, [C.c_void_p, _iptr, _iptr, _iptr]),
    "lifcal_mla_get_lenses": (C.c_int, [C.c_void_p, _fptr, _fptr, _iptr]),
    "lifcal_mla_get_maps": (C.c_int, [C.c_void_p, _iptr, _iptr]),
    "lifcal_mla_get_web": (C.c_int, [C.c_void_p, dptr, dptr, dptr, _iptr]),
    "lifcal_mla_project": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(MlaPoints), C.POINTER(MlaObservations)]),
}

class CameraModel(C.Structure):    # include/lifcal_io.h lifcal_camera_model
    _fields_ = [("image_width", C.c_int32), ("image_height", C.c_int32), ("pixel_size", C.c_double),
                ("fL", C.c_double), ("bL0", C.c_double), ("B", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("n_radial", C.c_int32), ("radial", C.c_double * 8), ("tangential", C.c_int32), ("tangential_dist", C.c_double * 2),
                ("ml_center_adjustment", C.c_int32)]


class Protocol(C.Structure):       # lifcal_protocol
    _fields_ = [("model", CameraModel), ("refine_poses", C.c_int32), ("refine_points", C.c_int32), ("robust_cost", C.c_int32),
                ("std_x", C.c_double), ("std_y", C.c_double), ("mae_x", C.c_double), ("mae_y", C.c_double)]


# every symbol include/lifcal_io.h declares
IO_PROTOTYPES = {
    "lifcal_write_camera_model": (C.c_int, [C.c_char_p, C.POINTER(CameraModel)]),
    "lifcal_write_extrinsic_orientations_xml": (C.c_int, [C.c_char_p, C.c_uint32, _iptr, dptr]),
    "lifcal_write_extrinsic_orientations_txt": (C.c_int, [C.c_char_p, C.c_uint32, _iptr, dptr]),
    "lifcal_write_raw_image_points_csv": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint32, _iptr, uptr, dptr, dptr, dptr, dptr, uptr]),
    "lifcal_write_protocol": (C.c_int, [C.c_char_p, C.POINTER(Protocol)]),
}

class ColmapInfo(C.Structure):     # include/lifcal_colmap.h lifcal_colmap_info
    _fields_ = [("n_frames", C.c_uint32), ("n_points", C.c_uint32), ("n_image_points", C.c_uint64), ("binary", C.c_int32),
                ("camera_model_id", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("reserved", C.c_int32),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("k1", C.c_double), ("k2", C.c_double), ("p1", C.c_double), ("p2", C.c_double), ("f", C.c_double)]


# every symbol include/lifcal_colmap.h declares
COLMAP_PROTOTYPES = {
    "lifcal_colmap_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "lifcal_colmap_get_info": (C.c_int, [C.c_void_p, C.POINTER(ColmapInfo)]),
    "lifcal_colmap_get_frames": (C.c_int, [C.c_void_p, _iptr, dptr, dptr, dptr]),
    "lifcal_colmap_get_points": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), dptr]),
    "lifcal_colmap_get_image_points": (C.c_int, [C.c_void_p, dptr, dptr, uptr, uptr]),
    "lifcal_colmap_free": (None, [C.c_void_p]),
}

# every symbol include/lifcal_ba.h declares: name -> (restype, argtypes)
PROTOTYPES = {
    "lifcal_ba_default_options": (None, [C.POINTER(Options)]),
    "lifcal_ba_create": (C.c_int, [C.POINTER(Problem), C.POINTER(Options), C.POINTER(C.c_void_p)]),
    "lifcal_ba_solve": (C.c_int, [C.c_void_p, C.POINTER(Summary)]),
    "lifcal_ba_sweep": (C.c_int, [C.c_void_p, C.c_double, C.POINTER(SweepOut)]),
    "lifcal_ba_sweep_enqueue": (C.c_int, [C.c_void_p, C.c_double]),
    "lifcal_ba_profile_begin": (C.c_int, [C.c_void_p, C.c_uint32]),
    "lifcal_ba_profile_begin_sampled": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "lifcal_ba_profile_end": (C.c_int, [C.c_void_p, C.POINTER(Profile)]),
    "lifcal_ba_reproj_stats": (C.c_int, [C.c_void_p, C.c_double, C.POINTER(Stats)]),
    "lifcal_ba_project_observations": (C.c_int, [C.c_void_p, dptr, dptr]),
    "lifcal_ba_set_fixed_frames": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint8)]),
    "lifcal_ba_solve_windowed": (C.c_int, [C.POINTER(Problem), C.POINTER(Options), C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(WindowReport), C.POINTER(C.c_uint32)]),
    "lifcal_ba_upload_parameters": (C.c_int, [C.c_void_p]),
    "lifcal_ba_download_parameters": (C.c_int, [C.c_void_p]),
    "lifcal_ba_set_allreduce": (C.c_int, [C.c_void_p, ALLREDUCE_FN, C.c_void_p]),
    "lifcal_ba_set_allgather": (C.c_int, [C.c_void_p, ALLGATHER_FN, C.c_void_p]),
    "lifcal_ba_comm_unique_id": (C.c_int, [C.c_void_p]),
    "lifcal_ba_comm_init_rccl": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lifcal_ba_get_info": (C.c_int, [C.c_void_p, C.POINTER(Info)]),
    "lifcal_ba_destroy": (None, [C.c_void_p]),
    "lifcal_init_plenoptic": (C.c_int, [C.POINTER(InitProblem), C.c_int32, C.POINTER(InitResult)]),
    "lifcal_init_plenoptic_recalibration": (C.c_int, [C.c_double, C.c_double, C.POINTER(InitResult)]),
    "lifcal_ba_strerror": (C.c_char_p, [C.c_int]),
    "lifcal_ba_last_error": (C.c_char_p, []),
    "lifcal_ba_version": (C.c_char_p, []),
    "lifcal_ba_plan": (C.c_int, [C.POINTER(Problem), C.c_int32, C.c_int32, C.POINTER(PlanInfo), uptr, uptr]),
    "lifcal_ba_partition_points": (C.c_int, [C.POINTER(Problem), C.POINTER(Partition)]),
    "lifcal_ba_create_shard": (C.c_int, [C.POINTER(Problem), C.POINTER(Partition), C.POINTER(Options), C.POINTER(C.c_void_p)]),
    "lifcal_ba_plan_shard": (C.c_int, [C.POINTER(Problem), C.POINTER(Partition), C.c_int32, C.POINTER(PlanInfo)]),
}

_lib = None


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """Load the HIP extension.  Fails loudly: there is no Python/CPU fallback for the hot path."""
    global _lib
    if _lib is None:
        if not os.path.exists(path):
            raise RuntimeError(
                f"lifcal_amd: native library {path} is missing. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for the bundle-adjustment path.")
        lib = C.CDLL(path)
        for name, (res, args) in list(PROTOTYPES.items()) + list(MLA_PROTOTYPES.items()) + list(IO_PROTOTYPES.items()) + list(COLMAP_PROTOTYPES.items()):
            fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def as_dptr(a: np.ndarray):
    return a.ctypes.data_as(dptr) if a is not None else None


def as_uptr(a: np.ndarray):
    return a.ctypes.data_as(uptr) if a is not None else None


class ProblemArrays:
    """Owns contiguous numpy buffers and the ctypes struct that points into them."""

    def __init__(self, u, v, mcx, mcy, pt, fr, cam, views, pts, spx, scale, config, spy=None,
                 fixed_mask=0, lower=None, upper=None, c_i=None, c_j=None, c_dist=None, c_sigma=None,
                 use_constraints=1):
        f8 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
        u4 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.uint32).reshape(-1))
        self.u, self.v, self.mcx, self.mcy = f8(u), f8(v), f8(mcx), f8(mcy)
        self.pt, self.fr = u4(pt), u4(fr)
        self.cam = f8(cam).copy()
        self.views = f8(views).copy()
        self.pts = f8(pts).copy()
        assert self.cam.shape[0] == 17
        self.lower = f8(lower) if lower is not None else None
        self.upper = f8(upper) if upper is not None else None
        m = 0 if c_i is None else len(np.asarray(c_i).reshape(-1))
        self.c_i = u4(c_i) if m else None
        self.c_j = u4(c_j) if m else None
        self.c_dist = f8(c_dist) if m else None
        self.c_sigma = f8(c_sigma) if m else None
        p = Problem()
        p.n_obs = self.u.shape[0]
        p.n_frames = self.views.shape[0] // 6
        p.n_points = self.pts.shape[0] // 3
        p.n_constraints = m
        p.u, p.v, p.mcx, p.mcy = as_dptr(self.u), as_dptr(self.v), as_dptr(self.mcx), as_dptr(self.mcy)
        p.pt, p.fr = as_uptr(self.pt), as_uptr(self.fr)
        p.cam, p.views, p.pts = as_dptr(self.cam), as_dptr(self.views), as_dptr(self.pts)
        p.spx = float(spx)
        p.spy = float(spx if spy is None else spy)
        p.scale = float(scale)
        p.config = int(config)
        p.fixed_mask = int(fixed_mask)
        p.lower = as_dptr(self.lower) if self.lower is not None else None
        p.upper = as_dptr(self.upper) if self.upper is not None else None
        p.c_i = as_uptr(self.c_i) if m else None
        p.c_j = as_uptr(self.c_j) if m else None
        p.c_dist = as_dptr(self.c_dist) if m else None
        p.c_sigma = as_dptr(self.c_sigma) if m else None
        p.use_constraints = int(use_constraints)
        self.struct = p

    @classmethod
    def from_scene(cls, scene, initial=True):
        return cls(scene.u, scene.v, scene.mcx, scene.mcy, scene.pt, scene.fr,
                   scene.cam0 if initial else scene.cam_gt,
                   scene.views0 if initial else scene.views_gt,
                   scene.pts0 if initial else scene.pts_gt,
                   scene.spx, scene.scale, scene.config, fixed_mask=scene.fixed_mask,
                   lower=scene.lower, upper=scene.upper, c_i=scene.c_i, c_j=scene.c_j,
                   c_dist=scene.c_dist, c_sigma=scene.c_sigma, use_constraints=scene.use_constraints)


def default_options_py() -> Options:
    """The same defaults lifcal_ba_default_options() writes (used when the .so is not needed)."""
    o = Options()
    o.function_tolerance = 1e-6
    o.parameter_tolerance = 1e-8
    o.gradient_tolerance = 1e-10
    o.initial_radius = 1e4
    o.max_radius = 1e16
    o.min_radius = 1e-32
    o.min_relative_decrease = 1e-3
    o.min_lm_diagonal = 1e-6
    o.max_lm_diagonal = 1e32
    o.loss_scale = 0.5
    o.max_iterations = 200
    o.jacobi_scaling = 1
    o.precision = 0
    o.device = 0
    o.rank = 0
    o.world_size = 1
    o.verbose = 0
    o.deterministic = 0
    return o
