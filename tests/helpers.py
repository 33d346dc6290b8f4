"""Shared helpers for the test-suite (scene variants, comparison metrics)."""
import numpy as np

from lifcal_amd import _capi as capi, scene

S = scene.SceneSpec

# (name, spec): every template instantiation of the device model + the three functor arities
SMALL_CASES = [
    ("r2_tan_full", S(6, 40, None, 0x506, 101)),
    ("r2_tan_adj_robust", S(6, 40, None, 0xF06, 102, outlier_fraction=0.05)),
    ("r1_adj", S(5, 30, None, 0xD01, 103)),
    ("r1_tan", S(5, 30, None, 0x505, 104)),
    ("r0", S(5, 30, None, 0x500, 105)),
    ("r0_adj", S(5, 30, None, 0xD00, 106)),
    ("r0_tan_adj", S(5, 30, None, 0xD04, 107)),
    ("r2_only", S(5, 30, None, 0x502, 108)),
    ("r2_adj_robust", S(5, 30, None, 0xF02, 109, outlier_fraction=0.05)),
    ("r0_tan", S(5, 30, None, 0x504, 110)),
    ("r1_tan_adj", S(5, 30, None, 0xD05, 111)),
    ("r1", S(5, 30, None, 0x501, 112)),
    ("camera_only", S(6, 40, None, 0x006, 113)),            # arity <2,17>
    ("camera_only_robust_adj", S(6, 40, None, 0xA06, 114)),
    ("poses_only", S(6, 40, None, 0x306, 115)),             # arity <2,17,6>
    ("points_flag_without_poses", S(5, 30, None, 0x406, 116)),  # reference: still camera only
    ("constraints", S(6, 40, None, 0x506, 117, n_constraints=3)),
    ("constraints_adj_robust", S(6, 40, None, 0xF06, 118, n_constraints=4, outlier_fraction=0.03)),
    ("windowed", S(24, 120, 6, 0xF06, 119, outlier_fraction=0.02)),
    ("recalib", S(8, 60, None, 0xF06, 120, recalib=True, outlier_fraction=0.02)),
]


def problem(sc, initial=True):
    return capi.ProblemArrays.from_scene(sc, initial=initial)


def bounded_problem(sc):
    """the recalib pattern with a tight box: fL and B fixed, bL0 / cx / cy boxed so closely that the projected LM step
    fails ceres' Armijo test and TrustRegionMinimizer::DoLineSearch backtracks (checked on the oracle with LO_DEBUG_LS=1)"""
    lower = np.full(17, -np.inf); upper = np.full(17, np.inf)
    for k in (1, 3, 4):
        lower[k] = sc.cam0[k] - 0.05 * abs(sc.cam_gt[k] - sc.cam0[k]) - 1e-3
        upper[k] = sc.cam0[k] + 0.05 * abs(sc.cam_gt[k] - sc.cam0[k]) + 1e-3
    return capi.ProblemArrays(sc.u, sc.v, sc.mcx, sc.mcy, sc.pt, sc.fr, sc.cam0, sc.views0, sc.pts0, sc.spx, sc.scale, sc.config,
                              fixed_mask=0b101, lower=lower, upper=upper)


def oracle_lens_selector(spec):
    """lens_selector for scene.make_scene built on the oracle's restatement of the reference's observation generator
    (oracle/lifcal_mla.cpp: projectPointsToRawImage with the lens maps and the epipolar web, reference src/CameraCalibration.cpp:637-769)"""
    from oracle import mla as omla
    grid = omla.MicroLensGrid(spec.raw_width, spec.raw_height, spec.lens_diameter, spec.lens_base_y, spec.grid_rotation, spec.grid_offset, True)

    def select(img_x, img_y, img_vd, img_fr, img_pt, scale):
        src, mx, my = [], [], []
        fr = np.asarray(img_fr)
        bounds = np.flatnonzero(np.diff(fr)) + 1
        starts = np.concatenate([[0], bounds]); ends = np.concatenate([bounds, [len(fr)]])
        for a, b in zip(starts, ends):
            if b <= a:
                continue
            o = grid.project_frame(img_x[a:b], img_y[a:b], img_vd[a:b], int(scale))
            src.append(o.point + a); mx.append(o.cX); my.append(o.cY)
        if not src:
            return np.zeros(0, np.int64), np.zeros(0), np.zeros(0)
        return np.concatenate(src), np.concatenate(mx), np.concatenate(my)
    return select


def free_port():
    """a TCP port the kernel just handed out (rendezvous of the multi-process tests)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def scaled_max_err(A, B, diag=None):
    """max |A-B| relative to sqrt(d_i d_j) with d the diagonal of B (block-scaled matrix comparison)."""
    d = np.sqrt(np.abs(np.diag(B if diag is None else diag))) + 1e-300
    return float(np.max(np.abs(A - B) / np.outer(d, d)))


def vec_err(a, b):
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


def load_golden(name):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    lower = g["lower"] if g["lower"].size else None
    upper = g["upper"] if g["upper"].size else None
    pa = capi.ProblemArrays(g["u"], g["v"], g["mcx"], g["mcy"], g["pt"], g["fr"], g["cam"], g["views"], g["pts"],
                            float(g["spx"]), float(g["scale"]), int(g["config"]), fixed_mask=int(g["fixed_mask"]),
                            lower=lower, upper=upper, c_i=g["c_i"], c_j=g["c_j"], c_dist=g["c_dist"], c_sigma=g["c_sigma"],
                            use_constraints=int(g["use_constraints"]))
    return g, pa


GOLDEN = ["cfg506", "cfgF06_robust_adj", "cfg006_camera_only", "cfg306_poses_only", "cfg506_constraints",
          "cfgF06_windowed", "cfgF06_recalib", "cfgD01"]
