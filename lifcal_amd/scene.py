"""Synthetic plenoptic calibration scenes — the inputs bench.py and the tests use.

Everything the solver consumes is produced here in the flattened form of include/lifcal_ba.h:
micro-image observations (u, v), their micro-lens centres (mcx, mcy), point / frame indices, the
camera block, poses and points (ground truth and perturbed initial guess).

What it mirrors of the reference (by behaviour, own code):
  * hex micro-lens grid      <- MicroLensGrid::createGrid   reference src/MicroLensGrid/MicroLensGrid.cpp:186-270
                                (two interleaved rectangular grids, rotated; centres kept as float32
                                 like MicroLens::centerX/Y, reference src/MicroLensGrid/MicroLens.h:22-23)
  * validity radius          <- lensDiameter/2 - 1.0         reference MicroLensGrid.cpp:108-111
  * which lenses see a point <- projectPointsToRawImage      reference src/CameraCalibration.cpp:640-769
                                (virtual depth 2 < v < 20, inside the sensor, inside the validity radius)
  * forward model            <- CameraModel::projectPoint    reference src/CameraModel.h:86-199 (values only)

Random numbers come from a counter-based splitmix64 generator implemented here (never from
implementation-defined library distributions), so a (spec, seed) pair names one scene everywhere.
This module is data generation only: it is neither the product path nor the oracle.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Optional

import numpy as np

MASK64 = (1 << 64) - 1

CFG_TANGENTIAL = 0x004
CFG_REFINE_POSES = 0x100
CFG_ROBUST = 0x200
CFG_REFINE_POINTS = 0x400
CFG_ML_CENTER_ADJ = 0x800


# ------------------------------------------------------------------------------------------------
# portable PRNG
# ------------------------------------------------------------------------------------------------
def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15))
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


class Stream:
    """Counter-based stream: value k of stream (seed, name) is a pure function of (seed, name, k)."""

    def __init__(self, seed: int, stream_id: int):
        base = _splitmix64(np.array([(seed * 0x2545F4914F6CDD1D + stream_id * 0x632BE59BD9B4E019) & MASK64], dtype=np.uint64))[0]
        self._base = int(base)
        self._ctr = 0

    def _raw(self, n: int) -> np.ndarray:
        idx = (np.arange(self._ctr, self._ctr + n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)) + np.uint64(self._base)
        self._ctr += n
        return _splitmix64(idx)

    def uniform(self, n: int, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
        u = (self._raw(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return lo + (hi - lo) * u

    def normal(self, n: int, sigma: float = 1.0) -> np.ndarray:
        u1 = 1.0 - self.uniform(n)  # (0, 1]
        u2 = self.uniform(n)
        return sigma * np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)

    def integers(self, n: int, hi: int) -> np.ndarray:
        return (self._raw(n) % np.uint64(hi)).astype(np.int64)


# ------------------------------------------------------------------------------------------------
# spec
# ------------------------------------------------------------------------------------------------
@dataclasses.dataclass
class SceneSpec:
    n_frames: int
    n_points: int
    window: Optional[int] = None          # frames a point is visible in (None: all frames)
    config: int = 0x506
    seed: int = 20241022
    noise_px: float = 0.1
    outlier_fraction: float = 0.0         # robust configs: fraction of obs replaced by +-5 px outliers
    n_constraints: int = 0
    recalib: bool = False                 # reference :927-953: fix fL,B; box bounds on bL0,cx,cy
    # camera ground truth
    pixel_size: float = 0.0055
    raw_width: int = 2048
    raw_height: int = 2048
    scale: int = 2
    fL: float = 35.0
    B: float = 0.40
    bL0: float = 34.15
    c: tuple = (511.3, 513.9)
    k: tuple = (5e-5, -2e-7)
    p: tuple = (1e-5, -1e-5)
    # MLA
    lens_diameter: float = 23.2
    lens_base_y: tuple = (0.5, 0.8660254)
    grid_rotation: float = 0.005
    grid_offset: tuple = (3.1, -2.4)
    vdepth_range: tuple = (2.3, 3.2)
    # initial-guess perturbation (relative for fL,bL0,B,cx,cy; absolute for poses / points)
    init_rel: tuple = (0.002, 0.002, 0.01, 0.01, 0.01)
    init_rot_deg: float = 0.3
    init_trans_mm: float = 0.5
    init_point_mm: float = 0.5


def baseline_spec(name: str) -> SceneSpec:
    """The BASELINE.json configs and the 1 M-observation metric point."""
    s = 20241022
    if name == "cfg1":   # calib_marker plumbing case: intrinsics only
        return SceneSpec(5, 100, None, 0x006, s + 1)
    if name == "cfg2":   # 20 frames, 500 points, 2 radial + tangential, refine extrinsics + points
        return SceneSpec(20, 500, None, 0x506, s + 2)
    if name == "cfg3":   # 100 frames, 5k points, mlCenterAdj, robust
        return SceneSpec(100, 5000, 10, 0xF06, s + 3, outlier_fraction=0.02)
    if name == "cfg4":   # 1000 frames, 50k points (8 GPUs)
        return SceneSpec(1000, 50000, 10, 0xF06, s + 4, outlier_fraction=0.02)
    if name == "cfg5":   # recalib streaming pose+point refine
        return SceneSpec(2000, 100000, 10, 0xF06, s + 5, outlier_fraction=0.02, recalib=True)
    if name == "metric":  # 1.0 M observations (48.3 micro-image observations per point with this MLA)
        return SceneSpec(334, 20700, 10, 0xF06, s + 6, outlier_fraction=0.02)
    if name == "metric_web":  # 1.0 M observations when the lenses are chosen by the reference's own generator (make_scene(lens_selector=...):
        return SceneSpec(334, 24720, 10, 0xF06, s + 6, outlier_fraction=0.02)   # 4.05 micro images per (point, frame) instead of 4.85)
    if name == "tiny":   # smoke / unit tests
        return SceneSpec(6, 40, None, 0x506, s + 7)
    raise KeyError(name)


# ------------------------------------------------------------------------------------------------
# geometry helpers
# ------------------------------------------------------------------------------------------------
def euler_xyz(a: np.ndarray) -> np.ndarray:
    """R = Rx(a0) Ry(a1) Rz(a2) for an (..., 3) array of angles (reference CameraModel.h:251-254)."""
    c0, s0 = np.cos(a[..., 0]), np.sin(a[..., 0])
    c1, s1 = np.cos(a[..., 1]), np.sin(a[..., 1])
    c2, s2 = np.cos(a[..., 2]), np.sin(a[..., 2])
    R = np.empty(a.shape[:-1] + (3, 3))
    R[..., 0, 0] = c1 * c2
    R[..., 0, 1] = -c1 * s2
    R[..., 0, 2] = s1
    R[..., 1, 0] = c0 * s2 + s0 * s1 * c2
    R[..., 1, 1] = c0 * c2 - s0 * s1 * s2
    R[..., 1, 2] = -s0 * c1
    R[..., 2, 0] = s0 * s2 - c0 * s1 * c2
    R[..., 2, 1] = s0 * c2 + c0 * s1 * s2
    R[..., 2, 2] = c0 * c1
    return R


def make_lens_grid(spec: SceneSpec) -> np.ndarray:
    """Lens centres (n, 2) as float32, following MicroLensGrid::createGrid with rotation applied."""
    f32 = np.float32
    w, h = spec.raw_width, spec.raw_height
    d = f32(spec.lens_diameter)
    by0, by1 = f32(spec.lens_base_y[0]), f32(spec.lens_base_y[1])
    imc = (f32(w) / f32(2) - f32(0.5), f32(h) / f32(2) - f32(0.5))
    off = (f32(spec.grid_offset[0]), f32(spec.grid_offset[1]))
    off_cv = (off[0] + imc[0], -off[1] + imc[1])
    x_min = -imc[0] - off[0] - d / f32(2)
    x_max = imc[0] - off[0] + d / f32(2)
    y_min = -imc[1] - off[1] - d / f32(2)
    y_max = imc[1] - off[1] + d / f32(2)
    pitch_y = f32(2) * by1 * d
    gx1 = (math.ceil(x_min / d), int(x_max / d))
    gy1 = (math.ceil(y_min / pitch_y), int(y_max / pitch_y))
    gx2 = (math.ceil(x_min / d - by0 - f32(1)), int(x_max / d - by0 - f32(1)))
    gy2 = (math.ceil(y_min / pitch_y - f32(0.5)), int(y_max / pitch_y - f32(0.5)))
    ca, sa = f32(math.cos(spec.grid_rotation)), f32(math.sin(spec.grid_rotation))
    out = []
    xs = np.arange(gx1[0], gx1[1] + 1, dtype=np.float32)
    ys = np.arange(gy1[0], gy1[1] + 1, dtype=np.float32)
    X, Y = np.meshgrid(xs * d, ys * d * f32(2) * by1, indexing="ij")
    out.append(np.stack([off_cv[0] + (X * ca - Y * sa), off_cv[1] - (X * sa + Y * ca)], -1).reshape(-1, 2))
    xs = np.arange(gx2[0], gx2[1] + 1, dtype=np.float32)
    ys = np.arange(gy2[0], gy2[1] + 1, dtype=np.float32)
    X, Y = np.meshgrid((xs + f32(1) + by0) * d, ((ys * f32(2) + f32(1)) * by1) * d, indexing="ij")
    out.append(np.stack([off_cv[0] + (X * ca - Y * sa), off_cv[1] - (X * sa + Y * ca)], -1).reshape(-1, 2))
    return np.concatenate(out, 0).astype(np.float32)


def _distortion(x, y, k, p):
    r2 = x * x + y * y
    dr = np.zeros_like(x)
    ri = r2.copy()
    for i, ki in enumerate(k):
        if i > 0:
            ri = ri * r2
        dr = dr + ki * ri
    dx, dy = x * dr, y * dr
    if p is not None:
        dx = dx + p[0] * (r2 + 2.0 * x * x) + 2.0 * p[1] * x * y
        dy = dy + p[1] * (r2 + 2.0 * y * y) + 2.0 * p[0] * x * y
    return dx, dy


def project(pc, ml, cam, config, spx_tot, scale):
    """Vectorised forward model (values only): camera-frame points (n,3) + lens centres (n,2) -> raw px (n,2)."""
    n_rad = config & 3
    tan = bool(config & CFG_TANGENTIAL)
    adj = bool(config & CFG_ML_CENTER_ADJ)
    fL, bL0, B = abs(cam[0]), abs(cam[1]), abs(cam[2])
    sp = spx_tot / scale
    craw = np.abs((np.asarray(cam[3:5]) + 0.5) * scale - 0.5)
    k = [cam[5 + i] for i in range(n_rad)]
    p = [cam[5 + n_rad], cam[6 + n_rad]] if tan else None
    cdx = (ml[:, 0] - craw[0]) * sp
    cdy = (ml[:, 1] - craw[1]) * sp
    cux, cuy = cdx.copy(), cdy.copy()
    if n_rad > 0 or tan:
        for _ in range(10):
            dx, dy = _distortion(cux, cuy, k, p)
            cux, cuy = cdx - dx, cdy - dy
    if adj:
        cux, cuy = cux / (bL0 + B) * bL0, cuy / (bL0 + B) * bL0
    D = fL - bL0
    zc0 = fL * bL0 / D
    zq = pc[:, 2] + zc0
    qx = (pc[:, 0] + cux * fL / D) / zq
    qy = (pc[:, 1] + cuy * fL / D) / zq
    mx = (qx - cux / fL) * fL * B / D
    my = (qy - cuy / fL) * fL * B / D
    if adj:
        px, py = mx + cux, my + cuy
        if n_rad > 0 or tan:
            dx, dy = _distortion(px, py, k, p)
            px, py = px + dx, py + dy
    else:
        px, py = mx + cdx, my + cdy
    return np.stack([px / sp + craw[0], py / sp + craw[1]], -1)


# ------------------------------------------------------------------------------------------------
# scene
# ------------------------------------------------------------------------------------------------
@dataclasses.dataclass
class Scene:
    spec: SceneSpec
    u: np.ndarray
    v: np.ndarray
    mcx: np.ndarray
    mcy: np.ndarray
    pt: np.ndarray
    fr: np.ndarray
    cam_gt: np.ndarray
    views_gt: np.ndarray
    pts_gt: np.ndarray
    cam0: np.ndarray
    views0: np.ndarray
    pts0: np.ndarray
    spx: float
    scale: float
    config: int
    fixed_mask: int
    lower: Optional[np.ndarray]
    upper: Optional[np.ndarray]
    c_i: np.ndarray
    c_j: np.ndarray
    c_dist: np.ndarray
    c_sigma: np.ndarray
    use_constraints: int
    n_lenses: int
    # the virtual-image points the observations were made from (what projectPointsToRawImage consumes), frame-major:
    # image coordinates in virtual-image pixels, virtual depth, frame and object-point index
    img_x: Optional[np.ndarray] = None
    img_y: Optional[np.ndarray] = None
    img_vd: Optional[np.ndarray] = None
    img_fr: Optional[np.ndarray] = None
    img_pt: Optional[np.ndarray] = None

    @property
    def n_obs(self) -> int:
        return int(self.u.shape[0])


def make_scene(spec: SceneSpec, lens_selector=None) -> Scene:
    """lens_selector: None = the K nearest lens centres filtered by the search radius of reference :692 (a numpy stand-in for the
    reference's walk along the epipolar web, which finds ~4 % more lenses than the web does); or a callable
    (img_x, img_y, img_vd, img_fr, img_pt, scale) -> (src, mcx, mcy) that returns, for the frame-major virtual-image points, the lenses
    the reference's own generator visits (CameraCalibration::projectPointsToRawImage, src/CameraCalibration.cpp:637-769) — bench.py
    passes the GPU port of it (lifcal_mla_project), tests pass the oracle's.  Either way the observation VALUES are the forward model
    at ground truth + noise, and an observation is kept if it lies inside its micro image and inside the sensor (:754-759)."""
    F, P = spec.n_frames, spec.n_points
    n_rad = spec.config & 3
    tan = bool(spec.config & CFG_TANGENTIAL)
    spx_tot = spec.pixel_size * spec.scale
    cam_gt = np.zeros(17)
    cam_gt[0:5] = [spec.fL, spec.bL0, spec.B, spec.c[0], spec.c[1]]
    for i in range(n_rad):
        cam_gt[5 + i] = spec.k[i]
    if tan:
        cam_gt[5 + n_rad] = spec.p[0]
        cam_gt[6 + n_rad] = spec.p[1]

    # trajectory: smooth, amplitudes chosen so that windowed / all-visible points stay in view
    all_visible = spec.window is None
    rot_amp = math.radians(2.0 if all_visible else 8.0)
    tr_amp = 60.0 if all_visible else 150.0
    s_traj = Stream(spec.seed, 1)
    ph = s_traj.uniform(6, 0.0, 2.0 * math.pi)
    tt = (np.arange(F) + 0.5) / max(F, 1)
    views_gt = np.zeros((F, 6))
    for kk in range(3):
        views_gt[:, kk] = rot_amp * np.sin(2.0 * math.pi * tt * (1.0 + 0.5 * kk) + ph[kk])
        views_gt[:, 3 + kk] = tr_amp * np.sin(2.0 * math.pi * tt * (1.0 + 0.3 * kk) + ph[3 + kk])
    R_gt = euler_xyz(views_gt[:, :3])

    # points: generated in the frustum of a home frame at depths giving the requested virtual depth
    s_pts = Stream(spec.seed, 2)
    home = (np.arange(P) * F // max(P, 1)).astype(np.int64) if not all_visible else s_pts.integers(P, F)
    if all_visible:
        _ = home  # home frame only positions the point; visibility is tested in every frame
    frac = 0.45 if all_visible else 0.8
    xi = s_pts.uniform(P, -frac, frac)
    yi = s_pts.uniform(P, -frac, frac)
    vd = s_pts.uniform(P, spec.vdepth_range[0], spec.vdepth_range[1])
    bL = spec.bL0 + vd * spec.B
    Z = spec.fL * bL / (bL - spec.fL)
    half_w = 0.5 * (spec.raw_width / spec.scale) * spx_tot
    half_h = 0.5 * (spec.raw_height / spec.scale) * spx_tot
    Xc = xi * half_w / bL * Z
    Yc = yi * half_h / bL * Z
    pc_home = np.stack([Xc, Yc, Z], -1)
    pts_gt = np.einsum("pji,pj->pi", R_gt[home], pc_home - views_gt[home, 3:])

    # candidate (point, frame) pairs
    if all_visible:
        pf_p = np.repeat(np.arange(P), F)
        pf_f = np.tile(np.arange(F), P)
    else:
        w = spec.window
        start = np.clip(home - w // 2, 0, max(F - w, 0))
        pf_p = np.repeat(np.arange(P), min(w, F))
        pf_f = (start[:, None] + np.arange(min(w, F))[None, :]).reshape(-1)
    pc = np.einsum("nij,nj->ni", R_gt[pf_f], pts_gt[pf_p]) + views_gt[pf_f, 3:]
    ok = pc[:, 2] > spec.fL * 1.5
    bLn = spec.fL * pc[:, 2] / np.where(ok, pc[:, 2] - spec.fL, 1.0)
    vdn = (bLn - spec.bL0) / spec.B
    sp_raw = spx_tot / spec.scale
    craw = np.abs((cam_gt[3:5] + 0.5) * spec.scale - 0.5)
    x_ups = craw[0] + pc[:, 0] * bLn / pc[:, 2] / sp_raw
    y_ups = craw[1] + pc[:, 1] * bLn / pc[:, 2] / sp_raw
    ok &= (vdn > 2.0) & (vdn < 20.0)
    ok &= (x_ups >= 0) & (x_ups <= spec.raw_width - 1) & (y_ups >= 0) & (y_ups <= spec.raw_height - 1)
    pf_p, pf_f, pc, x_ups, y_ups, vdn = pf_p[ok], pf_f[ok], pc[ok], x_ups[ok], y_ups[ok], vdn[ok]

    # lenses that can image the point, then the exact model + validity radius
    lenses = make_lens_grid(spec).astype(np.float64)
    n_pf = pf_p.shape[0]
    img = _image_points(pf_p, pf_f, x_ups, y_ups, vdn, spec.scale)
    if lens_selector is None:
        from scipy.spatial import cKDTree

        K = 24
        tree = cKDTree(lenses)
        _, nn = tree.query(np.stack([x_ups, y_ups], -1), k=K)
        cand_pf = np.repeat(np.arange(n_pf), K)
        cand_l = nn.reshape(-1)
        ml = lenses[cand_l]
    else:
        img_order = np.lexsort((pf_p, pf_f))   # frame-major position -> candidate pair (the order of _image_points)
        src, smx, smy = lens_selector(img["img_x"], img["img_y"], img["img_vd"], img["img_fr"], img["img_pt"], spec.scale)
        cand_pf = img_order[np.asarray(src, np.int64)]
        ml = np.stack([np.asarray(smx, np.float64), np.asarray(smy, np.float64)], -1)
    proj = project(pc[cand_pf], ml, cam_gt, spec.config, spx_tot, float(spec.scale))
    d2 = np.sum((proj - ml) ** 2, -1)
    rad_valid = spec.lens_diameter * 0.5 - 1.0
    keep = d2 < rad_valid * rad_valid
    keep &= (proj[:, 0] >= 0) & (proj[:, 0] <= spec.raw_width - 1) & (proj[:, 1] >= 0) & (proj[:, 1] <= spec.raw_height - 1)
    if lens_selector is None:
        # reference :692: the lens must lie within lensDiameter/2 * v + 2 px of the up-sampled point
        search_r = spec.lens_diameter * 0.5 * vdn[cand_pf] + 2.0
        keep &= ((ml[:, 0] - x_ups[cand_pf]) ** 2 + (ml[:, 1] - y_ups[cand_pf]) ** 2) <= search_r * search_r
    cand_pf, ml, proj = cand_pf[keep], ml[keep], proj[keep]
    obs_pt = pf_p[cand_pf]
    obs_fr = pf_f[cand_pf]
    # reference order: frame-major, then the order the points were visited
    order = np.lexsort((obs_pt, obs_fr))
    obs_pt, obs_fr, ml, proj = obs_pt[order], obs_fr[order], ml[order], proj[order]
    N = obs_pt.shape[0]

    s_noise = Stream(spec.seed, 3)
    uv = proj + np.stack([s_noise.normal(N, spec.noise_px), s_noise.normal(N, spec.noise_px)], -1)
    if spec.outlier_fraction > 0:
        s_out = Stream(spec.seed, 4)
        is_out = s_out.uniform(N) < spec.outlier_fraction
        uv[is_out] = proj[is_out] + np.stack([s_out.uniform(N, -5, 5), s_out.uniform(N, -5, 5)], -1)[is_out]

    # initial guess
    s_init = Stream(spec.seed, 5)
    cam0 = np.zeros(17)
    cam0[0:5] = cam_gt[0:5] * (1.0 + np.asarray(spec.init_rel) * s_init.uniform(5, -1.0, 1.0))
    fixed_mask = 0
    lower = upper = None
    if spec.recalib:
        cam0[0] = cam_gt[0]   # fL and B come from the previous calibration (reference :503-512)
        cam0[2] = cam_gt[2]
        fixed_mask = (1 << 0) | (1 << 2)
        lower = np.full(17, -np.inf)
        upper = np.full(17, np.inf)
        for idx in (1, 3, 4):  # reference :943-952
            lower[idx] = 0.7 * cam0[idx]
            upper[idx] = 1.3 * cam0[idx]
    views0 = views_gt.copy()
    views0[:, :3] += s_init.normal(3 * F, math.radians(spec.init_rot_deg)).reshape(F, 3)
    views0[:, 3:] += s_init.normal(3 * F, spec.init_trans_mm).reshape(F, 3)
    pts0 = pts_gt + s_init.normal(3 * P, spec.init_point_mm).reshape(P, 3)

    # distance constraints between observed points (reference: ArUco marker pairs)
    M = spec.n_constraints
    c_i = np.zeros(M, np.uint32); c_j = np.zeros(M, np.uint32)
    c_dist = np.zeros(M); c_sigma = np.zeros(M)
    if M > 0:
        seen = np.unique(obs_pt)
        s_c = Stream(spec.seed, 6)
        pick = s_c.integers(2 * M, seen.shape[0])
        for m in range(M):
            a, b = int(seen[pick[2 * m]]), int(seen[pick[2 * m + 1]])
            if a == b:
                b = int(seen[(pick[2 * m + 1] + 1) % seen.shape[0]])
            c_i[m], c_j[m] = a, b
            c_dist[m] = float(np.linalg.norm(pts_gt[a] - pts_gt[b]))
            c_sigma[m] = 0.5

    return Scene(
        spec=spec,
        u=np.ascontiguousarray(uv[:, 0]), v=np.ascontiguousarray(uv[:, 1]),
        mcx=np.ascontiguousarray(ml[:, 0]), mcy=np.ascontiguousarray(ml[:, 1]),
        pt=obs_pt.astype(np.uint32), fr=obs_fr.astype(np.uint32),
        cam_gt=cam_gt, views_gt=views_gt.reshape(-1).copy(), pts_gt=pts_gt.reshape(-1).copy(),
        cam0=cam0, views0=views0.reshape(-1).copy(), pts0=pts0.reshape(-1).copy(),
        spx=spx_tot, scale=float(spec.scale), config=spec.config,
        fixed_mask=fixed_mask, lower=lower, upper=upper,
        c_i=c_i, c_j=c_j, c_dist=c_dist, c_sigma=c_sigma,
        use_constraints=0 if spec.recalib else 1,
        n_lenses=int(lenses.shape[0]),
        **img,
    )


def _image_points(pf_p, pf_f, x_ups, y_ups, vdn, scale):
    order = np.lexsort((pf_p, pf_f))
    return dict(img_x=(x_ups[order] + 0.5) / scale - 0.5, img_y=(y_ups[order] + 0.5) / scale - 0.5, img_vd=vdn[order].copy(),
                img_fr=pf_f[order].astype(np.uint32), img_pt=pf_p[order].astype(np.uint32))
