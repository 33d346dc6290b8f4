"""Randomised campaign for constant poses and the frame-windowed driver (lifcal_ba_set_fixed_frames / lifcal_ba_solve_windowed):
random sequences, window lengths and overlaps, against the same schedule driven through the CPU restatement window by window
(tests/test_gpu_windowed.py: oracle_windowed).  gpurun -- tools/gpurun.sh run tools/fuzz_windowed.py [n_cases] [first_seed]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import oracle                                                                    # noqa: E402
from lifcal_amd import _capi as capi, scene, performBundleAdjustmentWindowed    # noqa: E402
from tests.helpers import S                                                      # noqa: E402
from tests.test_gpu_windowed import oracle_windowed                              # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 15000
bad = 0; t_start = time.time()
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    F = int(rng.integers(30, 110)); P = int(rng.integers(100, 500)); vis = int(rng.integers(3, 11))
    win = int(rng.integers(max(12, vis + 4), 41)); ov = int(rng.integers(max(2, vis // 2), max(3, win // 2)))
    cfg = int(rng.integers(0, 3)) | (int(rng.integers(0, 2)) << 2) | 0x500 | (0x200 if rng.random() < 0.6 else 0) | (0x800 if rng.random() < 0.5 else 0)
    fixed_intrinsics = rng.random() < 0.7
    tag = f"S({F}, {P}, {vis}, {cfg:#x}, {seed0 + case}) window {win} overlap {ov} intrinsics {'constant' if fixed_intrinsics else 'free'}"
    try:
        sc = scene.make_scene(S(F, P, vis, cfg, seed0 + case, outlier_fraction=0.02 if cfg & 0x200 else 0.0))
        live = 5 + (sc.config & 3) + (2 if sc.config & 4 else 0)
        mk = lambda: capi.ProblemArrays(sc.u, sc.v, sc.mcx, sc.mcy, sc.pt, sc.fr, (sc.cam_gt if fixed_intrinsics else sc.cam0).copy(), sc.views0.copy(), sc.pts0.copy(),
                                        sc.spx, sc.scale, sc.config, fixed_mask=((1 << live) - 1) if fixed_intrinsics else 0, use_constraints=0)
        pg = mk(); reps = performBundleAdjustmentWindowed(pg, win, ov)
        po = mk(); oreps = oracle_windowed(sc, po, win, ov)
        same_sched = [(r.first_frame, r.n_frames, r.n_points, r.n_obs) for r in reps] == [o[:4] for o in oreps]
        same_traj = [(r.summary.iterations, r.summary.termination) for r in reps] == [o[4:] for o in oreps]
        dv = np.abs(pg.views - po.views).max() / (1 + np.abs(po.views).max()); dp = np.abs(pg.pts - po.pts).max() / (1 + np.abs(po.pts).max())
        dcam = np.abs(pg.cam[:live] - po.cam[:live]).max() / np.abs(po.cam[:live]).max()
        if not (same_sched and same_traj and dv < 1e-6 and dp < 1e-6 and dcam < 1e-6):
            bad += 1
            print(f"FAIL {tag}: schedule {same_sched} trajectories {same_traj} "
                  f"{[(r.summary.iterations, r.summary.termination) for r in reps]} vs {[o[4:] for o in oreps]} dviews {dv:.1e} dpts {dp:.1e} dcam {dcam:.1e}", flush=True)
        elif case % 5 == 0:
            print(f"ok   {tag}: {len(reps)} windows, obs {sc.n_obs} ({time.time() - t_start:.0f} s)", flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1; print(f"ERROR {tag}: {e!r}", flush=True)
    finally:
        oracle.set_fixed_frames(None)
print(f"{n_cases} windowed cases, {bad} failures, {time.time() - t_start:.0f} s")
