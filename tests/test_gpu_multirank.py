"""Two (or three) ranks on ONE GPU: real inter-process exchange through the collective hooks (torch.distributed,
gloo), so sharding, the iteration-0 Jacobi all-reduce, the slab all-gather of the reduced block, the LM loop and the
final point gather run as they do on N GPUs (only RCCL itself is replaced).  Rank 0 compares with the single-process
oracle."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir, spec_kw, mode):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lifcal_amd import BundleAdjustment, _capi as capi, scene
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    if "stress_case" in spec_kw:          # a deformed scene of tests/test_gpu_stress.py
        from tests.test_gpu_stress import deformed_problem
        _, mk, _ = deformed_problem(spec_kw["stress_case"])
        pa = mk()
    else:
        spec_kw = dict(spec_kw); bounded = spec_kw.pop("bounded", False)
        window_args = {k: spec_kw.pop(k) for k in ("_window", "_overlap") if k in spec_kw}
        sc = scene.make_scene(scene.SceneSpec(**spec_kw))
        if bounded:
            from tests.helpers import bounded_problem
            pa = bounded_problem(sc)
        else:
            pa = capi.ProblemArrays.from_scene(sc)
    o = capi.default_options_py(); o.rank = rank; o.world_size = world
    if mode == "windowed":
        spec_kw = dict(spec_kw, **window_args)
        live = 5 + (sc.config & 3) + (2 if sc.config & 4 else 0)   # BASELINE configs[4]: intrinsics constant, poses + points refined
        pa = capi.ProblemArrays(sc.u, sc.v, sc.mcx, sc.mcy, sc.pt, sc.fr, sc.cam_gt.copy(), sc.views0.copy(), sc.pts0.copy(), sc.spx, sc.scale, sc.config,
                                fixed_mask=(1 << live) - 1, use_constraints=0)
    full = pa
    if mode == "shard":   # the rank receives ONLY the observations of its points (lifcal_ba_partition_points + lifcal_ba_create_shard)
        pa.struct.use_constraints = 0
        part = capi.PartitionArrays(pa, world)
        pa = part.shard_of(full, rank)
        ba = BundleAdjustment(pa, o, partition=part)
    else:
        ba = BundleAdjustment(pa, o)

    def hook(ptr, count, stream):
        hip.hipStreamSynchronize(stream)
        buf = np.empty(count)
        hip.hipMemcpy(buf.ctypes.data, ptr, count * 8, 2)
        t = torch.from_numpy(buf)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        hip.hipMemcpy(ptr, buf.ctypes.data, count * 8, 1)
        return 0
    ba.set_allreduce(hook)
    calls = {"gather": 0, "gather_doubles": 0}

    def ghook(send, recv, count, stream):
        hip.hipStreamSynchronize(stream)
        buf = np.empty(count)
        hip.hipMemcpy(buf.ctypes.data, send, count * 8, 2)
        parts = [torch.empty(count, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(buf))
        allb = torch.cat(parts).numpy()
        hip.hipMemcpy(recv, allb.ctypes.data, world * count * 8, 1)
        calls["gather"] += 1; calls["gather_doubles"] = count
        return 0
    if mode in ("allgather", "windowed", "shard"):
        ba.set_allgather(ghook)
    if mode == "windowed":   # lifcal_ba_solve_windowed with this handle's collectives: every window sharded over the ranks
        from lifcal_amd import performBundleAdjustmentWindowed
        reps = performBundleAdjustmentWindowed(pa, spec_kw["_window"], spec_kw["_overlap"], options=o, comm_template=ba)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), cam=pa.cam, views=pa.views, pts=pa.pts,
                 it=np.array([r.summary.iterations for r in reps]), term=np.array([r.summary.termination for r in reps]),
                 first=np.array([r.first_frame for r in reps]), gather_calls=calls["gather"])
        ba.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    sw = ba.sweep(1e4, want_matrices=(rank == 0))
    summ = ba.performBundleAdjustment()
    st = ba.calcReprojectionError()
    info = ba.info()
    xp, yp = ba.projectObservations()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), cam=pa.cam, views=pa.views, pts=pa.pts, cost0=sw.cost, xp=xp, yp=yp,
             S=sw.S if rank == 0 else np.zeros(1), rhs=sw.rhs if rank == 0 else np.zeros(1),
             gather_calls=calls["gather"], gather_doubles=calls["gather_doubles"], n_red=info.n_reduced,
             it=summ.iterations, term=summ.termination, final=summ.final_cost, n_local=info.n_obs_local,
             steps=np.array([summ.successful_steps, summ.unsuccessful_steps]),
             stats=np.array([st.std_x, st.std_y, st.num_points, st.num_inliers, st.mae_x, st.mae_y]))
    ba.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("spec_kw,mode,world", [
    (dict(n_frames=24, n_points=160, window=6, config=0xF06, seed=1401, outlier_fraction=0.02), "allreduce", 2),
    (dict(n_frames=8, n_points=60, window=None, config=0x506, seed=1402, n_constraints=3), "allreduce", 2),
    (dict(n_frames=24, n_points=160, window=6, config=0xF06, seed=1401, outlier_fraction=0.02), "allgather", 2),
    (dict(n_frames=60, n_points=400, window=8, config=0xF06, seed=1403, outlier_fraction=0.02), "allgather", 3),
    (dict(n_frames=8, n_points=60, window=None, config=0x506, seed=1402, n_constraints=3), "allgather", 2),
], ids=["windowed_robust", "constraints", "windowed_robust_slabs", "three_ranks_slabs", "constraints_fall_back_to_allreduce"])
def test_ranks_match_the_single_process_oracle(built, tmp_path, spec_kw, mode, world):
    import oracle
    from lifcal_amd import _capi as capi, scene
    from tests.helpers import free_port, scaled_max_err, vec_err
    mp.spawn(_worker, args=(world, free_port(), str(tmp_path), spec_kw, mode), nprocs=world, join=True)
    r0 = np.load(os.path.join(str(tmp_path), "rank0.npz")); r1 = np.load(os.path.join(str(tmp_path), f"rank{world - 1}.npz"))
    if mode == "allgather" and not spec_kw.get("n_constraints"):
        # the reduced block went through the slab exchange, and a slab is well below the whole block
        n_red = int(r0["n_red"])
        assert int(r0["gather_calls"]) > 0 and int(r0["gather_doubles"]) < 0.8 * (n_red * (n_red + 1) // 2)
    else:
        assert int(r0["gather_calls"]) == 0
    sc = scene.make_scene(scene.SceneSpec(**spec_kw))
    ref = oracle.sweep(capi.ProblemArrays.from_scene(sc), radius=1e4, threads=4)
    assert abs(float(r0["cost0"]) - ref.cost) <= 1e-12 * ref.cost and abs(float(r1["cost0"]) - ref.cost) <= 1e-12 * ref.cost
    assert scaled_max_err(r0["S"], ref.S) < 1e-9 and vec_err(r0["rhs"], ref.rhs) < 1e-9
    n_local = [int(np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))["n_local"]) for r in range(world)]
    assert sum(n_local) == sc.n_obs and min(n_local) > 0.6 * sc.n_obs / world
    pb = capi.ProblemArrays.from_scene(sc)
    so = oracle.solve(pb, threads=4)
    for r in (r0, r1):   # every rank ends with the full, identical result
        assert (int(r["it"]), int(r["term"])) == (so.iterations, so.termination)
        assert abs(float(r["final"]) - so.final_cost) <= 1e-8 * so.final_cost
        assert np.allclose(r["cam"][:5], pb.cam[:5], rtol=1e-6)
        assert np.allclose(r["pts"], pb.pts, rtol=0, atol=1e-6 * (1 + np.abs(pb.pts).max()))
        assert np.allclose(r["views"], pb.views, rtol=0, atol=1e-6 * (1 + np.abs(pb.views).max()))
    assert np.array_equal(r0["pts"], r1["pts"]) and np.array_equal(r0["cam"], r1["cam"])
    stt, err = oracle.reproj_stats(capi.ProblemArrays(sc.u, sc.v, sc.mcx, sc.mcy, sc.pt, sc.fr, r0["cam"], r0["views"], r0["pts"], sc.spx, sc.scale, sc.config), want_errors=True)
    assert abs(float(r0["stats"][0]) - stt.std_x) < 1e-9 and int(r0["stats"][2]) == sc.n_obs
    # the maxima (reference CameraCalibration.cpp:1083-1084) are over ALL observations, on every rank
    for r in (r0, r1):
        assert abs(float(r["stats"][4]) - stt.mae_x) < 1e-9 and abs(float(r["stats"][5]) - stt.mae_y) < 1e-9
    # lifcal_ba_project_observations: a rank fills the observations of the points it owns (NaN elsewhere); together the ranks
    # cover every observation exactly once, with the projection of the final parameters
    xs = np.stack([np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))["xp"] for r in range(world)])
    filled = np.isfinite(xs)
    assert np.all(filled.sum(0) == 1) and [int(f.sum()) for f in filled] == n_local
    assert np.max(np.abs(np.nansum(xs, 0) - (sc.u + err[:, 0]))) < 1e-8


@pytest.mark.parametrize("k,world", [(10, 2), (11, 2), (22, 3), (26, 4), (37, 2), (41, 3)])   # 10: recalib (bounds + fixed mask), 131 frames, one 364-observation group
def test_slab_exchange_on_deformed_scenes(built, tmp_path, k, world):
    """ragged / tiny / wide-window scenes: ranks with few or no frames of their own, every rank must end with the
    single-process reduced system and solve"""
    import oracle
    from tests.helpers import free_port, scaled_max_err, vec_err
    from tests.test_gpu_stress import deformed_problem
    spec, mk, n = deformed_problem(k)
    mp.spawn(_worker, args=(world, free_port(), str(tmp_path), {"stress_case": k}, "allgather"), nprocs=world, join=True)
    ref = oracle.sweep(mk(), radius=1e4, threads=4)
    so = oracle.solve(mk(), threads=4)
    r0 = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    assert abs(float(r0["cost0"]) - ref.cost) <= 1e-11 * ref.cost
    assert scaled_max_err(r0["S"], ref.S) < 1e-8 and vec_err(r0["rhs"], ref.rhs) < 1e-8
    assert sum(int(np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))["n_local"]) for r in range(world)) == n
    for r in range(world):
        rr = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        assert (int(rr["it"]), int(rr["term"])) == (so.iterations, so.termination)
        assert abs(float(rr["final"]) - so.final_cost) <= 1e-7 * so.final_cost
    # reprojection statistics of the common result: sums AND maxima are those of all observations on every rank
    pf = mk(); pf.cam[:] = r0["cam"]; pf.views[:] = r0["views"]; pf.pts[:] = r0["pts"]
    stt = oracle.reproj_stats(pf)
    for r in range(world):
        rr = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        assert abs(float(rr["stats"][0]) - stt.std_x) < 1e-9 and int(rr["stats"][2]) == n
        assert abs(float(rr["stats"][4]) - stt.mae_x) < 1e-9 and abs(float(rr["stats"][5]) - stt.mae_y) < 1e-9


@pytest.mark.parametrize("spec_kw,world", [
    (dict(n_frames=8, n_points=60, window=None, config=0xF06, seed=1320, outlier_fraction=0.02, bounded=True), 2),   # 12 backtracks, leaves through the minimum-step test
    (dict(n_frames=16, n_points=100, window=5, config=0xF06, seed=1322, outlier_fraction=0.02, bounded=True), 3),    # several searches with one backtrack each
], ids=["min_step_exit", "windowed_three_ranks"])
def test_bounded_problems_backtrack_in_lockstep(built, tmp_path, spec_kw, world):
    """BASELINE configs[4]'s code path (recalib: fixed mask + box bounds) at world size > 1 with a box tight enough that the
    Armijo line search runs: every trial is a fused sweep with two all-reduces, so all ranks must take every branch of the
    search together (the largest step component, the interpolated step lengths and the exit test are global quantities)."""
    import oracle
    from lifcal_amd import scene
    from tests.helpers import bounded_problem, free_port
    mp.spawn(_worker, args=(world, free_port(), str(tmp_path), spec_kw, "allgather"), nprocs=world, join=True)
    kw = dict(spec_kw); kw.pop("bounded")
    pb = bounded_problem(scene.make_scene(scene.SceneSpec(**kw)))
    so = oracle.solve(pb, threads=4)
    rs = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    for rr in rs:
        assert (int(rr["it"]), int(rr["term"])) == (so.iterations, so.termination)
        assert (int(rr["steps"][0]), int(rr["steps"][1])) == (so.successful_steps, so.unsuccessful_steps)
        assert abs(float(rr["final"]) - so.final_cost) <= 1e-8 * so.final_cost
        assert np.allclose(rr["cam"][:5], pb.cam[:5], rtol=1e-6)
        assert np.array_equal(rr["cam"], rs[0]["cam"]) and np.array_equal(rr["views"], rs[0]["views"]) and np.array_equal(rr["pts"], rs[0]["pts"])


def test_windowed_driver_on_two_ranks(built, tmp_path):
    """lifcal_ba_solve_windowed at world size 2 (the shape of BASELINE configs[4]: frame windows, each sharded by 3D point over the
    ranks, collectives of the caller's handle): both ranks end with the single-process result"""
    from lifcal_amd import _capi as capi, scene, performBundleAdjustmentWindowed
    from tests.helpers import free_port
    spec_kw = dict(n_frames=48, n_points=300, window=8, config=0xF06, seed=1450, outlier_fraction=0.02, _window=20, _overlap=6)
    mp.spawn(_worker, args=(2, free_port(), str(tmp_path), spec_kw, "windowed"), nprocs=2, join=True)
    kw = {k: v for k, v in spec_kw.items() if not k.startswith("_")}
    sc = scene.make_scene(scene.SceneSpec(**kw))
    live = 5 + (sc.config & 3) + (2 if sc.config & 4 else 0)
    pa = capi.ProblemArrays(sc.u, sc.v, sc.mcx, sc.mcy, sc.pt, sc.fr, sc.cam_gt.copy(), sc.views0.copy(), sc.pts0.copy(), sc.spx, sc.scale, sc.config,
                            fixed_mask=(1 << live) - 1, use_constraints=0)
    reps = performBundleAdjustmentWindowed(pa, 20, 6)
    r0 = np.load(os.path.join(str(tmp_path), "rank0.npz")); r1 = np.load(os.path.join(str(tmp_path), "rank1.npz"))
    assert int(r0["gather_calls"]) > 0
    for r in (r0, r1):
        assert list(r["first"]) == [w.first_frame for w in reps] and list(r["it"]) == [w.summary.iterations for w in reps]
        assert list(r["term"]) == [w.summary.termination for w in reps]
        assert np.allclose(r["views"], pa.views, rtol=0, atol=1e-6 * (1 + np.abs(pa.views).max()))
        assert np.allclose(r["pts"], pa.pts, rtol=0, atol=1e-6 * (1 + np.abs(pa.pts).max()))
    assert np.array_equal(r0["views"], r1["views"]) and np.array_equal(r0["pts"], r1["pts"])


def test_shard_local_problems_give_the_whole_problem_result(built, tmp_path):
    """lifcal_ba_create_shard: every rank is handed only the observations of the points it owns (partition computed once from the
    index arrays); sweep, solve and statistics are those of the whole-problem path and of the single-process oracle"""
    import oracle
    from lifcal_amd import _capi as capi, scene
    from tests.helpers import free_port, scaled_max_err, vec_err
    spec_kw = dict(n_frames=40, n_points=300, window=8, config=0xF06, seed=1460, outlier_fraction=0.02)
    mp.spawn(_worker, args=(3, free_port(), str(tmp_path), spec_kw, "shard"), nprocs=3, join=True)
    sc = scene.make_scene(scene.SceneSpec(**spec_kw))
    ref = oracle.sweep(capi.ProblemArrays.from_scene(sc), radius=1e4, threads=4)
    pb = capi.ProblemArrays.from_scene(sc)
    so = oracle.solve(pb, threads=4)
    rs = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(3)]
    assert sum(int(r["n_local"]) for r in rs) == sc.n_obs
    assert abs(float(rs[0]["cost0"]) - ref.cost) <= 1e-12 * ref.cost
    assert scaled_max_err(rs[0]["S"], ref.S) < 1e-9 and vec_err(rs[0]["rhs"], ref.rhs) < 1e-9
    for r in rs:
        assert int(r["gather_calls"]) > 0
        assert (int(r["it"]), int(r["term"])) == (so.iterations, so.termination)
        assert abs(float(r["final"]) - so.final_cost) <= 1e-8 * so.final_cost
        assert np.allclose(r["cam"][:5], pb.cam[:5], rtol=1e-6)
        assert np.allclose(r["pts"], pb.pts, rtol=0, atol=1e-6 * (1 + np.abs(pb.pts).max()))
        assert np.array_equal(r["pts"], rs[0]["pts"]) and np.array_equal(r["views"], rs[0]["views"])
    stt = oracle.reproj_stats(capi.ProblemArrays(sc.u, sc.v, sc.mcx, sc.mcy, sc.pt, sc.fr, rs[0]["cam"], rs[0]["views"], rs[0]["pts"], sc.spx, sc.scale, sc.config))
    for r in rs:
        assert abs(float(r["stats"][0]) - stt.std_x) < 1e-9 and int(r["stats"][2]) == sc.n_obs and abs(float(r["stats"][4]) - stt.mae_x) < 1e-9


def test_block_reduction_is_replicated_bitwise_across_ranks(built, tmp_path, monkeypatch):
    """bandchol3.hpp under world size 3: every rank factors the same all-reduced system with the block odd-even reduction (forced:
    the scene has 14 super-blocks) — no atomics, one writer per block, fixed summation orders — so the replicated step, and with
    it every host decision, must be IDENTICAL on all ranks; the trajectory is the single-process oracle's."""
    import oracle
    from lifcal_amd import scene
    from tests.helpers import free_port
    monkeypatch.setenv("LIFCAL_CR", "1")
    spec_kw = dict(n_frames=40, n_points=260, window=4, config=0xF06, seed=1460, outlier_fraction=0.02)
    mp.spawn(_worker, args=(3, free_port(), str(tmp_path), spec_kw, "allgather"), nprocs=3, join=True)
    pb = capi_problem(scene.make_scene(scene.SceneSpec(**spec_kw)))
    so = oracle.solve(pb, threads=4)
    rs = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(3)]
    for rr in rs:
        assert (int(rr["it"]), int(rr["term"])) == (so.iterations, so.termination)
        assert abs(float(rr["final"]) - so.final_cost) <= 1e-8 * so.final_cost
        assert np.array_equal(rr["cam"], rs[0]["cam"]) and np.array_equal(rr["views"], rs[0]["views"])
    assert np.allclose(rs[0]["cam"][:5], pb.cam[:5], rtol=1e-6)


def capi_problem(sc):
    from lifcal_amd import _capi as capi
    return capi.ProblemArrays.from_scene(sc)
