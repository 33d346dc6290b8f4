#!/usr/bin/env python3
"""bench.py — micro-image observations/second through one Jacobian+Schur sweep (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one lifcal_ba sweep (tables + residual/Jacobian/robust-weight accumulation + point-block
elimination + reduced system) over observations already resident in HBM.
N = 1: the 1.0 M-observation metric point (`metric_web`: 334 frames, 24 720 points, 999 994 observations, window 10,
config 0xF06, fp64).
N > 1: one process per GPU.  Called as `python bench.py --gpus N` from a bare shell (no WORLD_SIZE) this process only
starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child, relays rank 0's JSON line and exits
with the child's code; launched by torch.distributed.run itself (the driver's N > 1 command line) it is a rank.  Weak scaling
(default): rank r owns the r-th copy of the metric scene along the trajectory (334 N frames, 24 720 N points, sharded by 3D
point by the library); `--scaling strong --workload cfg4` shards ONE problem (BASELINE configs[3]).  The only data-path
exchange is the reduced normal equations per sweep, on the library's own RCCL communicator; if that cannot be set up the run
FAILS unless --allow-comm-fallback is given (a scaling number is never silently the fallback's).
Timing: `--preheat-ms` (default 150) of untimed sweeps bring the GPU to its sustained clocks (the same kernel is 5 % slower in the
first milliseconds after idle), then W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize pairs; every 8th
timed step carries the dominant kernel's own start / stop events (a stamped launch costs ~5 us of queue time).
Rank 0 prints ONE JSON line.  `value` is whole-job obs/s; `roofline` prices the dominant kernel
against the 8 TB/s HBM peak with the algorithmic bytes of DESIGN.md; `cpu_baseline` times the CPU
restatement (oracle/, kind "port") on a bounded sample of the same workload on this host's cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md
FP64_PEAK = 78.6e12  # flop/s, fp64 vector peak (256 CU x 4 SIMD x 16 lanes x 2 x 2.4 GHz)
# fp64 flops the sweep EXECUTES per observation at the metric point (config 0xF06), counted in the gfx950 ISA / kernel
# source: observation loop ~600 + block emission ~155 + Schur product ~570 + factor/Z ~35
FP64_FLOP_PER_OBS = 1360.0


def tiled_problem(sc, copies):
    """`copies` disjoint copies of the scene along the trajectory sharing one camera block."""
    from lifcal_amd import _capi as capi
    F, P = sc.spec.n_frames, sc.spec.n_points
    cat = lambda a: np.concatenate([a] * copies)
    pt = np.concatenate([sc.pt.astype(np.uint32) + np.uint32(c * P) for c in range(copies)])
    fr = np.concatenate([sc.fr.astype(np.uint32) + np.uint32(c * F) for c in range(copies)])
    return capi.ProblemArrays(cat(sc.u), cat(sc.v), cat(sc.mcx), cat(sc.mcy), pt, fr, sc.cam0, cat(sc.views0), cat(sc.pts0),
                              sc.spx, sc.scale, sc.config, use_constraints=0)


def algorithmic_bytes(n_obs, n_points, n_frames, n_red, b_obs=40):
    """B_alg = N*b_obs + 2*(24 P + 48 F + 8*17) + 8*n_red(n_red+1)/2 + 8*n_red; b_obs = 40 (fp64) or 24 (precision = 1)."""
    return n_obs * b_obs + 2 * (24 * n_points + 48 * n_frames + 8 * 17) + 8 * (n_red * (n_red + 1) // 2) + 8 * n_red


def accumulate_kernel_bytes(n_obs, n_points, n_frames, b_obs=40):
    """share of B_alg the dominant kernel (k_sweep) must move: the observation stream + one parameter read."""
    return n_obs * b_obs + (24 * n_points + 48 * n_frames + 8 * 17)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sc, pa, budget_s=25.0):
    """The CPU restatement (oracle/, kind "port": real Ceres is not in the image) on THIS host's cores, on the identical flattened
    input the GPU sweep was timed on: the dual-number arm (what ceres::AutoDiffCostFunction evaluates on the
    reference's functor — `value`) and the analytic-Jacobian arm with per-lens / per-frame tables (oracle/analytic.hpp), both
    followed by the same dense Schur elimination, all host threads (reference CameraCalibration.cpp:961).  Secondary: a full solve
    of BASELINE configs[2] (cfg3) with the dual-number arm."""
    import oracle
    from lifcal_amd import _capi as capi, scene
    threads = max(1, min(oracle.hardware_threads(), len(os.sched_getaffinity(0))))

    def arm(analytic, budget):
        oracle.sweep(pa, radius=1e4, threads=threads, want_matrices=False, analytic=analytic)   # warm-up (page faults, thread start)
        best_jac, best_schur, best_total, reps, t_end = None, None, None, 0, time.time() + budget
        while reps < 2 or (time.time() < t_end and reps < 10):
            r = oracle.sweep(pa, radius=1e4, threads=threads, want_matrices=False, analytic=analytic)
            best_jac = r.seconds_eval if best_jac is None else min(best_jac, r.seconds_eval)
            best_schur = r.seconds_schur if best_schur is None else min(best_schur, r.seconds_schur)
            best_total = r.seconds if best_total is None else min(best_total, r.seconds)
            reps += 1
        return {"seconds": best_total, "seconds_jacobian": best_jac, "seconds_schur": best_schur, "best_of": reps}
    dual = arm(False, 0.5 * budget_s)
    ana = arm(True, 0.3 * budget_s)
    ref = oracle.sweep(pa, radius=1e4, threads=threads, want_matrices=True, analytic=False)   # the cross-check operand (not timed)
    # the two arms differ in the Jacobian evaluation only; the dense Schur elimination behind it is the same code (and, on 256
    # threads, a noisy one): each arm is priced with its own best Jacobian time + the best elimination time seen in either arm
    schur = min(dual["seconds_schur"], ana["seconds_schur"])
    for a in (dual, ana):
        a["value"] = sc.n_obs / (a["seconds_jacobian"] + schur); a["unit"] = "obs/s"
        a["value_own_best_total"] = sc.n_obs / a["seconds"]   # the arm's own best end-to-end sweep (no cross-arm composition)
    out = {"value": dual["value"], "value_is": "composite: the arm's best Jacobian time + the best dense-elimination time seen in either arm (favours the CPU)",
           "unit": "obs/s", "cores": threads, "cpu_model": cpu_model(), "kind": "port",
           "sample": f"the whole bench workload, identical flattened input: {sc.spec.n_frames} frames, {sc.spec.n_points} points, {sc.n_obs} obs, "
                     f"config {sc.config:#x}; one Jacobian + dense-Schur sweep, dual-number (autodiff-equivalent) Jacobian, best of {dual['best_of']} "
                     f"(best Jacobian time + best elimination time)",
           "arms": {"dual_number": dual, "analytic": ana}}
    try:   # secondary: the same CPU restatement through a full solve of a smaller BASELINE config (works on a private copy)
        sc3 = scene.make_scene(scene.baseline_spec("cfg3"))
        pa3 = capi.ProblemArrays.from_scene(sc3)
        t0 = time.perf_counter()
        summ = oracle.solve(pa3, threads=threads)
        t_solve = time.perf_counter() - t0
        st = oracle.reproj_stats(pa3, 1.0)
        out["solve_cfg3"] = {"observations": sc3.n_obs, "seconds": t_solve, "iterations": int(summ.iterations), "final_cost": float(summ.final_cost),
                             "final_rms_reproj_px": [float(st.std_x), float(st.std_y)]}
    except Exception as e:  # noqa: BLE001
        out["solve_cfg3"] = {"error": repr(e)}
    return out, ref


def load_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/traffic.json: counters cannot
    be collected inside this process), and where they came from; (None, None) if absent."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        e = t.get(workload, {})
        return e.get("k_sweep_hbm_bytes"), "profiles/traffic.json <- " + str(e.get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes"))
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)    # (26 ms of sweeps at N = 1: the launch ramp and the final synchronisation, ~40 us once, stay below 0.2 % of a step)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--preheat-ms", type=float, default=150.0,
                    help="untimed sweeps for this long BEFORE the W warm-up steps: the GPU's clocks ramp over the first ~100 ms of load (the same kernel "
                         "measures 0.119 ms in a 6 ms run from idle and 0.113 ms sustained), and the metric is sustained throughput; 0 disables it")
    ap.add_argument("--workload", default="metric_web",
                    help="metric_web (default): the 1.0 M-observation point with the lenses of every image point chosen by the reference's own generator, "
                         "projectPointsToRawImage, through its GPU port lifcal_mla_project; metric: the round-1 scene (K-nearest lens stand-in); cfg1..cfg5")
    ap.add_argument("--lens-selection", choices=["auto", "web", "nearest"], default="auto", help="auto: web for *_web workloads, nearest otherwise")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = rank r adds the r-th copy of the workload along the trajectory (default, per-GPU work fixed); "
                         "strong = ONE problem (e.g. --workload cfg4: BASELINE configs[3], 1000 frames / 50 k points) sharded by 3D point over the N ranks")
    ap.add_argument("--precision", type=int, choices=[0, 1], default=0,
                    help="0: fp64 everywhere (the BASELINE metric configuration); 1: options.precision = 1, residual / Jacobian of an observation in fp32 "
                         "(fp32 observation words + fp32 lens table), all accumulation and the solve in fp64 (BASELINE configs[4]'s arithmetic)")
    ap.add_argument("--deterministic", action="store_true", help="options.deterministic = 1: ordered reductions, bitwise reproducible (slower)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-solve", action="store_true", help="skip the full LM solve after the timed sweeps (profiling runs: only sweep kernels in the trace)")
    ap.add_argument("--comm", choices=["rccl", "gloo"], default="rccl",
                    help="rccl: library-owned RCCL communicator (default, one GPU per rank); gloo: rehearsal of the multi-process path on ONE GPU (all ranks on cuda:0, all-reduce through host memory)")
    ap.add_argument("--allow-comm-fallback", action="store_true",
                    help="N > 1 with --comm rccl: if the library's RCCL communicator cannot be set up, continue on torch.distributed's group with "
                         "host-synchronising copy hooks (config.comm says so) instead of failing")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher.  Nothing in this process has touched HIP or torch yet; the ranks run in
        # a child (never os.exec*), its stdout (rank 0's JSON line) and return code are relayed.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        if lines:
            print(lines[-1])
        else:
            sys.stdout.write(res.stdout)
        raise SystemExit(res.returncode if res.returncode != 0 or lines else 1)

    import torch
    from lifcal_amd import BundleAdjustment, _capi as capi, scene, comm_unique_id

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world   # (launched by torch.distributed.run with another --nproc-per-node: the process group decides)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the bundle-adjustment path has no CPU fallback")
    if args.comm == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.comm == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    spec = scene.baseline_spec(args.workload)
    use_web = args.lens_selection == "web" or (args.lens_selection == "auto" and args.workload.endswith("_web"))
    selector = None
    if use_web:
        # which micro lenses see an image point: the reference's walk over the nearest lens and the epipolar web
        # (src/CameraCalibration.cpp:661-752), run by this library's port of projectPointsToRawImage on the GPU
        from lifcal_amd.mla import MicroLensGrid
        grid = MicroLensGrid(spec.raw_width, spec.raw_height, spec.lens_diameter, spec.lens_base_y, spec.grid_rotation, spec.grid_offset, True, device=local_rank)

        def selector(img_x, img_y, img_vd, img_fr, img_pt, scale):
            o = grid.projectPointsToRawImage(img_x, img_y, img_vd, int(scale), fr=img_fr, pt=img_pt)
            return o.src, o.mcx, o.mcy
    sc = scene.make_scene(spec, lens_selector=selector)
    if use_web:
        grid.close()
    strong = args.scaling == "strong" and world > 1
    pa = tiled_problem(sc, world) if (world > 1 and not strong) else capi.ProblemArrays.from_scene(sc)
    n_obs_total = int(pa.struct.n_obs)
    o = capi.default_options_py()
    o.device = local_rank
    o.rank = rank
    o.world_size = world
    o.precision = args.precision
    o.deterministic = 1 if args.deterministic else 0
    t_create = time.perf_counter()
    ba = BundleAdjustment(pa, o)
    t_create = time.perf_counter() - t_create   # planner (host) + upload: once per problem, outside the metric
    rccl_ok = True
    comm_used = args.comm if world > 1 else None
    if world > 1 and args.comm == "rccl":
        # the library's own RCCL communicator (ncclAllReduce / ncclAllGather on the solver's stream).  If it cannot be set up on
        # this node, every rank falls back together to torch.distributed's RCCL group on device buffers (same transport, one
        # staging copy per call) rather than losing the measurement.
        try:
            uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                uid = torch.frombuffer(bytearray(comm_unique_id()), dtype=torch.uint8).cuda()
            dist.broadcast(uid, src=0)
            ba.comm_init_rccl(bytes(uid.cpu().numpy().tobytes()))
        except Exception as e:  # noqa: BLE001
            print(f"[bench rank {rank}] library RCCL communicator failed ({e!r}); falling back to the torch.distributed group", file=sys.stderr)
            rccl_ok = False
        flag = torch.tensor([1 if rccl_ok else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        rccl_ok = bool(flag.item())
        if not rccl_ok and not args.allow_comm_fallback:
            ba.close()
            dist.destroy_process_group()
            raise SystemExit("bench.py: the library's RCCL communicator could not be set up on every rank (pass --allow-comm-fallback to measure the torch.distributed fallback instead)")
        if not rccl_ok:
            comm_used = "torch-rccl (fallback)"
            import ctypes as C
            hip = C.CDLL("libamdhip64.so")
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            hip.hipStreamSynchronize.argtypes = [C.c_void_p]

            def dev_hook(ptr, count, stream):
                hip.hipStreamSynchronize(stream)
                t = torch.empty(count, dtype=torch.float64, device="cuda")
                hip.hipMemcpy(t.data_ptr(), ptr, count * 8, 3)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                torch.cuda.synchronize()
                hip.hipMemcpy(ptr, t.data_ptr(), count * 8, 3)
                return 0
            ba.set_allreduce(dev_hook)

            def dev_ghook(send, recv, count, stream):
                hip.hipStreamSynchronize(stream)
                t = torch.empty(count, dtype=torch.float64, device="cuda")
                hip.hipMemcpy(t.data_ptr(), send, count * 8, 3)
                allb = torch.empty(world * count, dtype=torch.float64, device="cuda")
                dist.all_gather_into_tensor(allb, t)
                torch.cuda.synchronize()
                hip.hipMemcpy(recv, allb.data_ptr(), world * count * 8, 3)
                return 0
            ba.set_allgather(dev_ghook)
    elif world > 1:
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]

        def hook(ptr, count, stream):
            hip.hipStreamSynchronize(stream)
            buf = np.empty(count)
            hip.hipMemcpy(buf.ctypes.data, ptr, count * 8, 2)
            t = torch.from_numpy(buf)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            hip.hipMemcpy(ptr, buf.ctypes.data, count * 8, 1)
            return 0
        ba.set_allreduce(hook)

        def ghook(send, recv, count, stream):
            hip.hipStreamSynchronize(stream)
            buf = np.empty(count)
            hip.hipMemcpy(buf.ctypes.data, send, count * 8, 2)
            parts = [torch.empty(count, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(parts, torch.from_numpy(buf))
            allb = torch.cat(parts).numpy()
            hip.hipMemcpy(recv, allb.ctypes.data, world * count * 8, 1)
            return 0
        ba.set_allgather(ghook)
    info = ba.info()
    radius = 1e4

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    first = ba.sweep(radius)                       # also fixes the Jacobi scaling (iteration-0 semantics)
    if args.preheat_ms > 0:   # clocks up (see --preheat-ms); the same work as the timed steps, untimed
        def heat_batch():
            for _ in range(50):
                ba.sweep_enqueue(radius)
            ba.sweep(radius)
        t_heat = time.perf_counter()
        heat_batch()
        batch_ms = max(1e-3, (time.perf_counter() - t_heat) * 1e3)
        rounds = max(0, int(args.preheat_ms / batch_ms + 0.999) - 1)
        if dist is not None:   # every sweep holds a collective: the ranks must run the SAME number of them
            t = torch.tensor([rounds], dtype=torch.int64, device="cpu" if args.comm == "gloo" else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            rounds = int(t.item())
        for _ in range(min(rounds, 200)):
            heat_batch()
    for _ in range(args.warmup):
        ba.sweep_enqueue(radius)
    ba.sweep(radius)
    # every 8th timed sweep carries the dominant kernel's own start / stop events (HIP events written by the dispatch packet, on the
    # library's stream): a stamped launch costs ~5 us of queue time, so stamping all K of them would add 4 % to the very steps being timed
    prof_stride = 8 if args.steps >= 16 else 1
    ba.profile_begin(args.steps, prof_stride)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ba.sweep_enqueue(radius)
    prof = ba.profile_end()                        # synchronises the library's stream
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.comm == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    last = ba.sweep(radius, want_matrices=(world == 1 and not args.no_cpu_baseline))   # (matrices: the in-run cross-check against the CPU restatement below)
    assert abs(last.cost - first.cost) <= 1e-9 * abs(first.cost), "sweep is not idempotent"

    # second half of the BASELINE metric: final RMS reprojection error after the full LM solve (outside the timed region)
    # (N = 1 only, like the CPU baseline: the weak-scaled N > 1 problem is a different scene; never let it cost the bench line)
    solve = None
    if world == 1 and not args.no_solve:
        try:
            before = ba.calcReprojectionError(1.0)
            torch.cuda.synchronize()
            t_solve = time.perf_counter()
            summ = ba.performBundleAdjustment()
            t_solve = time.perf_counter() - t_solve
            after = ba.calcReprojectionError(1.0)
            solve = {"iterations": int(summ.iterations), "successful_steps": int(summ.successful_steps), "termination": int(summ.termination),
                     "seconds": t_solve, "create_seconds": t_create, "seconds_in_sweeps": float(summ.seconds_sweep), "seconds_in_linear_solve": float(summ.seconds_linear_solve),
                     "initial_cost": float(summ.initial_cost), "final_cost": float(summ.final_cost),
                     "rms_initial_px": [float(before.std_x), float(before.std_y)],
                     "final_rms_reproj_px": [float(after.std_x), float(after.std_y)],
                     "max_abs_error_px": [float(after.mae_x), float(after.mae_y)],
                     "inliers": int(after.num_inliers), "observations": int(after.num_points)}
        except Exception as e:  # noqa: BLE001
            solve = {"error": repr(e)}

    out = None
    if rank == 0:
        copies = 1 if strong else world
        F_tot, P_tot = spec.n_frames * copies, spec.n_points * copies
        n_loc = info.n_obs_local
        n_red = 17 + 6 * F_tot
        p_loc = info.n_points_local if strong else spec.n_points   # per-rank share of the algorithmic bytes (rank 0's shard)
        b_obs = 24 if args.precision == 1 else 40   # SURVEY.md 8(d); the fp32 stream this library reads is 12 B per observation (du, dv, lens index)
        b_kernel = accumulate_kernel_bytes(n_loc, p_loc, spec.n_frames, b_obs)
        n_red_loc = 17 + 6 * spec.n_frames
        # what the band + arrow layout of the reduced system stores: F (bw + 1) pose blocks of 36 + (NA + 1) arrow rows
        nc_live = 5 + (spec.config & 3) + (2 if spec.config & 4 else 0)
        na = nc_live + 3 * int(info.n_promoted)
        stored_doubles = F_tot * int(info.max_window_frames) * 36 + (na + 1) * (6 * F_tot + na + 1)
        b_sweep = algorithmic_bytes(n_loc, p_loc, spec.n_frames, n_red_loc, b_obs)
        t_kernel = prof.ms_accumulate * 1e-3
        t_total = prof.ms_total * 1e-3
        achieved = b_kernel / t_kernel / 1e9
        out = {
            "metric": "micro-image obs/sec through Jacobian+Schur at 1M obs",
            "value": n_obs_total * args.steps / dt,
            "unit": "obs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64" if args.precision == 0 else "f32 residual+Jacobian / f64 accumulate+solve", "data": "synthetic",
            "config": {"lens_selection": "reference generator (lifcal_mla_project)" if use_web else "K-nearest stand-in",
                       "workload": f"{args.workload}: {F_tot} frames, {P_tot} points, {n_obs_total} micro-image observations, "
                                   f"window {spec.window}, config {spec.config:#x} (2 radial + tangential, mlCenterAdj, Cauchy(0.5), refine poses+points)",
                       "obs_per_gpu": n_loc, "n_reduced": n_red, "sharding": "by 3D point" if world > 1 else "single GPU", "comm": comm_used,
                       "exchange_ms_per_sweep": (prof.ms_exchange if world > 1 else None), "preheat_ms": args.preheat_ms,
                       "deterministic": bool(args.deterministic)},
            "roofline": {"bound": "hbm", "kernel": "k_sweep (residual+Jacobian+block accumulation)",
                         "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved * 1e9 / HBM_PEAK,
                         "traffic": load_traffic(args.workload)[0], "traffic_source": load_traffic(args.workload)[1],
                         "algorithmic_bytes_per_launch": b_kernel, "kernel_ms": prof.ms_accumulate,
                         "kernel_launches_timed": int(prof.n_sampled),   # of the K timed steps (every 8th: see profile_begin above)
                         "special_points": int(prof.special_points),   # points on the global-atomic kernels (their time is in ms_outside_dominant_kernel)
                         "whole_sweep": {"algorithmic_bytes": b_sweep, "ms": prof.ms_total, "ms_outside_dominant_kernel": prof.ms_schur,
                                         "achieved": b_sweep / t_total / 1e9, "frac": b_sweep / t_total / HBM_PEAK,
                                         # SURVEY's formula charges S dense (n_red (n_red + 1) / 2 doubles); the band + arrow layout stores this much of it:
                                         "bytes_moved_actual": b_sweep - 8 * (n_red_loc * (n_red_loc + 1) // 2) + 8 * stored_doubles}},
            "cost": last.cost,
            "solve": solve,
        }
        out["roofline"]["fp64_valu"] = {"flop_per_obs": FP64_FLOP_PER_OBS, "achieved": FP64_FLOP_PER_OBS * n_loc / t_kernel / 1e12,
                                        "peak": FP64_PEAK / 1e12, "unit": "TFLOP/s", "frac": FP64_FLOP_PER_OBS * n_loc / t_kernel / FP64_PEAK}
    ba.close()
    if rank == 0:
        out["cpu_baseline"] = None
        out["parity"] = None
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"], ref = cpu_baseline(sc, capi.ProblemArrays.from_scene(sc))
                # GPU sweep against the CPU restatement on the identical input, in this very run (tests/test_gpu_scale.py asserts the bars)
                dS = np.sqrt(np.abs(np.diag(ref.S))) + 1e-300
                out["parity"] = {"against": "oracle.sweep (CPU restatement, dual-number Jacobian) on the identical input, radius 1e4",
                                 "cost_rel": abs(last.cost - ref.cost) / abs(ref.cost),
                                 "S_block_scaled_max": float(np.max(np.abs(last.S - ref.S) / np.outer(dS, dS))),
                                 "rhs_rel": float(np.max(np.abs(last.rhs - ref.rhs)) / (np.max(np.abs(ref.rhs)) + 1e-300))}
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
