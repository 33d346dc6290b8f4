"""Randomised multi-rank campaign on ONE GPU: 2-4 processes exchange through the collective hooks (torch.distributed / gloo, the
worker of tests/test_gpu_multirank.py), random scene shapes, arities, distance constraints and box bounds; every rank must end on
the single-process oracle's LM trajectory and all ranks with bitwise identical parameters.  A case that does not finish in two
minutes (ranks that took different branches wait for each other for ever) is reported and its processes are ended.
Run on the GPU box: gpurun -- tools/gpurun.sh run tools/fuzz_multirank.py [n_cases] [first_seed]"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, ".")
import torch.multiprocessing as mp                                  # noqa: E402
import oracle                                                       # noqa: E402
from lifcal_amd import _capi as capi, scene                         # noqa: E402
from tests.helpers import bounded_problem, free_port                # noqa: E402
from tests.test_gpu_multirank import _worker                        # noqa: E402

if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
    bad = 0
    t_start = time.time()
    for case in range(n_cases):
        rng = np.random.default_rng(seed0 + case)
        world = int(rng.integers(2, 5))
        F = int(rng.integers(6, 70)); P = int(rng.integers(40, 400))
        window = None if rng.random() < 0.25 else int(rng.integers(2, min(F, 22) + 1))
        nr = int(rng.integers(0, 3)); tan = int(rng.integers(0, 2))
        arity = [0x500, 0x500, 0x500, 0x100, 0x000][int(rng.integers(0, 5))]
        cfg = nr | (tan << 2) | arity | (0x200 if rng.random() < 0.5 else 0) | (0x800 if rng.random() < 0.5 else 0)
        ncons = int(rng.integers(1, 6)) if (arity == 0x500 and rng.random() < 0.25) else 0
        bounded = bool(arity == 0x500 and ncons == 0 and nr == 2 and tan == 1 and rng.random() < 0.3)
        outl = float(rng.choice([0.0, 0.02])) if cfg & 0x200 else 0.0
        kw = dict(n_frames=F, n_points=P, window=window, config=cfg, seed=seed0 + case, outlier_fraction=outl, n_constraints=ncons)
        if bounded: kw["config"] = 0xF06 if cfg & 0x200 else 0x506
        tag = f"world {world} {kw} bounded={bounded}"
        out = tempfile.mkdtemp(prefix="fuzzmr_")
        wkw = dict(kw); 
        if bounded: wkw["bounded"] = True
        ctx = mp.spawn(_worker, args=(world, free_port(), out, wkw, "allgather"), nprocs=world, join=False)
        t0 = time.time(); done = False
        try:
            while time.time() - t0 < 120:
                if ctx.join(timeout=1.0): done = True; break
        except Exception as e:  # noqa: BLE001
            bad += 1; print(f"ERROR {tag}: worker raised {e!r}", flush=True); continue
        if not done:
            bad += 1
            print(f"HANG  {tag}: not finished after 120 s, ending the ranks", flush=True)
            for p in ctx.processes:
                if p.is_alive(): p.terminate()
            for p in ctx.processes: p.join(5)
            continue
        try:
            sc = scene.make_scene(scene.SceneSpec(**kw))
            pb = bounded_problem(sc) if bounded else capi.ProblemArrays.from_scene(sc)
            so = oracle.solve(pb, threads=8)
            rs = [np.load(os.path.join(out, f"rank{r}.npz")) for r in range(world)]
            ok = True; why = ""
            for rr in rs:
                if (int(rr["it"]), int(rr["term"]), int(rr["steps"][0]), int(rr["steps"][1])) != (so.iterations, so.termination, so.successful_steps, so.unsuccessful_steps):
                    ok = False; why += f" trajectory {(int(rr['it']), int(rr['term']), int(rr['steps'][0]), int(rr['steps'][1]))} vs {(so.iterations, so.termination, so.successful_steps, so.unsuccessful_steps)}"
                if abs(float(rr["final"]) - so.final_cost) > 1e-7 * so.final_cost: ok = False; why += f" cost {float(rr['final'])} vs {so.final_cost}"
                if not (np.array_equal(rr["cam"], rs[0]["cam"]) and np.array_equal(rr["views"], rs[0]["views"]) and np.array_equal(rr["pts"], rs[0]["pts"])): ok = False; why += " ranks differ in bits"
            if not ok:
                bad += 1; print(f"FAIL  {tag}:{why}", flush=True)
            elif case % 5 == 0:
                print(f"ok    {tag}: obs {sc.n_obs} it {so.iterations} ({time.time() - t_start:.0f} s)", flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1; print(f"ERROR {tag}: {e!r}", flush=True)
    print(f"{n_cases} multi-rank cases, {bad} failures, {time.time() - t_start:.0f} s")
