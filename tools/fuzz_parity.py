"""Randomised parity campaign: random scene shapes x configurations x options, GPU sweep and solve against the CPU restatement.
Not part of the test-suite (run on the GPU box: gpurun -- tools/gpurun.sh run tools/fuzz_parity.py [n_cases] [first_seed]);
a failing case is printed with the SceneSpec arguments that reproduce it, to be added to tests/ once understood."""
import sys, time, traceback
import numpy as np
sys.path.insert(0, ".")
import oracle                                                     # noqa: E402
from lifcal_amd import BundleAdjustment, _capi as capi, scene     # noqa: E402
from tests.helpers import S, problem, scaled_max_err, vec_err     # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
prec1 = len(sys.argv) > 3 and sys.argv[3] == "precision1"   # options.precision = 1 (fp32 evaluation): its own, wider bars
big = len(sys.argv) > 3 and sys.argv[3] == "big"             # long sequences: 100-400 frames, 500-4000 points (the block odd-even reduction, many blocks)
bad = 0
t_start = time.time()
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    F = int(rng.integers(3, 70))
    P = int(rng.integers(20, 420))
    window = None if rng.random() < 0.3 else int(rng.integers(2, min(F, 26) + 1))
    if big:
        F = int(rng.integers(100, 400)); P = int(rng.integers(500, 4000)); window = int(rng.integers(2, 13))
    nr = int(rng.integers(0, 3)); tan = int(rng.integers(0, 2))
    arity = [0x000, 0x100, 0x400, 0x500, 0x500, 0x500][int(rng.integers(0, 6))]
    cfg = nr | (tan << 2) | arity | (0x200 if rng.random() < 0.5 else 0) | (0x800 if rng.random() < 0.5 else 0)
    ncons = int(rng.integers(0, 7)) if (arity == 0x500 and rng.random() < 0.3) else 0
    recalib = bool(arity == 0x500 and ncons == 0 and rng.random() < 0.15)
    outl = float(rng.choice([0.0, 0.02, 0.05])) if cfg & 0x200 else 0.0
    det = int(rng.random() < 0.35)
    kw = dict(outlier_fraction=outl, n_constraints=ncons, recalib=recalib)
    tag = f"S({F}, {P}, {window}, {cfg:#x}, {seed0 + case}, " + ", ".join(f"{k}={v}" for k, v in kw.items()) + f") det={det}"
    try:
        sc = scene.make_scene(S(F, P, window, cfg, seed0 + case, **kw))
        ref = oracle.sweep(problem(sc), radius=1e3, threads=8)
        o = capi.default_options_py(); o.deterministic = det; o.precision = 1 if prec1 else 0
        pa = problem(sc)
        with BundleAdjustment(pa, o) as ba:
            got = ba.sweep(1e3, want_matrices=True)
            s = ba.performBundleAdjustment()
            st = ba.calcReprojectionError()
        errs = dict(cost=abs(got.cost - ref.cost) / ref.cost, S=scaled_max_err(got.S, ref.S), rhs=vec_err(got.rhs, ref.rhs),
                    g=vec_err(got.gradient_reduced, ref.gradient_reduced))
        ok = errs["cost"] <= 1e-12 and errs["S"] < 1e-9 and errs["rhs"] < 1e-9 and errs["g"] < 1e-9
        if prec1:   # tests/test_gpu_precision1.py: cost 2e-6, S 2e-5 block-scaled, gradients 2e-4 (of the terms they cancel from: not checked here)
            ok = errs["cost"] <= 5e-6 and errs["S"] < 1e-4
        pb = problem(sc)
        so = oracle.solve(pb, threads=8)
        sto = oracle.reproj_stats(pb)
        traj = (s.iterations, s.successful_steps, s.unsuccessful_steps, s.termination) == (so.iterations, so.successful_steps, so.unsuccessful_steps, so.termination)
        dc = abs(s.final_cost - so.final_cost) / so.final_cost
        dst = max(abs(st.std_x - sto.std_x), abs(st.std_y - sto.std_y))
        ok2 = traj and dc <= 1e-8 and dst < 1e-7
        if prec1:   # the fp32 arm may take another path through the last iterations; what it must reach is the same minimum
            ok2 = dc <= 2e-6 and dst < 1e-4 and abs(s.iterations - so.iterations) <= max(3, so.iterations // 4)
        if not (ok and ok2):
            bad += 1
            print(f"FAIL {tag}: sweep {errs} | solve gpu {(s.iterations, s.successful_steps, s.unsuccessful_steps, s.termination)} oracle "
                  f"{(so.iterations, so.successful_steps, so.unsuccessful_steps, so.termination)} dcost {dc:.2e} dstd {dst:.2e}", flush=True)
        elif case % 10 == 0:
            print(f"ok   {tag}: obs {sc.n_obs} S {errs['S']:.1e} it {s.iterations} ({time.time() - t_start:.0f} s)", flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(f"ERROR {tag}: {e!r}", flush=True)
        traceback.print_exc()
print(f"{n_cases} cases, {bad} failures, {time.time() - t_start:.0f} s")
