"""Diagnostic: solve one deformed stress scene (tests/test_gpu_stress.py family) on the GPU and on the oracle, print both summaries."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from lifcal_amd import BundleAdjustment
from tests.test_gpu_stress import deformed_problem

k = int(sys.argv[1])
spec, mk, n = deformed_problem(k)
print(f"case {k}: F={spec.n_frames} P={spec.n_points} window={spec.window} cfg={spec.config:#x} recalib={spec.recalib} n={n}", flush=True)
pg = mk()
t0 = time.time()
with BundleAdjustment(pg) as ba:
    sg = ba.performBundleAdjustment()
    st = ba.calcReprojectionError()
print("gpu   ", sg.iterations, sg.termination, sg.successful_steps, sg.unsuccessful_steps, sg.final_cost, f"{time.time() - t0:.2f}s", flush=True)
pb = mk()
so = oracle.solve(pb, threads=8)
print("oracle", so.iterations, so.termination, so.successful_steps, so.unsuccessful_steps, so.final_cost)
print("cam diff", np.max(np.abs(pg.cam - pb.cam) / (np.abs(pb.cam) + 1e-300)))
