"""Comparator for the reduced solve (SURVEY.md §2 work list K3): rocSOLVER dpotrf + dpotrs on the DENSE reduced system S, the
linear solver the reference asks ceres for (DENSE_SCHUR, src/CameraCalibration.cpp:956 -> Eigen LLT of the (17 + 6F)^2 matrix),
against this library's block-banded + arrow Cholesky of the same system.  Run on the GPU box:

    python tools/rocsolver_comparator.py [cfg3 metric]

Per workload: n_red, rocSOLVER potrf / potrs times (HIP events, best of 5 after a warm-up), residual |S x - rhs| / |rhs| of both
solutions, and the library's own linear-solve time per LM iteration (k_band_chol_w + k_band_backsolve_w, from HIP events around
lifcal_ba's stream via the profile of a full solve).  rocSOLVER is a comparator only: the product never calls it."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from lifcal_amd import BundleAdjustment, _capi as capi, scene


def main():
    names = sys.argv[1:] or ["cfg3", "metric"]
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")   # the copies PyTorch-ROCm ships (same HIP runtime as its streams)
    def load(n):
        p = os.path.join(tl, n)
        return C.CDLL(p if os.path.exists(p) else n)
    rb = load("librocblas.so")
    rs = load("librocsolver.so")
    for f in (rs.rocsolver_dpotrf, rs.rocsolver_dpotrs, rb.rocblas_create_handle, rb.rocblas_set_stream):
        f.restype = C.c_int
    handle = C.c_void_p()
    assert rb.rocblas_create_handle(C.byref(handle)) == 0
    stream = torch.cuda.current_stream().cuda_stream
    assert rb.rocblas_set_stream(handle, C.c_void_p(stream)) == 0
    LOWER = 122
    for name in names:
        sc = scene.make_scene(scene.baseline_spec(name))
        pa = capi.ProblemArrays.from_scene(sc)
        with BundleAdjustment(pa) as ba:
            sw = ba.sweep(1e4, want_matrices=True)
            n = sw.n_reduced
            live = np.flatnonzero(np.abs(sw.rhs) + (np.abs(sw.S).sum(1) - 1.0) != 0)   # identity rows (dead camera slots) stay in: harmless
            S = torch.from_numpy(np.ascontiguousarray(sw.S)).cuda()
            b = torch.from_numpy(sw.rhs.copy()).cuda()
            info = torch.zeros(1, dtype=torch.int32, device="cuda")
            t_f, t_s = [], []
            for rep in range(6):
                A = S.clone(); x = b.clone()
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
                rc = rs.rocsolver_dpotrf(handle, LOWER, n, C.c_void_p(A.data_ptr()), n, C.c_void_p(info.data_ptr()))
                e1.record()
                rc2 = rs.rocsolver_dpotrs(handle, LOWER, n, 1, C.c_void_p(A.data_ptr()), n, C.c_void_p(x.data_ptr()), n)
                e2.record()
                torch.cuda.synchronize()
                assert rc == 0 and rc2 == 0 and int(info.item()) == 0, (rc, rc2, int(info.item()))
                if rep:
                    t_f.append(e0.elapsed_time(e1)); t_s.append(e1.elapsed_time(e2))
            xr = x.cpu().numpy()
            res_r = np.linalg.norm(sw.S @ xr - sw.rhs) / np.linalg.norm(sw.rhs)
            # the library's own solve: full LM run, linear solve + candidate time per iteration
            t0 = time.perf_counter(); s = ba.performBundleAdjustment(); t_solve = time.perf_counter() - t0
        print(f"{name}: n_red {n} ({len(live)} live rows), dense S {n * n * 8 / 1e6:.1f} MB | rocSOLVER dpotrf {min(t_f):.3f} ms + dpotrs {min(t_s):.3f} ms "
              f"= {min(t_f) + min(t_s):.3f} ms (residual {res_r:.1e}) | lifcal_ba: {s.iterations} LM iterations in {t_solve * 1e3:.2f} ms, "
              f"linear solve + candidate evaluation {s.seconds_linear_solve / max(1, s.iterations) * 1e3:.3f} ms per iteration, sweeps {s.seconds_sweep / max(1, s.iterations + 1) * 1e3:.3f} ms each", flush=True)


if __name__ == "__main__":
    main()
