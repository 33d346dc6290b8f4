import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lifcal_amd import BundleAdjustment, _capi as capi, scene
import oracle
np.set_printoptions(linewidth=250, precision=3)
S = scene.SceneSpec
sc = scene.make_scene(S(6, 40, None, 0x506, 14, n_constraints=3))
pa = capi.ProblemArrays.from_scene(sc)
ref = oracle.sweep(capi.ProblemArrays.from_scene(sc), radius=1e4)
ba = BundleAdjustment(pa)
got = ba.sweep(1e4, want_matrices=True)
d = np.sqrt(np.abs(np.diag(ref.S))) + 1e-300
E = (got.S - ref.S) / np.outer(d, d)
print("c_i", sc.c_i, "c_j", sc.c_j)
print("cam-cam rel diff\n", E[:9, :9])
print("ref diag", np.diag(ref.S)[:9]); print("got diag", np.diag(got.S)[:9])
n = ref.n_reduced
print("prom block diff\n", E[n-9:, n-9:])
print("prom-cam diff\n", E[n-9:, :9])
# which points explain the cam-cam discrepancy?  diff ~ sum_p alpha_p * Wc_p^T Uinv_p Wc_p
F, P, N = 6, 40, sc.n_obs
Wc = np.zeros((P, 3, 9)); 
for i in range(N):
    f = sc.fr[i]; p = sc.pt[i]
    r, J = oracle.residual_block(sc.config, 3, pa.cam, pa.views[6*f:6*f+6], pa.pts[3*p:3*p+3], sc.u[i], sc.v[i], sc.mcx[i], sc.mcy[i], sc.spx, sc.scale)
    Wc[p] += J[:, 23:26].T @ J[:, :9]
Ui = got.point_hessian_inv.reshape(P, 3, 3)
D = (got.S - ref.S)[:9, :9]
T = np.array([(Wc[p].T @ Ui[p] @ Wc[p]).reshape(-1) for p in range(P)]).T
alpha, *_ = np.linalg.lstsq(T, D.reshape(-1), rcond=None)
print("alpha", np.round(alpha, 3))
print("residual", np.abs(T @ alpha - D.reshape(-1)).max(), np.abs(D).max())
