import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lifcal_amd import BundleAdjustment, _capi as capi, scene
import oracle
np.set_printoptions(linewidth=250, precision=3)
S = scene.SceneSpec
seed=int(sys.argv[1]); ncon=int(sys.argv[2])
sc = scene.make_scene(S(6, 40, None, 0x506, seed, n_constraints=ncon))
pa = capi.ProblemArrays.from_scene(sc)
F, P, N = 6, 40, sc.n_obs
Jc = []
for i in range(N):
    f = sc.fr[i]; p = sc.pt[i]
    r, J = oracle.residual_block(sc.config, 3, pa.cam, pa.views[6*f:6*f+6], pa.pts[3*p:3*p+3], sc.u[i], sc.v[i], sc.mcx[i], sc.mcy[i], sc.spx, sc.scale)
    Jc.append(J[:, :9])
Jc = np.concatenate(Jc, 0)
Bcc = Jc.T @ Jc
os.environ["LIFCAL_DEBUG_SKIP_SCHUR"] = "1"
ba = BundleAdjustment(pa)
got = ba.sweep(1e30, want_matrices=True)
G = got.S[:9, :9]
d = np.sqrt(np.diag(Bcc))
print(seed, ncon, "max rel diff", np.abs((G - Bcc) / np.outer(d, d)).max())
