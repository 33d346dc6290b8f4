"""N > 1 on CPU (gloo, world_size 2): the point sharding the library plans must make the per-rank partial
reduced systems add up to the single-rank system with ONE sum all-reduce.  Each rank runs
the oracle on its own shard of the observations; torch.distributed (gloo) does the reduction."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from lifcal_amd import _capi as capi, scene, plan
    sc = scene.make_scene(scene.SceneSpec(20, 90, 6, 0xF06, 1001, outlier_fraction=0.02))
    full = capi.ProblemArrays.from_scene(sc)
    info, order, owner = plan(full, rank, world)
    mine = order[order != 0xFFFFFFFF]
    shard = capi.ProblemArrays(sc.u[mine], sc.v[mine], sc.mcx[mine], sc.mcy[mine], sc.pt[mine], sc.fr[mine],
                               sc.cam0, sc.views0, sc.pts0, sc.spx, sc.scale, sc.config)
    o = capi.default_options_py(); o.jacobi_scaling = 0
    # undamped partial systems: radius so large that the LM diagonal vanishes next to J^T J
    part = oracle.sweep(shard, radius=1e290, options=o)
    S = part.S.copy(); rhs = part.rhs.copy()
    # rows this shard does not touch come back as identity rows from the oracle; they are not contributions
    frames_here = np.zeros(sc.spec.n_frames, bool); frames_here[sc.fr[mine]] = True
    for f in np.flatnonzero(~frames_here):
        for k in range(6):
            S[17 + 6 * f + k, 17 + 6 * f + k] -= 1.0
    dead = np.arange(9, 17)
    S[dead, dead] -= 1.0
    t = torch.from_numpy(np.concatenate([S.reshape(-1), rhs, [part.cost]]))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        ref = oracle.sweep(full, radius=1e290, options=o)
        n = ref.n_reduced
        Ssum = t[: n * n].numpy().reshape(n, n); rsum = t[n * n: n * n + n].numpy(); csum = float(t[-1])
        Sref = ref.S.copy(); Sref[dead, dead] -= 1.0
        d = np.sqrt(np.abs(np.diag(ref.S))) + 1e-300
        err = np.max(np.abs(Ssum - Sref) / np.outer(d, d))
        np.save(os.path.join(out_dir, "result.npy"), np.array([err, np.abs(rsum - ref.rhs).max() / np.abs(ref.rhs).max(), abs(csum - ref.cost) / ref.cost,
                                                             float(len(mine)), float(sc.n_obs)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_partial_systems_sum_to_the_full_system(tmp_path, built):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    err_s, err_r, err_c, n_mine, n_all = np.load(os.path.join(str(tmp_path), "result.npy"))
    assert err_s < 1e-11 and err_r < 1e-11 and err_c < 1e-13
    assert 0.3 * n_all < n_mine < 0.7 * n_all
