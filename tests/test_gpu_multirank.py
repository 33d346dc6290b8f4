"""Two ranks on ONE GPU: real inter-process exchange through the all-reduce hook (torch.distributed, gloo),
so sharding, the iteration-0 Jacobi all-reduce, the LM loop and the final point gather run as they do on N GPUs
(only RCCL itself is replaced).  Rank 0 compares with the single-process oracle."""
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


def _worker(rank, world, port, out_dir, spec_kw):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lifcal_amd import BundleAdjustment, _capi as capi, scene
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    sc = scene.make_scene(scene.SceneSpec(**spec_kw))
    pa = capi.ProblemArrays.from_scene(sc)
    o = capi.default_options_py(); o.rank = rank; o.world_size = world
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
    sw = ba.sweep(1e4, want_matrices=(rank == 0))
    summ = ba.performBundleAdjustment()
    st = ba.calcReprojectionError()
    info = ba.info()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), cam=pa.cam, views=pa.views, pts=pa.pts, cost0=sw.cost,
             S=sw.S if rank == 0 else np.zeros(1), rhs=sw.rhs if rank == 0 else np.zeros(1),
             it=summ.iterations, term=summ.termination, final=summ.final_cost, n_local=info.n_obs_local,
             stats=np.array([st.std_x, st.std_y, st.num_points, st.num_inliers]))
    ba.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("spec_kw", [
    dict(n_frames=24, n_points=160, window=6, config=0xF06, seed=1401, outlier_fraction=0.02),
    dict(n_frames=8, n_points=60, window=None, config=0x506, seed=1402, n_constraints=3),
], ids=["windowed_robust", "constraints"])
def test_two_ranks_match_the_single_process_oracle(built, tmp_path, spec_kw):
    import oracle
    from lifcal_amd import _capi as capi, scene
    from tests.helpers import scaled_max_err, vec_err
    world = 2
    port = 29700 + (os.getpid() % 1500)
    mp.spawn(_worker, args=(world, port, str(tmp_path), spec_kw), nprocs=world, join=True)
    r0 = np.load(os.path.join(str(tmp_path), "rank0.npz")); r1 = np.load(os.path.join(str(tmp_path), "rank1.npz"))
    sc = scene.make_scene(scene.SceneSpec(**spec_kw))
    ref = oracle.sweep(capi.ProblemArrays.from_scene(sc), radius=1e4, threads=4)
    assert abs(float(r0["cost0"]) - ref.cost) <= 1e-12 * ref.cost and abs(float(r1["cost0"]) - ref.cost) <= 1e-12 * ref.cost
    assert scaled_max_err(r0["S"], ref.S) < 1e-9 and vec_err(r0["rhs"], ref.rhs) < 1e-9
    assert int(r0["n_local"]) + int(r1["n_local"]) == sc.n_obs and min(int(r0["n_local"]), int(r1["n_local"])) > 0.3 * sc.n_obs
    pb = capi.ProblemArrays.from_scene(sc)
    so = oracle.solve(pb, threads=4)
    for r in (r0, r1):   # every rank ends with the full, identical result
        assert (int(r["it"]), int(r["term"])) == (so.iterations, so.termination)
        assert abs(float(r["final"]) - so.final_cost) <= 1e-8 * so.final_cost
        assert np.allclose(r["cam"][:5], pb.cam[:5], rtol=1e-6)
        assert np.allclose(r["pts"], pb.pts, rtol=0, atol=1e-6 * (1 + np.abs(pb.pts).max()))
        assert np.allclose(r["views"], pb.views, rtol=0, atol=1e-6 * (1 + np.abs(pb.views).max()))
    assert np.array_equal(r0["pts"], r1["pts"]) and np.array_equal(r0["cam"], r1["cam"])
    stt = oracle.reproj_stats(capi.ProblemArrays(sc.u, sc.v, sc.mcx, sc.mcy, sc.pt, sc.fr, r0["cam"], r0["views"], r0["pts"], sc.spx, sc.scale, sc.config))
    assert abs(float(r0["stats"][0]) - stt.std_x) < 1e-9 and int(r0["stats"][2]) == sc.n_obs
