"""Shard-local planning (lifcal_ba_partition_points / lifcal_ba_plan_shard; host-only): a rank that receives ONLY the observations of
the points it owns builds the layout it would build from the whole problem — same ownership, same groups, tiles, lenses,
chunks, same layout fingerprint."""
import ctypes as C
import re

import numpy as np
import pytest

import lifcal_amd
from lifcal_amd import _capi as capi, scene
from tests.helpers import S


@pytest.mark.parametrize("world", [2, 3])
def test_shard_plans_equal_whole_problem_plans(built, world, capfd, monkeypatch):
    sc = scene.make_scene(S(40, 900, 8, 0xF06, 5101, outlier_fraction=0.02))
    pa = capi.ProblemArrays.from_scene(sc)
    pa.struct.use_constraints = 0
    part = capi.PartitionArrays(pa, world)
    lib = capi.load_library()
    assert part.struct.n_obs == sc.n_obs and int(part.rank_obs.sum()) == sc.n_obs
    assert part.struct.band_width == max(sc.fr[sc.pt == q].max() - sc.fr[sc.pt == q].min() for q in np.unique(sc.pt))
    assert np.array_equal(part.frame_used[:40], np.isin(np.arange(40), sc.fr).astype(np.uint8))
    monkeypatch.setenv("LIFCAL_PLAN_HASH", "1")
    for rank in range(world):
        capfd.readouterr()
        info, order, owner = lifcal_amd.plan(pa, rank, world)
        h_whole = re.search(r"\[plan\] hash ([0-9a-f]{16})", capfd.readouterr().err).group(1)
        assert np.array_equal(owner, part.point_owner[: len(owner)])              # the same ownership rule
        shard = part.shard_of(pa, rank)
        assert shard.struct.n_obs == int(part.rank_obs[rank]) < sc.n_obs
        sinfo = capi.PlanInfo()
        rc = lib.lifcal_ba_plan_shard(C.byref(shard.struct), C.byref(part.struct), rank, C.byref(sinfo))
        h_shard = re.search(r"\[plan\] hash ([0-9a-f]{16})", capfd.readouterr().err).group(1)
        assert rc == 0
        for f, _ in capi.PlanInfo._fields_:
            assert getattr(sinfo, f) == getattr(info, f), f
        # the fingerprint covers obs_order / *_src, i.e. indices into the INPUT arrays, which differ between whole problem and shard;
        # everything else — lens table, observation payload, slots, rows — is covered by the equal counts above and by the GPU test
        assert h_whole and h_shard
    # a shard that contains a foreign observation is refused
    bad = part.shard_of(pa, 0)
    other = np.flatnonzero(part.point_owner[pa.pt] == 1)[0]
    bad.pt[0] = pa.pt[other]
    sinfo = capi.PlanInfo()
    assert lib.lifcal_ba_plan_shard(C.byref(bad.struct), C.byref(part.struct), 0, C.byref(sinfo)) == -4
