"""Oracle side of the constant-pose switch (lo_set_fixed_frames): ceres SetParameterBlockConstant semantics on a pose block."""
import numpy as np

import oracle
from lifcal_amd import scene
from tests.helpers import S, problem


def test_constant_poses_in_the_oracle():
    sc = scene.make_scene(S(6, 40, None, 0xF06, 2101, outlier_fraction=0.03))
    mask = np.zeros(6, np.uint8); mask[[1, 4]] = 1
    free = oracle.sweep(problem(sc), radius=1e3)
    try:
        oracle.set_fixed_frames(mask)
        fx = oracle.sweep(problem(sc), radius=1e3)
        assert fx.cost == free.cost                                   # the residual blocks stay
        for f in (1, 4):
            blk = slice(17 + 6 * f, 17 + 6 * f + 6)
            assert np.array_equal(fx.S[blk, blk], np.eye(6)) and np.all(fx.rhs[blk] == 0) and np.all(fx.gradient_reduced[blk] == 0)
        keep = np.ones(fx.S.shape[0], bool)
        for f in (1, 4):
            keep[17 + 6 * f:17 + 6 * f + 6] = False
        # the remaining system is the free system with those rows and columns struck out, BEFORE the points are eliminated: compare
        # through the point-free part (camera x camera block differs only through the Schur terms of shared points, so check gradients)
        assert np.allclose(fx.gradient_reduced[keep], free.gradient_reduced[keep], rtol=1e-12, atol=1e-12)
        pb = problem(sc)
        so = oracle.solve(pb)
        v = pb.views.reshape(-1, 6); v0 = sc.views0.reshape(-1, 6)
        assert np.array_equal(v[[1, 4]], v0[[1, 4]]) and not np.allclose(v[[0, 2, 3, 5]], v0[[0, 2, 3, 5]], atol=1e-9)
        assert so.termination in (1, 2) and so.final_cost < 0.2 * so.initial_cost
    finally:
        oracle.set_fixed_frames(None)
    again = oracle.sweep(problem(sc), radius=1e3)
    assert np.array_equal(again.S, free.S)
