"""The oracle against the committed golden vectors (tests/golden/, made by tools/make_golden.py)."""
import numpy as np
import pytest

import oracle
from tests.helpers import GOLDEN, load_golden, scaled_max_err, vec_err


@pytest.mark.parametrize("name", GOLDEN)
def test_oracle_reproduces_golden_sweep(name):
    g, pa = load_golden(name)
    sw = oracle.sweep(pa, radius=float(g["radius"]))
    assert abs(sw.cost - float(g["cost"])) <= 1e-13 * float(g["cost"])
    assert scaled_max_err(sw.S, g["S"]) < 1e-11
    assert vec_err(sw.rhs, g["rhs"]) < 1e-10
    assert vec_err(sw.gradient_reduced, g["gradient_reduced"]) < 1e-11
    assert vec_err(sw.point_gradient, g["point_gradient"]) < 1e-11 or np.abs(g["point_gradient"]).max() == 0
    assert vec_err(sw.point_hessian_inv, g["point_hessian_inv"]) < 1e-10 or np.abs(g["point_hessian_inv"]).max() == 0


@pytest.mark.parametrize("name", GOLDEN)
def test_oracle_reproduces_golden_blocks_and_solve(name):
    g, pa = load_golden(name)
    for i in range(g["sample_r"].shape[0]):
        f, p = int(g["fr"][i]), int(g["pt"][i])
        r, J = oracle.residual_block(int(g["config"]), int(g["arity"]), pa.cam, pa.views[6 * f:6 * f + 6], pa.pts[3 * p:3 * p + 3],
                                     g["u"][i], g["v"][i], g["mcx"][i], g["mcy"][i], float(g["spx"]), float(g["scale"]))
        assert np.allclose(r, g["sample_r"][i], rtol=0, atol=1e-11)
        assert np.allclose(J, g["sample_J"][i], rtol=1e-11, atol=1e-12 * np.abs(g["sample_J"][i]).max())
    s = oracle.solve(pa)
    assert s.iterations == int(g["solve_iterations"]) and s.termination == int(g["solve_termination"])
    assert abs(s.final_cost - float(g["solve_final_cost"])) <= 1e-9 * float(g["solve_final_cost"])
    assert np.allclose(pa.cam, g["solve_cam"], rtol=1e-7, atol=1e-12)
