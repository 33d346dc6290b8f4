"""The analytic-Jacobian arm of the CPU oracle (oracle/analytic.hpp, timed by bench.py as the second CPU baseline) against the
dual-number arm (oracle/model.hpp through oracle/jet.hpp = what ceres::AutoDiffCostFunction evaluates on the reference's functor,
src/BundleAdjustment/BundleAdjustment.h:199-222): both must form the same reduced system for every instantiation of the model
and every functor arity.  Two derivations of the same Jacobian, written independently, agreeing to round-off."""
import numpy as np
import pytest

import oracle
from lifcal_amd import scene
from tests.helpers import SMALL_CASES, problem, scaled_max_err, vec_err


@pytest.mark.parametrize("name,spec", SMALL_CASES, ids=[c[0] for c in SMALL_CASES])
def test_analytic_arm_forms_the_same_reduced_system(name, spec):
    sc = scene.make_scene(spec)
    a = oracle.sweep(problem(sc), radius=1e3, threads=2)
    b = oracle.sweep(problem(sc), radius=1e3, threads=2, analytic=True)
    assert a.rc == 0 and b.rc == 0
    assert abs(a.cost - b.cost) <= 1e-13 * a.cost
    assert scaled_max_err(b.S, a.S) < 1e-10
    assert vec_err(b.rhs, a.rhs) < 1e-10
    assert vec_err(b.gradient_reduced, a.gradient_reduced) < 1e-11
    if sc.config & 0x400 and sc.config & 0x100:
        assert vec_err(b.point_gradient, a.point_gradient) < 1e-11
        assert vec_err(b.point_hessian_inv, a.point_hessian_inv) < 1e-9


def test_analytic_arm_with_negative_stored_parameters():
    """sign folding (BundleAdjustment.h:123-133): the derivative w.r.t. a negative stored value flips sign in both arms"""
    sc = scene.make_scene(SMALL_CASES[1][1])
    def mk():
        pa = problem(sc)
        pa.cam[0] = -pa.cam[0]; pa.cam[2] = -pa.cam[2]
        return pa
    a = oracle.sweep(mk(), radius=1e3)
    b = oracle.sweep(mk(), radius=1e3, analytic=True)
    assert abs(a.cost - b.cost) <= 1e-13 * a.cost
    assert vec_err(b.gradient_reduced, a.gradient_reduced) < 1e-11 and scaled_max_err(b.S, a.S) < 1e-10
