"""SURVEY.md 8(f) f2 on the GPU: lifcal_init_plenoptic (device reduction + host 2x2 solve, through the C ABI) against the
oracle's restatement of reference CameraCalibration::initPlenopticParameters (src/CameraCalibration.cpp:456-499)."""
import numpy as np
import pytest

import oracle
from lifcal_amd import LifcalError, initPlenopticParameters, _capi as capi, scene
from tests.test_oracle_init_params import init_inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("spec,noise", [
    (scene.SceneSpec(6, 80, None, 0x506, 601), 0.0),
    (scene.SceneSpec(40, 3000, 8, 0xF06, 602), 0.05),
    (scene.baseline_spec("cfg3"), 0.02),
])
def test_matches_the_oracle(built, spec, noise):
    sc, arrs, (v, bL) = init_inputs(spec, seed=spec.seed, noise=noise)
    arrs.vdepth[::11] = 1.9                     # masked rows (reference :485)
    ref = oracle.init_plenoptic(arrs)
    w2c = np.transpose(arrs.w2c.reshape(-1, 4, 4), (0, 2, 1))
    got = initPlenopticParameters(arrs.vdepth, arrs.fr, arrs.pt, w2c, arrs.pts, sc.spec.fL)
    assert got.rank == ref.rank == 2 and got.n_used == ref.n_used
    assert abs(got.B_init - ref.B_init) <= 1e-11 * abs(ref.B_init)
    assert abs(got.bL0_init - ref.bL0_init) <= 1e-11 * abs(ref.bL0_init)
    if noise == 0.0:                            # the truth is known
        assert abs(got.B_init - sc.spec.B) < 1e-8 and abs(got.bL0_init - sc.spec.bL0) < 1e-7


def test_degenerate_and_invalid_inputs(built):
    sc, arrs, _ = init_inputs(scene.SceneSpec(4, 30, None, 0x506, 603))
    w2c = np.transpose(arrs.w2c.reshape(-1, 4, 4), (0, 2, 1))
    v = np.full(len(arrs.vdepth), 2.5)          # parallel columns: minimum-norm solution, rank 1
    arrs.vdepth[:] = v
    ref = oracle.init_plenoptic(arrs)
    got = initPlenopticParameters(v, arrs.fr, arrs.pt, w2c, arrs.pts, sc.spec.fL)
    assert got.rank == ref.rank == 1
    assert abs(got.B_init - ref.B_init) <= 1e-10 * abs(ref.B_init) and abs(got.bL0_init - ref.bL0_init) <= 1e-10 * abs(ref.bL0_init)
    got = initPlenopticParameters(np.ones(len(v)), arrs.fr, arrs.pt, w2c, arrs.pts, sc.spec.fL)   # every row masked
    assert got.rank == 0 and got.n_used == 0 and got.B_init == 0.0 and got.bL0_init == 0.0
    bad = arrs.fr.copy(); bad[3] = 10_000
    with pytest.raises(LifcalError):
        initPlenopticParameters(v, bad, arrs.pt, w2c, arrs.pts, sc.spec.fL)
    got = initPlenopticParameters(np.zeros(0), np.zeros(0, np.uint32), np.zeros(0, np.uint32), w2c, arrs.pts, sc.spec.fL)   # empty input
    assert got.n_used == 0 and got.rank == 0
