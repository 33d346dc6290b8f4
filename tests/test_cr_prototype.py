"""tools/cr_prototype.py — the block odd-even reduction of bandchol3.hpp at index level, in numpy — against a dense solve."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import cr_prototype as cr


@pytest.mark.parametrize("F,bw,NA", [(40, 3, 5), (37, 4, 17), (334, 9, 17), (9, 2, 3), (8, 2, 3), (5, 1, 1), (64, 8, 17)])
def test_block_odd_even_reduction_solves_the_band_arrow_system(F, bw, NA):
    S, rhs, Sband, Sarrow = cr.random_system(F, bw, NA, seed=F + bw)
    x = cr.cr_solve(Sband, Sarrow, F, bw, NA)
    xr = np.linalg.solve(S, rhs)
    assert np.max(np.abs(x - xr)) <= 1e-11 * np.max(np.abs(xr))
