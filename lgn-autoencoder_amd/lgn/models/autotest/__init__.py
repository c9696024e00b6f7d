"""Equivariance / permutation harness of the autoencoder (counterpart of the reference's
lgn/models/autotest/: lgn_tests.py:23-423, utils.py:11-140, lgn/g_lib/rotations.py:7-156)."""
from .lgn_tests import (lgn_tests, covariance_test, permutation_invariance_test, lorentz_D, rotate_rep,  # noqa: F401
                        cartesian_lorentz, check_equivariance, DEFAULT_THRESHOLDS)
