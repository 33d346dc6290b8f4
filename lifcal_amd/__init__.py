"""lifcal_amd — MI355X-native plenoptic bundle adjustment behind LiFCal's performBundleAdjustment seam.

Only what the hot path needs: csrc/ (HIP kernels + C ABI), the ctypes view of include/lifcal_ba.h,
the host mirror of the reference entry points and the synthetic scene generator.
"""
from . import _capi, scene  # noqa: F401
from .bundle_adjustment import (BundleAdjustment, LifcalError, make_config, plan, comm_unique_id, initPlenopticParameters,  # noqa: F401
                                performBundleAdjustmentWindowed)
from .mla import MicroLensGrid, RawObservations  # noqa: F401
