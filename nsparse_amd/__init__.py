"""nsparse_amd: MI355X-native hash SpGEMM + AMB SpMV behind the reference's C entry points.

The product is the C-ABI library (include/nsparse.h -> nsparse_amd/lib/libnsparse_{d,s}.so,
sources in nsparse_amd/csrc/).  This Python package only holds the ctypes binding used by the
tests and the bench harness, and the row-sharded multi-GPU driver.
"""
from .capi import (load, load_vendor, load_dist, Lib, VendorLib, DistLib, sfCSR, sfAMB, sfPlan, SpgemmStats,  # noqa: F401
                   SIGNATURES, VENDOR_SIGNATURES, DIST_SIGNATURES, DIST_ID_BYTES)
