"""MI355X-native Instance Stixels column-DP core (host-side Python helpers).

The product is the HIP library `instance_stixels_amd/lib/libis_core.so` (C ABI in
include/instance_stixels_core.h) and the header-compatible C++ `Stixels` class built on it.
This package only carries the ctypes bindings, the synthetic-input generator and the
multi-GPU batch sharding used by tests and bench.py.
"""
from .config import (StixelConfig, StixelParams, SECTION_DTYPE, PRESETS, make_config,  # noqa: F401
                     GROUND, OBJECT, SKY)
