"""oval_amd — MI355X (gfx950) data-pattern engine behind the API of
peng225/oval's internal/pattern and internal/object packages.

  oval_amd.pattern / oval_amd.object   the reference's function names
  oval_amd.engine                      batch API over the C ABI (include/oval_hip.h)
  oval_amd.batch                       descriptor tables for synthetic batches, sharding by worker key range
"""
from ._lib import (DESC_DTYPE, NO_BAD_UNIT, NUM_COUNTERS, RESULT_DTYPE, UNIT, OvalError, lib)  # noqa: F401

__all__ = ["DESC_DTYPE", "RESULT_DTYPE", "NUM_COUNTERS", "NO_BAD_UNIT", "UNIT", "OvalError", "lib"]
