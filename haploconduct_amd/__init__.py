"""haploconduct_amd — MI355X-native edge calculation for HaploConduct's overlap graph.

The product is ``csrc/libhcedge.so`` (hand-written HIP kernels for gfx950 behind the
C ABI of ``include/hcedge.h``).  This package is the thin Python host side used by
the tests and by ``bench.py``: ctypes bindings, record dtypes, a synthetic workload
generator, and a mirror of the reference's ``EdgeCalculator`` interface.

There is no CPU fallback: importing works without a GPU (so the CPU test-suite can
check the ABI), but creating a context without a HIP device raises ``HcError``.
"""
from ._native import HcError, lib, lib_path, device_count, version  # noqa: F401
from .records import OVERLAP_DTYPE, RESULT_DTYPE, Settings, CLS_NAMES  # noqa: F401
from .readstore import ReadSet  # noqa: F401
from .scorer import EdgeScorer  # noqa: F401
