"""MI355X-native AT-TPC Monte-Carlo engine (hot path of ATTPC/attpc_engine).

Same Python surface as the reference package for the hot path (``kinematics`` and
``detector`` sub-packages, module-level ``nuclear_map`` -- reference
``src/attpc_engine/__init__.py:1-3``); the arithmetic runs in hand-written HIP
kernels behind the C ABI of ``include/attpc_engine.h``.
"""
from .nuclear import NuclearDataMap, NucleusData
from .target import GasTarget

try:  # a user who has spyral_utils gets its (catima/AME backed) objects, as in the reference
    from spyral_utils.nuclear.nuclear_map import NuclearDataMap as _SpyralMap  # type: ignore

    nuclear_map = _SpyralMap()
except Exception:  # spyral_utils is not installed here: built-in light-nuclide table
    nuclear_map = NuclearDataMap()

__all__ = ["nuclear_map", "NuclearDataMap", "NucleusData", "GasTarget"]
__version__ = "0.1.0"
