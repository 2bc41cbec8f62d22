"""Physical constants baked into the kernels (reference ``detector/constants.py:20-35``;
the reference reads them from scipy's CODATA table -- the exact doubles are repeated in
``csrc/common.hpp`` and checked against scipy in the CPU tests)."""
from scipy.constants import elementary_charge, physical_constants, speed_of_light

NUM_TB: int = 512
MEV_2_JOULE: float = physical_constants["electron volt-joule relationship"][0] * 1.0e6
MEV_2_KG: float = physical_constants["electron volt-kilogram relationship"][0] * 1.0e6
C: float = speed_of_light
E_CHARGE: float = elementary_charge
