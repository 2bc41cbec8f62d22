"""Pads in the beam region that never record charge (data of reference
``detector/beam_pads.py:11-134``: 122 pad ids, stored here as inclusive runs)."""
import numpy as np

_BEAM_PAD_RUNS = (
    (134, 164), (166, 166), (435, 457), (459, 459), (733, 733), (735, 735), (738, 738),
    (740, 741), (5254, 5284), (5286, 5286), (5555, 5577), (5579, 5579), (5853, 5853),
    (5855, 5855), (5858, 5858), (5860, 5861),
)
BEAM_PADS: list[int] = [p for lo, hi in _BEAM_PAD_RUNS for p in range(lo, hi + 1)]
BEAM_PADS_ARRAY = np.array(BEAM_PADS)
