"""Centre-of-mass polar-angle distributions (reference ``kinematics/angle.py``)."""
from __future__ import annotations

from typing import Protocol

import numpy as np
from numpy.random import Generator

from .._abi import POLAR_ARBITRARY, POLAR_UNIFORM


class PolarDistribution(Protocol):
    """Anything with ``sample(rng) -> float`` (radians in [0, pi]); reference angle.py:6-32."""

    def sample(self, rng: Generator) -> float: ...


class PolarUniform:
    """Uniform in cos(theta) between two polar angles (reference angle.py:35-80).
    cos() reverses the ordering, hence the swapped attribute names."""

    def __init__(self, angle_min: float, angle_max: float):
        self.cos_angle_min = np.cos(angle_max)
        self.cos_angle_max = np.cos(angle_min)

    def sample(self, rng: Generator) -> float:
        return np.arccos(rng.uniform(self.cos_angle_min, self.cos_angle_max))

    def device_desc(self) -> dict:
        return {
            "kind": POLAR_UNIFORM,
            "cos_min": float(self.cos_angle_min),
            "cos_max": float(self.cos_angle_max),
        }


class PolarArbitrary:
    """Binned angular distribution: pick a bin with probability ``probabilities`` and smear
    uniformly inside ``angle_bin_width`` (reference angle.py:83-152).  Only a sum > 1 is
    rejected, exactly as the reference does (:128-131)."""

    def __init__(self, angles: np.ndarray, probabilities: np.ndarray, angle_bin_width: float):
        total = np.sum(probabilities)
        if total > 1.0:
            raise ValueError(
                "The sum of the probabilities passed to PolarArbitrary should be 1.0. "
                f"Yours sum to {total}"
            )
        self.angle_width = angle_bin_width
        self.probs = probabilities
        self.angles = angles

    def sample(self, rng: Generator) -> float:
        return rng.choice(self.angles, p=self.probs) + rng.uniform(0.0, 1.0) * self.angle_width

    def device_desc(self) -> dict:
        # numpy Generator.choice: cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, 'right')
        cdf = np.cumsum(np.asarray(self.probs, dtype=np.float64))
        cdf /= cdf[-1]
        return {
            "kind": POLAR_ARBITRARY,
            "bin_width": float(self.angle_width),
            "angles": np.ascontiguousarray(self.angles, dtype=np.float64),
            "cdf": np.ascontiguousarray(cdf),
        }
