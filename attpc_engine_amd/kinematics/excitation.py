"""Excitation-energy distributions (reference ``kinematics/excitation.py``).

Each class keeps the reference's constructor, attributes and ``sample(rng)`` host method
(a one-line numpy draw, used when a user mixes in custom Python distributions) and adds
``device_desc()``: the parameters the HIP sampler consumes, so that a pipeline made only
of these built-in types is sampled entirely on the GPU.
"""
from __future__ import annotations

from typing import Protocol

import numpy as np
from numpy.random import Generator

from .._abi import EX_GAUSSIAN, EX_TABLE, EX_UNIFORM


class ExcitationDistribution(Protocol):
    """Anything with ``sample(rng) -> float`` (MeV); reference excitation.py:6-29."""

    def sample(self, rng: Generator) -> float: ...


class ExcitationGaussian:
    """Gaussian state: ``centroid`` and FWHM ``width`` in MeV; sigma = FWHM / 2.355
    (reference excitation.py:32-80)."""

    def __init__(self, centroid: float = 0.0, width: float = 0.0):
        self.centroid = centroid
        self.width = width
        self.sigma = self.width / 2.355

    def sample(self, rng: Generator) -> float:
        return rng.normal(self.centroid, self.sigma)

    def device_desc(self) -> dict:
        return {"kind": EX_GAUSSIAN, "p0": float(self.centroid), "p1": float(self.sigma)}


class ExcitationUniform:
    """Flat between ``min_value`` and ``max_value`` MeV (reference excitation.py:83-128)."""

    def __init__(self, min_value: float = 0.0, max_value: float = 0.0):
        self.min_value = min_value
        self.max_value = max_value

    def sample(self, rng: Generator) -> float:
        return rng.uniform(self.min_value, self.max_value)

    def device_desc(self) -> dict:
        return {"kind": EX_UNIFORM, "p0": float(self.min_value), "p1": float(self.max_value)}


class ExcitationBreitWigner:
    """Relativistic Breit-Wigner in the total energy (reference excitation.py:131-188):
    ``E_tot ~ rel_breitwigner(rho=(rest_mass+centroid)/width, scale=width)``,
    ``Ex = E_tot - rest_mass``.

    scipy draws this by numerically inverting the closed-form CDF for every sample
    (~2 ms each).  For the device the same CDF is tabulated once on ``table_nodes``
    abscissae placed at the quantiles of the matching Lorentzian (dense where the density
    is) and inverted by binary search + linear interpolation.
    """

    def __init__(self, rest_mass: float, centroid: float, width: float, table_nodes: int = 16385):
        self.rest_mass = rest_mass
        self.centroid = centroid
        self.width = width
        self.table_nodes = int(table_nodes)
        self._table: tuple[np.ndarray, np.ndarray] | None = None

    def sample(self, rng: Generator) -> float:
        from scipy.stats import rel_breitwigner

        rho = (self.rest_mass + self.centroid) / self.width
        return rel_breitwigner.rvs(rho, scale=self.width, random_state=rng) - self.rest_mass

    def cdf_table(self) -> tuple[np.ndarray, np.ndarray]:
        if self._table is None:
            from scipy.stats import rel_breitwigner

            rho = (self.rest_mass + self.centroid) / self.width
            v = (np.arange(self.table_nodes) + 0.5) / self.table_nodes
            eps = 1.0e-7
            v = eps + (1.0 - 2.0 * eps) * v
            x = rho + 0.5 * np.tan(np.pi * (v - 0.5))  # Lorentzian(HWHM 1/2) quantiles around rho
            x = np.unique(np.clip(x, 1.0e-9, None))
            cdf = rel_breitwigner.cdf(x, rho)
            cdf = np.maximum.accumulate(cdf)
            cdf = (cdf - cdf[0]) / (cdf[-1] - cdf[0])
            self._table = (np.ascontiguousarray(x * self.width), np.ascontiguousarray(cdf))
        return self._table

    def device_desc(self) -> dict:
        x, cdf = self.cdf_table()
        return {"kind": EX_TABLE, "p0": float(self.rest_mass), "table_x": x, "table_cdf": cdf}
