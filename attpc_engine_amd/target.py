"""Gas target with a built-in analytic stopping-power model (host side only).

The reference's target is ``spyral_utils.nuclear.target.GasTarget`` whose numbers
come from ``pycatima`` (call sites: reference ``detector/solver.py:64-66`` --
``get_dedx(nucleus, KE) [MeV/(g/cm^2)]`` and ``.density [g/cm^3]`` -- and
``kinematics/pipeline.py:256-264`` -- ``get_energy_loss(nucleus, KE, distances_m)``).
Neither package is part of the reference tree nor available offline, so the
engine treats the target as a *duck-typed input*: whatever object the user passes
is sampled into look-up tables at configure time (``attpc_engine_amd.detector.
luts``).  A real spyral_utils ``GasTarget`` therefore works unchanged.

This class gives the same interface with a self-contained model so the package is
usable (and benchmarkable) without catima: Bethe electronic stopping with an
effective projectile charge, Lindhard-Scharff velocity-proportional stopping at
low energy (harmonic interpolation between the two), plus ZBL universal nuclear
stopping; Bragg additivity for compounds.  Stopping powers are "parity unpinned"
with respect to catima -- no reference test pins one.
"""
from __future__ import annotations

import math
import numpy as np

GAS_CONSTANT = 8.31446261815324  # J / (mol K)
ROOM_TEMPERATURE = 293.15  # K
TORR_2_PASCAL = 133.32236842105263
AVOGADRO = 6.02214076e23
ELECTRON_MASS_MEV = 0.51099895
BETHE_K = 0.307075  # MeV cm^2 / mol
AMU_MEV = 931.49410242

# mean excitation energies (eV), ICRU-37/49 values for elemental gases/solids
_MEAN_EXCITATION_EV = {
    1: 19.2, 2: 41.8, 3: 40.0, 4: 63.7, 5: 76.0, 6: 81.0, 7: 82.0, 8: 95.0, 9: 115.0,
    10: 137.0, 13: 166.0, 14: 173.0, 18: 188.0, 36: 352.0, 54: 482.0,
}


def _mean_excitation_ev(z: int) -> float:
    if z in _MEAN_EXCITATION_EV:
        return _MEAN_EXCITATION_EV[z]
    return 9.76 * z + 58.8 * z ** (-0.19) if z >= 13 else 12.0 * z + 7.0


def _elemental_stopping(zp: int, mp_mev: float, ke_mev: float, zt: int, at_u: float) -> float:
    """Total (electronic + nuclear) stopping of (zp, mp) at ke in element zt. MeV cm^2/g."""
    if ke_mev <= 0.0 or zp == 0:
        return 0.0
    gamma = 1.0 + ke_mev / mp_mev
    beta2 = 1.0 - 1.0 / (gamma * gamma)
    beta = math.sqrt(beta2)
    mp_u = mp_mev / AMU_MEV
    # --- electronic, high energy: Bethe with effective charge, log regularised ---
    zeff = float(zp)
    if zp >= 2:
        zeff = zp * (1.0 - math.exp(-125.0 * beta * zp ** (-2.0 / 3.0)))
    i_mev = _mean_excitation_ev(zt) * 1.0e-6
    arg = 2.0 * ELECTRON_MASS_MEV * beta2 * gamma * gamma / i_mev
    s_bethe = BETHE_K * zeff * zeff * (zt / at_u) / beta2 * (math.log1p(arg) - beta2)
    # --- electronic, low energy: Lindhard-Scharff, eV cm^2 / 1e15 atoms ---
    e_kev_per_u = ke_mev * 1.0e3 / mp_u
    k_ls = 1.212 * zp ** (7.0 / 6.0) * zt / ((zp ** (2.0 / 3.0) + zt ** (2.0 / 3.0)) ** 0.75)
    s_ls = k_ls * math.sqrt(e_kev_per_u) * (AVOGADRO * 1.0e-21 / at_u)  # -> MeV cm^2/g
    s_elec = s_ls * s_bethe / (s_ls + s_bethe) if (s_ls > 0.0 and s_bethe > 0.0) else 0.0
    # --- nuclear: ZBL universal ---
    e_kev = ke_mev * 1.0e3
    zfac = zp**0.23 + zt**0.23
    eps = 32.53 * at_u * e_kev / (zp * zt * (mp_u + at_u) * zfac)
    if eps <= 30.0:
        sn_red = math.log1p(1.1383 * eps) / (
            2.0 * (eps + 0.01321 * eps**0.21226 + 0.19593 * math.sqrt(eps))
        )
    else:
        sn_red = math.log(eps) / (2.0 * eps)
    s_nuc = 8.462 * zp * zt * mp_u * sn_red / ((mp_u + at_u) * zfac)  # eV cm^2/1e15 atoms
    s_nuc *= AVOGADRO * 1.0e-21 / at_u
    return s_elec + s_nuc


_ELOSS_MEMO: dict = {}  # GasTarget.get_energy_loss results of this process (a pure function of its arguments)


class GasTarget:
    """API-compatible stand-in for ``spyral_utils.nuclear.target.GasTarget``.

    Parameters
    ----------
    compound: list[tuple[int, int, int]]
        ``(Z, A, stoichiometry)`` per element, e.g. ``[(1, 2, 2)]`` for D2.
    pressure: float
        Gas pressure in Torr.
    nuclear_map:
        Object with ``get_data(z, a)`` returning something with ``.atomic_mass`` (u).
    """

    def __init__(self, compound, pressure: float, nuclear_map):
        self.compound = [(int(z), int(a), int(s)) for (z, a, s) in compound]
        self.pressure = float(pressure)
        self.compound_key = tuple(self.compound)
        self._elements = []
        molar_mass = 0.0
        for z, a, s in self.compound:
            at_u = float(nuclear_map.get_data(z, a).atomic_mass)
            self._elements.append((z, at_u, s))
            molar_mass += at_u * s
        self.molar_mass = molar_mass
        # ideal gas at room temperature, g/cm^3
        self.density = (
            molar_mass * self.pressure * TORR_2_PASCAL / (GAS_CONSTANT * ROOM_TEMPERATURE) * 1.0e-6
        )

    def get_dedx(self, projectile_data, projectile_energy: float) -> float:
        """Stopping power in MeV/(g/cm^2) for kinetic energy in MeV."""
        total = 0.0
        for z, at_u, s in self._elements:
            weight = at_u * s / self.molar_mass
            total += weight * _elemental_stopping(
                int(projectile_data.Z), float(projectile_data.mass), float(projectile_energy), z, at_u
            )
        return total

    def get_energy_loss(self, projectile_data, projectile_energy: float, distances) -> np.ndarray:
        """Energy lost (MeV) after each path length in ``distances`` (m); RK4 in path length."""
        distances = np.atleast_1d(np.asarray(distances, dtype=float))
        # a pure function of (gas, projectile, energy, distances), and a slow one in pure Python (one RK4 integration per
        # distance: seconds for the 2049-node table a pipeline configures): remembered per process
        memo_key = (self.compound_key, self.pressure, int(projectile_data.Z), float(projectile_data.mass),
                    float(projectile_energy), distances.tobytes())
        cached = _ELOSS_MEMO.get(memo_key)
        if cached is not None:
            return cached.copy()
        out = np.zeros_like(distances)
        scale = self.density * 100.0  # MeV/(g/cm^2) -> MeV/m

        def f(e: float) -> float:
            return -self.get_dedx(projectile_data, e) * scale if e > 0.0 else 0.0

        for i, dist in enumerate(distances):
            if dist <= 0.0:
                continue
            n_steps = max(16, int(dist / 1.0e-3))
            h = dist / n_steps
            e = float(projectile_energy)
            for _ in range(n_steps):
                k1 = f(e)
                k2 = f(e + 0.5 * h * k1)
                k3 = f(e + 0.5 * h * k2)
                k4 = f(e + h * k3)
                e += h / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
                if e <= 0.0:
                    e = 0.0
                    break
            out[i] = projectile_energy - e
        if len(_ELOSS_MEMO) >= 64:
            _ELOSS_MEMO.clear()
        _ELOSS_MEMO[memo_key] = out.copy()
        return out

    def get_number_density(self) -> float:
        """Molecules per cm^3."""
        return self.density / self.molar_mass * AVOGADRO
