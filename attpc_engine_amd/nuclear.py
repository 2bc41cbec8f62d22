"""Nuclear data carrier used by the engine (host side only).

The reference takes its nuclei from ``spyral_utils.nuclear`` (``NucleusData`` /
``NuclearDataMap``; used at reference ``src/attpc_engine/__init__.py:1-3``,
``kinematics/reaction.py:54,217`` and ``detector/simulator.py:99``).  That package
is a third-party dependency that is *not* part of the reference tree and is not
installable offline, so this module provides an API-compatible carrier: every
engine entry point only duck-types on ``.Z .A .mass .isotopic_symbol``
(SURVEY.md section 8b), so real ``spyral_utils`` objects work unchanged too.

Mass convention (the one spyral_utils uses, pinned by the reference's LISE
known-answer test ``tests/test_kinematics.py:13-36``):

    mass[MeV] = atomic_mass[u] * 931.49410242 - Z * 0.51099895

The built-in table covers the light nuclides the AT-TPC programme uses (atomic
masses in u as tabulated by AME2020).  ``NuclearDataMap.load_ame`` reads a full
``mass_1.mas20``-format file when the user has one.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path

AMU_2_MEV: float = 931.49410242
ELECTRON_MASS: float = 0.51099895  # MeV

ELEMENTS = (
    "n H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn "
    "Ga Ge As Se Br Kr Rb Sr Y Zr Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce "
    "Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir Pt Au Hg Tl Pb Bi Po At Rn "
    "Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr Rf Db Sg Bh Hs Mt Ds Rg Cn Nh Fl "
    "Mc Lv Ts Og"
).split()

# (Z, A): atomic mass in u
_ATOMIC_MASS_U: dict[tuple[int, int], float] = {
    (0, 1): 1.00866491590,
    (1, 1): 1.00782503190, (1, 2): 2.01410177812, (1, 3): 3.01604928132,
    (2, 3): 3.01602932197, (2, 4): 4.00260325413, (2, 5): 5.012057224, (2, 6): 6.018885889,
    (2, 8): 8.033934388,
    (3, 5): 5.012537800, (3, 6): 6.0151228874, (3, 7): 7.0160034366, (3, 8): 8.022486244,
    (3, 9): 9.026790191, (3, 11): 11.04372358,
    (4, 7): 7.016928714, (4, 8): 8.005305102, (4, 9): 9.012183062, (4, 10): 10.013534692,
    (4, 11): 11.021661080, (4, 12): 12.026922082,
    (5, 8): 8.024607315, (5, 9): 9.013329645, (5, 10): 10.012936862, (5, 11): 11.009305166,
    (5, 12): 12.014352638, (5, 13): 13.017779981,
    (6, 9): 9.031037202, (6, 10): 10.016853217, (6, 11): 11.011432597, (6, 12): 12.0,
    (6, 13): 13.00335483534, (6, 14): 14.00324198862, (6, 15): 15.010599256,
    (7, 12): 12.018613180, (7, 13): 13.005738609, (7, 14): 14.00307400425,
    (7, 15): 15.00010889827, (7, 16): 16.006101925,
    (8, 14): 14.008596706, (8, 15): 15.003065636, (8, 16): 15.99491461926,
    (8, 17): 16.99913175595, (8, 18): 17.99915961214,
    (9, 17): 17.002095237, (9, 18): 18.000937324, (9, 19): 18.99840316207,
    (10, 18): 18.005708696, (10, 19): 19.001880906, (10, 20): 19.99244017525,
    (10, 21): 20.993846685, (10, 22): 21.991385114,
    (11, 21): 20.997654459, (11, 22): 21.994437547, (11, 23): 22.98976928195,
    (12, 22): 21.999570597, (12, 23): 22.994123768, (12, 24): 23.985041689,
    (12, 25): 24.985836966, (12, 26): 25.982592972,
    (13, 26): 25.986891876, (13, 27): 26.981538408,
    (14, 28): 27.97692653442, (14, 29): 28.97649466434, (14, 30): 29.973770137,
    (16, 32): 31.97207117354, (18, 36): 35.967545105, (18, 40): 39.96238312204,
    (20, 40): 39.962590850, (20, 48): 47.952522654,
}


@dataclass
class NucleusData:
    """Same fields the engine reads from ``spyral_utils.nuclear.NucleusData``."""

    mass: float = 0.0  # nuclear mass, MeV
    atomic_mass: float = 0.0  # u
    element_symbol: str = ""
    isotopic_symbol: str = ""
    pretty_iso_symbol: str = ""
    Z: int = 0
    A: int = 0

    def __str__(self) -> str:
        return self.isotopic_symbol

    def get_latex_rep(self) -> str:
        return "$^{" + str(self.A) + "}$" + self.element_symbol


def make_nucleus(z: int, a: int, atomic_mass_u: float) -> NucleusData:
    elem = ELEMENTS[z] if 0 <= z < len(ELEMENTS) else f"Z{z}"
    return NucleusData(
        mass=atomic_mass_u * AMU_2_MEV - z * ELECTRON_MASS,
        atomic_mass=atomic_mass_u,
        element_symbol=elem,
        isotopic_symbol=f"{a}{elem}",
        pretty_iso_symbol=f"<sup>{a}</sup>{elem}",
        Z=int(z),
        A=int(a),
    )


class NuclearDataMap:
    """``get_data(z, a) -> NucleusData`` like ``spyral_utils.nuclear.NuclearDataMap``."""

    def __init__(self, ame_path: Path | str | None = None):
        self.map: dict[tuple[int, int], NucleusData] = {
            key: make_nucleus(key[0], key[1], m) for key, m in _ATOMIC_MASS_U.items()
        }
        if ame_path is not None:
            self.load_ame(ame_path)

    def get_data(self, z: int, a: int) -> NucleusData:
        key = (int(z), int(a))
        if key not in self.map:
            raise KeyError(
                f"Nucleus Z={z} A={a} is not in the built-in light-nuclide table; "
                "load a full AME file with NuclearDataMap.load_ame(path) or add it with add()."
            )
        return self.map[key]

    def add(self, z: int, a: int, atomic_mass_u: float) -> NucleusData:
        self.map[(int(z), int(a))] = make_nucleus(z, a, atomic_mass_u)
        return self.map[(int(z), int(a))]

    def load_ame(self, path: Path | str) -> int:
        """Parse an AME ``mass_1.mas20``-style fixed-width table. Returns #nuclides read."""
        count = 0
        with open(path, "r") as handle:
            for line in handle:
                if len(line) < 118 or not line[1:4].strip().lstrip("-").isdigit():
                    continue
                try:
                    z = int(line[9:14])
                    a = int(line[14:19])
                    mass_u = float(line[106:109]) + float(
                        line[110:123].replace("#", "").replace(" ", "")
                    ) * 1.0e-6
                except ValueError:
                    continue
                self.add(z, a, mass_u)
                count += 1
        return count
