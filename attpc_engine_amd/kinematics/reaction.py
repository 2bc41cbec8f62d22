"""Two-body reaction / decay steps (reference ``kinematics/reaction.py``).

``Reaction`` and ``Decay`` carry the nuclei of one step.  Their ``calculate`` /
``is_excitation_allowed`` methods evaluate a single parameter set through the same HIP
kernel the pipeline uses (``attpc_kin_calculate``), so the known-answer tests of the
reference (``tests/test_kinematics.py:13-36``) exercise the device arithmetic.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from .. import _abi


@dataclass
class FourVector:
    """Momentum 4-vector (MeV); the fields the reference reads from ``vector`` objects."""

    px: float
    py: float
    pz: float
    E: float

    @property
    def M(self) -> float:
        m2 = self.E * self.E - (self.px * self.px + self.py * self.py + self.pz * self.pz)
        return math.sqrt(m2) if m2 >= 0.0 else -math.sqrt(-m2)

    def as_array(self) -> np.ndarray:
        return np.array([self.px, self.py, self.pz, self.E])


def _nuclear_map():
    from .. import nuclear_map

    return nuclear_map


class _Chain:
    """Minimal pipeline-shaped object for descriptor building."""

    def __init__(self, reaction, decays):
        self.reaction = reaction
        self.decays = decays
        self.event_sample_limit = 1
        self.beam_energy = 0.0
        self.target_material = None
        self.excitations = []
        self.polar_dists = []


def device_calculate(reaction, decays, beam_energy, ex, polar, azim, ctx=None):
    """Evaluate n parameter sets on the device -> (p4 [n, rows, 4], status [n])."""
    from ._device import build_kin_desc

    ctx = ctx or _abi.default_context()
    desc, _keep = build_kin_desc(_Chain(reaction, decays), deterministic_only=True)
    ctx.check(ctx.lib.attpc_kin_configure(ctx.handle, desc), "attpc_kin_configure")
    n_steps = 1 + len(decays)
    n_rows = 4 + 2 * len(decays)
    beam = np.ascontiguousarray(np.atleast_1d(beam_energy), dtype=np.float64)
    n = beam.size
    ex = np.ascontiguousarray(np.asarray(ex, dtype=np.float64).reshape(n, n_steps))
    polar = np.ascontiguousarray(np.asarray(polar, dtype=np.float64).reshape(n, n_steps))
    azim = np.ascontiguousarray(np.asarray(azim, dtype=np.float64).reshape(n, n_steps))
    p4 = np.empty((n, n_rows, 4), dtype=np.float64)
    status = np.empty(n, dtype=np.int32)
    ctx.check(
        ctx.lib.attpc_kin_calculate(
            ctx.handle, n, _abi.dptr(beam), _abi.dptr(ex), _abi.dptr(polar), _abi.dptr(azim),
            _abi.dptr(p4), _abi.iptr(status, _abi.C.c_int32),
        ),
        "attpc_kin_calculate",
    )
    return p4, status


class Reaction:
    """target(projectile, ejectile)residual; the residual follows from Z/A conservation
    (reference reaction.py:35-58)."""

    def __init__(self, target, projectile, ejectile):
        self.projectile = projectile
        self.target = target
        self.ejectile = ejectile
        resid_z = projectile.Z + target.Z - ejectile.Z
        resid_a = projectile.A + target.A - ejectile.A
        if resid_z < 0:
            raise ValueError("Reaction calculated a residual Z (proton number) < 0, illegal reaction!")
        if resid_a < 0:
            raise ValueError("Reaction calculated a residual A (mass number) < 0, illegal reaction!")
        self.residual = _nuclear_map().get_data(resid_z, resid_a)
        self.reaction_symbol = f"{self.target}({self.projectile},{self.ejectile}){self.residual}"

    def __str__(self) -> str:
        return self.reaction_symbol

    def is_excitation_allowed(self, projectile_energy: float, residual_excitation: float) -> bool:
        """True iff m_ejectile + m_residual + Ex < E_cm (reference reaction.py:70-101)."""
        _, status = device_calculate(self, [], projectile_energy, [residual_excitation], [0.0], [0.0])
        return bool(status[0] not in (1, -2))

    def calculate(
        self,
        projectile_energy: float,
        ejectile_polar: float,
        ejectile_azimuthal: float,
        residual_excitation: float,
    ) -> list[FourVector]:
        """[target, projectile, ejectile, residual] lab 4-vectors (reference reaction.py:103-178)."""
        p4, status = device_calculate(
            self, [], projectile_energy, [residual_excitation], [ejectile_polar], [ejectile_azimuthal]
        )
        if status[0] in (-1, -2):
            raise ValueError("Beam energy below kinematic threshold!")
        return [FourVector(*row) for row in p4[0]]


class Decay:
    """parent -> residual_1 + residual_2 (reference reaction.py:203-218)."""

    def __init__(self, parent, residual_1):
        self.parent = parent
        self.residual_1 = residual_1
        resid_2_z = parent.Z - residual_1.Z
        resid_2_a = parent.A - residual_1.A
        if resid_2_z < 0:
            raise ValueError("Decay calculated a residual2 Z (proton number) < 0, illegal decay!")
        if resid_2_a < 0:
            raise ValueError("Decay calculated a residual2 A (mass number) < 0, illegal decay!")
        self.residual_2 = _nuclear_map().get_data(resid_2_z, resid_2_a)
        self.decay_symbol = f"{self.parent}->{self.residual_1}+{self.residual_2}"

    def __str__(self) -> str:
        return self.decay_symbol

    def is_excitation_allowed(self, parent_vector, residual_2_excitation: float) -> bool:
        """True iff M(parent 4-vector) - (m1 + m2 + Ex) > 0 (reference reaction.py:230-250)."""
        from ._device import decay_only_calculate

        _, ok = decay_only_calculate(self, parent_vector, 0.0, 0.0, residual_2_excitation)
        return ok

    def calculate(
        self,
        parent_vector,
        residual_1_polar: float,
        residual_1_azimuthal: float,
        residual_2_excitation: float,
    ) -> list[FourVector]:
        """[parent, residual_1, residual_2] lab 4-vectors (reference reaction.py:252-303).

        Evaluated on the device as a one-decay chain whose "reaction residual" row is the
        given parent 4-vector (``attpc_kin_calculate`` with a parent override)."""
        from ._device import decay_only_calculate

        rows, ok = decay_only_calculate(
            self, parent_vector, residual_1_polar, residual_1_azimuthal, residual_2_excitation
        )
        if not ok:
            raise ValueError("Parent doesn't have enough energy to decay!")
        parent = FourVector(parent_vector.px, parent_vector.py, parent_vector.pz, parent_vector.E)
        return [parent, FourVector(*rows[0]), FourVector(*rows[1])]
