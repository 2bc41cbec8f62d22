"""Concrete synthetic workloads for the BASELINE.json configs.

Each builder returns ``(pipeline, config, indices)`` made of ordinary API objects, so the
benchmark, the smoke test and the parity tests all drive the engine the way a user script
does.  BASELINE.json leaves beam energies and gas pressures open; the values fixed here
are recorded in every bench JSON line through ``describe()``.
"""
from __future__ import annotations

import numpy as np

from . import GasTarget, nuclear_map
from .detector import Config, DetectorParams, ElectronicsParams, PadParams
from .detector.simulator import default_indices
from .kinematics import (
    Decay, ExcitationGaussian, KinematicsPipeline, KinematicsTargetMaterial, PolarUniform, Reaction,
)


def detector_config(gas, diffusion: float = 0.277, path_step: float = 0.0) -> Config:
    """Detector/electronics defaults of the reference's own test (tests/test_detector.py:15-33)."""
    det = DetectorParams(length=1.0, efield=45000.0, bfield=2.85, mpgd_gain=175000, gas_target=gas,
                         diffusion=diffusion, fano_factor=0.2, w_value=34.0, path_step=path_step)
    elec = ElectronicsParams(clock_freq=6.25, amp_gain=900, shaping_time=1000, micromegas_edge=10,
                             windows_edge=560, adc_threshold=40)
    return Config(det, elec, PadParams())


def c12pp(seed: int = 1, **kw):
    """configs[0]: 12C(p,p) elastic, 10 MeV protons, kinematics only (no target material)."""
    nm = nuclear_map
    pipeline = KinematicsPipeline(
        [Reaction(target=nm.get_data(6, 12), projectile=nm.get_data(1, 1), ejectile=nm.get_data(1, 1))],
        [ExcitationGaussian(0.0, 0.0)], [PolarUniform(0.0, np.pi)], beam_energy=10.0, seed=seed, **kw)
    return pipeline, None, default_indices(4)


def be10dp(seed: int = 2, **kw):
    """configs[1]: 10Be(d,p)11Be in inverse kinematics: 96 MeV 10Be on D2 at 600 Torr,
    11Be first excited state (0.32 MeV), full detector, indices [2, 3]."""
    nm = nuclear_map
    gas = GasTarget([(1, 2, 2)], 600.0, nm)
    pipeline = KinematicsPipeline(
        [Reaction(target=nm.get_data(1, 2), projectile=nm.get_data(4, 10), ejectile=nm.get_data(1, 1))],
        [ExcitationGaussian(0.32, 0.0)], [PolarUniform(0.0, np.pi)], beam_energy=96.0,
        target_material=KinematicsTargetMaterial(gas, (0.0, 1.0), 0.007), seed=seed, **kw)
    return pipeline, detector_config(gas), default_indices(4)


def o16aa(seed: int = 3, **kw):
    """configs[2] (headline): 16O(a,a')16O* -> a + 12C in inverse kinematics: 160 MeV 16O on
    He at 600 Torr, 16O* = Gaussian(9.585 MeV, FWHM 0.42), 12C ground state, indices [2,4,5]."""
    nm = nuclear_map
    gas = GasTarget([(2, 4, 1)], 600.0, nm)
    pipeline = KinematicsPipeline(
        [Reaction(target=nm.get_data(2, 4), projectile=nm.get_data(8, 16), ejectile=nm.get_data(2, 4)),
         Decay(parent=nm.get_data(8, 16), residual_1=nm.get_data(2, 4))],
        [ExcitationGaussian(9.585, 0.42), ExcitationGaussian(0.0, 0.0)],
        [PolarUniform(0.0, np.pi), PolarUniform(0.0, np.pi)], beam_energy=160.0,
        target_material=KinematicsTargetMaterial(gas, (0.0, 1.0), 0.007), seed=seed, **kw)
    return pipeline, detector_config(gas), default_indices(6)


def b10chain(seed: int = 5, diffusion: float = 2.77, path_step: float = 1.0e-4, **kw):
    """configs[4] (stress): 3-step chain 10B(3He,a)9B -> a + 5Li -> a + p, 24 MeV 3He, He 600 Torr, 0.1 mm dE/dx path step, 10x diffusion, indices [2,4,6,7]
    (the chain of the reference's test_pipeline, tests/test_kinematics.py:42-69; ``path_step`` and the
    diffusion are the stress extensions BASELINE.json names: a track sample every 0.1 mm of arc
    length instead of the reference's 1e-10 s grid, and 10x the default diffusion coefficient to
    maximise scatter contention)."""
    nm = nuclear_map
    gas = GasTarget([(2, 4, 1)], 600.0, nm)
    pipeline = KinematicsPipeline(
        [Reaction(target=nm.get_data(5, 10), projectile=nm.get_data(2, 3), ejectile=nm.get_data(2, 4)),
         Decay(parent=nm.get_data(5, 9), residual_1=nm.get_data(2, 4)),
         Decay(parent=nm.get_data(3, 5), residual_1=nm.get_data(2, 4))],
        [ExcitationGaussian(16.8, 0.2), ExcitationGaussian(0.0, 1.25), ExcitationGaussian(0.0, 0.0)],
        [PolarUniform(0.0, np.pi)] * 3, beam_energy=24.0,
        target_material=KinematicsTargetMaterial(gas, (0.0, 1.0), 0.007), seed=seed, **kw)
    return pipeline, detector_config(gas, diffusion=diffusion, path_step=path_step), default_indices(8)


WORKLOADS = {"c12pp": c12pp, "be10dp": be10dp, "o16aa": o16aa, "b10chain": b10chain}


def describe(name: str) -> str:
    return (WORKLOADS[name].__doc__ or name).split("\n")[0].strip()
