"""Kinematics phase-space sampling (reference ``kinematics/__init__.py:20-33``)."""
from .pipeline import (
    KinematicsPipeline,
    run_kinematics_pipeline,
    KinematicsTargetMaterial,
    PipelineError,
    Sample,
)
from .excitation import (
    ExcitationDistribution,
    ExcitationGaussian,
    ExcitationUniform,
    ExcitationBreitWigner,
)
from .angle import PolarDistribution, PolarUniform, PolarArbitrary
from .reaction import Reaction, Decay, FourVector

__all__ = [
    "KinematicsPipeline",
    "run_kinematics_pipeline",
    "KinematicsTargetMaterial",
    "ExcitationDistribution",
    "ExcitationGaussian",
    "ExcitationUniform",
    "ExcitationBreitWigner",
    "PolarDistribution",
    "PolarArbitrary",
    "PolarUniform",
    "Reaction",
    "Decay",
]
