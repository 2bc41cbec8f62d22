"""Kinematics phase-space sampling: same public names as the reference package
(reference ``kinematics/__init__.py:20-33``), device backed."""
from . import angle as _angle
from . import excitation as _excitation
from . import pipeline as _pipeline
from . import reaction as _reaction

_EXPORTS = {
    _pipeline: ("KinematicsPipeline", "run_kinematics_pipeline", "KinematicsTargetMaterial", "PipelineError",
                "Sample"),
    _excitation: ("ExcitationDistribution", "ExcitationGaussian", "ExcitationUniform", "ExcitationBreitWigner"),
    _angle: ("PolarDistribution", "PolarArbitrary", "PolarUniform"),
    _reaction: ("Reaction", "Decay", "FourVector"),
}
__all__ = []
for _module, _names in _EXPORTS.items():
    for _name in _names:
        globals()[_name] = getattr(_module, _name)
        __all__.append(_name)
del _module, _names, _name
