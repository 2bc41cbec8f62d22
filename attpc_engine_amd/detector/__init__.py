"""Detector effects: same public names as the reference package (reference
``detector/__init__.py:13-21``) plus the batch entry points, device backed."""
from . import parameters as _parameters
from . import simulator as _simulator
from . import writer as _writer

_EXPORTS = {
    _parameters: ("Config", "DetectorParams", "ElectronicsParams", "PadParams"),
    _simulator: ("run_simulation", "simulate", "simulate_batch"),
    _writer: ("SimulationWriter", "SpyralWriter"),
}
__all__ = []
for _module, _names in _EXPORTS.items():
    for _name in _names:
        globals()[_name] = getattr(_module, _name)
        __all__.append(_name)
del _module, _names, _name
