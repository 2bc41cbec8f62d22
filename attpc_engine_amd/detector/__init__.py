"""Detector effects (reference ``detector/__init__.py:13-21``)."""
from .parameters import Config, DetectorParams, ElectronicsParams, PadParams
from .simulator import run_simulation, simulate, simulate_batch
from .writer import SimulationWriter, SpyralWriter

__all__ = [
    "run_simulation",
    "DetectorParams",
    "ElectronicsParams",
    "PadParams",
    "Config",
    "SpyralWriter",
    "SimulationWriter",
]
