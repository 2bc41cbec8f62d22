"""Point-cloud writers (reference ``detector/writer.py``).

``SimulationWriter`` is the protocol ``run_simulation`` drives (write once per non-empty
event in event order, then close).  ``SpyralWriter`` produces the Spyral layout; the
per-point response scaling / row conversion runs on the device (``attpc_spyral_rows``; the
fused path ``Engine.run_spyral`` also thresholds and z-sorts there).  h5py is imported lazily;
without it the same datasets go to one ``.npz`` per run file, after a warning (the HDF5 layout is
"parity unpinned" in this container; tests drive it through a stand-in module).
"""
from __future__ import annotations

from pathlib import Path
from typing import Protocol

import numpy as np

from .. import _abi
from .parameters import Config
from .response import get_response


class SimulationWriter(Protocol):
    """write(data [P,3], labels [P], config, event_number); get_directory_name(); close()
    (reference writer.py:12-58)."""

    def write(self, data: np.ndarray, labels: np.ndarray, config: Config, event_number: int) -> None: ...

    def get_directory_name(self) -> Path: ...

    def close(self) -> None: ...


def convert_to_spyral(points: np.ndarray, window_edge: int, mm_edge: int, length: float,
                      response: np.ndarray, pad_centers: np.ndarray, pad_sizes: np.ndarray,
                      ctx: _abi.Context | None = None) -> np.ndarray:
    """[P,3] (pad, tb, electrons) -> [P,8] (x mm, y mm, z mm, amplitude, integral, pad, tb,
    pad scale) on the device (reference writer.py:61-112)."""
    ctx = ctx or _abi.default_context()
    points = np.ascontiguousarray(points, dtype=np.float64)
    response = np.ascontiguousarray(response, dtype=np.float64)
    centers = np.ascontiguousarray(pad_centers, dtype=np.float64)
    sizes = np.ascontiguousarray(pad_sizes, dtype=np.float64)
    rows = np.empty((len(points), 8), dtype=np.float64)
    if len(points) == 0:
        return rows
    ctx.check(
        ctx.lib.attpc_spyral_rows(
            ctx.handle, len(points), _abi.dptr(points), _abi.dptr(response), _abi.dptr(centers),
            _abi.dptr(sizes), len(sizes), int(window_edge), int(mm_edge), float(length),
            _abi.dptr(rows),
        ),
        "attpc_spyral_rows",
    )
    return rows


class _NpzRunFile:
    """Fallback container when h5py is absent: same dataset names/attrs, one npz per run."""

    def __init__(self, path: Path):
        self.path = path.with_suffix(".npz")
        self.arrays: dict[str, np.ndarray] = {}

    def create_dataset(self, name: str, data: np.ndarray, attrs: dict | None = None) -> None:
        self.arrays[f"cloud/{name}"] = np.asarray(data)
        for key, value in (attrs or {}).items():
            self.arrays[f"cloud/{name}@{key}"] = np.asarray(value)

    def set_attr(self, key: str, value) -> None:
        self.arrays[f"cloud@{key}"] = np.asarray(value)

    def close(self) -> None:
        np.savez_compressed(self.path, **self.arrays)


class _H5RunFile:
    def __init__(self, path: Path, h5):
        self.file = h5.File(path, "w")
        self.group = self.file.create_group("cloud")

    def create_dataset(self, name: str, data: np.ndarray, attrs: dict | None = None) -> None:
        dset = self.group.create_dataset(name, data=data)
        for key, value in (attrs or {}).items():
            dset.attrs[key] = value

    def set_attr(self, key: str, value) -> None:
        self.group.attrs[key] = value

    def close(self) -> None:
        self.file.close()


class SpyralWriter:
    """Spyral-format output split into files of ``max_events_per_file`` events
    (reference writer.py:115-281): ``run_%04d.h5`` / group ``cloud`` / ``cloud_{event}``
    [P,8] + ``labels_{event}``, attrs orig_run, orig_event, ic_* = -1, and min_event /
    max_event on the group."""

    def __init__(self, directory_path: Path, config: Config, max_events_per_file: int = 5_000,
                 first_run_number: int = 0, npz_fallback: bool = True):
        self.directory_path = Path(directory_path)
        self.npz_fallback = npz_fallback  # without h5py: warn and write .npz (True) or raise (False)
        self.response = get_response(config).copy()
        self.max_events_per_file = max_events_per_file
        self.run_number = first_run_number
        self.starting_event = 0
        self.last_event = 0
        self.events_written = 0
        self.file = self._open(self.run_number)

    def _open(self, run_number: int):
        from ..io import hdf5_or_fallback

        path = self.directory_path / f"run_{run_number:04d}.h5"
        h5py = hdf5_or_fallback(path, self.npz_fallback)
        return _H5RunFile(path, h5py) if h5py is not None else _NpzRunFile(path)

    def create_next_file(self) -> None:
        self.run_number += 1
        self.file = self._open(self.run_number)

    def write(self, data: np.ndarray, labels: np.ndarray, config: Config, event_number: int) -> None:
        """[P,3] cloud -> Spyral rows (device), ADC threshold, z-sort, datasets (writer.py:194-255)."""
        if config.pad_centers is None:
            raise ValueError("Pad centers are not assigned at write!")
        rows = convert_to_spyral(
            data, config.elec_params.windows_edge, config.elec_params.micromegas_edge,
            config.det_params.length, self.response, config.pad_centers, config.pad_sizes,
        )
        keep = rows[:, 3] > config.elec_params.adc_threshold  # writer.py:232-234
        self.write_rows(rows[keep], labels[keep], event_number)

    def write_rows(self, rows: np.ndarray, labels: np.ndarray, event_number: int, presorted: bool = False) -> None:
        """Already converted and thresholded rows [P',8]: z-sort (writer.py:236-238) unless the rows
        come ``presorted`` from the device (``Engine.run_spyral``), file roll-over (:214-218), datasets
        and attributes (:240-251)."""
        if self.events_written == self.max_events_per_file:
            self.close()
            self.create_next_file()
            self.starting_event = event_number
            self.events_written = 0
        if not presorted:
            order = np.argsort(rows[:, 2])
            rows, labels = rows[order], labels[order]
        self.file.create_dataset(
            f"cloud_{event_number}", rows,
            {"orig_run": self.run_number, "orig_event": event_number, "ic_amplitude": -1.0,
             "ic_multiplicity": -1.0, "ic_integral": -1.0, "ic_centroid": -1.0},
        )
        self.file.create_dataset(f"labels_{event_number}", labels)
        self.last_event = event_number
        self.events_written += 1

    def set_number_of_events(self) -> None:
        self.file.set_attr("min_event", self.starting_event)
        self.file.set_attr("max_event", self.last_event)

    def get_directory_name(self) -> Path:
        return self.directory_path

    def close(self) -> None:
        self.set_number_of_events()
        self.file.close()
