"""Kinematics file layout shared by the two stages (reference pipeline.py:449-493 writes
it, simulator.py:146-202 reads it).  HDF5 through h5py when it is importable; otherwise
the same content in one ``.npz`` (h5py is not installed in the build container)."""
from __future__ import annotations

from pathlib import Path

import numpy as np


def _h5py():
    try:
        import h5py  # type: ignore

        return h5py
    except ImportError:
        return None


def hdf5_or_fallback(path: Path, npz_fallback: bool):
    """h5py if it is importable; otherwise ``None`` after a warning that ``path`` is written as
    ``.npz`` (which the reference and Spyral cannot read), or ImportError when the fallback was declined."""
    h5 = _h5py()
    if h5 is None:
        message = (f"h5py is not installed: {path} cannot be written as HDF5; the same datasets go to "
                   f"{Path(path).with_suffix('.npz')} (not readable by the reference / Spyral tooling)")
        if not npz_fallback:
            raise ImportError(message + " -- install h5py or pass npz_fallback=True")
        import warnings

        warnings.warn(message, RuntimeWarning, stacklevel=3)
    return h5


class KinematicsFileWriter:
    def __init__(self, path: Path, n_events: int, proton_numbers, mass_numbers, chunk_size: int,
                 npz_fallback: bool = True):
        self.path = Path(path)
        self.n_events = int(n_events)
        self.chunk_size = int(chunk_size)
        self.h5 = hdf5_or_fallback(self.path, npz_fallback) if self.path.suffix.lower() in (".h5", ".hdf5") else None
        self.z = np.asarray(proton_numbers)
        self.a = np.asarray(mass_numbers)
        if self.h5 is not None:
            self.file = self.h5.File(self.path, "w")
            self.group = self.file.create_group("data")
            self.group.attrs["n_events"] = self.n_events
            self.group.attrs["proton_numbers"] = self.z
            self.group.attrs["mass_numbers"] = self.a
            self.group.attrs["chunk_size"] = self.chunk_size
            self.chunks = {}
        else:
            self.vertex = np.empty((self.n_events, 3))
            self.p4 = np.empty((self.n_events, len(self.z), 4))

    def write_batch(self, first_event: int, vertex: np.ndarray, p4: np.ndarray) -> None:
        if self.h5 is None:
            self.vertex[first_event:first_event + len(p4)] = vertex
            self.p4[first_event:first_event + len(p4)] = p4
            return
        for i in range(len(p4)):
            event = first_event + i
            chunk = event // self.chunk_size
            if chunk not in self.chunks:
                grp = self.group.create_group(f"chunk_{chunk}")
                grp.attrs["min_event"] = chunk * self.chunk_size
                grp.attrs["max_event"] = min(self.n_events, (chunk + 1) * self.chunk_size) - 1
                self.chunks[chunk] = grp
            dset = self.chunks[chunk].create_dataset(f"event_{event}", data=p4[i])
            dset.attrs["vertex_x"], dset.attrs["vertex_y"], dset.attrs["vertex_z"] = vertex[i]

    def close(self) -> None:
        n_chunks = max(1, -(-self.n_events // self.chunk_size))
        if self.h5 is not None:
            self.group.attrs["n_chunks"] = n_chunks
            self.file.close()
        else:
            target = self.path if self.path.suffix == ".npz" else self.path.with_suffix(".npz")
            np.savez(target, n_events=self.n_events, proton_numbers=self.z, mass_numbers=self.a,
                     chunk_size=self.chunk_size, n_chunks=n_chunks, vertex=self.vertex, p4=self.p4)


class KinematicsFileReader:
    def __init__(self, path: Path):
        self.path = Path(path)
        npz_path = self.path if self.path.suffix == ".npz" else self.path.with_suffix(".npz")
        self.h5 = None
        if self.path.suffix.lower() in (".h5", ".hdf5") and self.path.exists() and _h5py() is not None:
            self.h5 = _h5py()
            self.file = self.h5.File(self.path, "r")
            grp = self.file["data"]
            self.group = grp
            self.proton_numbers = np.asarray(grp.attrs["proton_numbers"])
            self.mass_numbers = np.asarray(grp.attrs["mass_numbers"])
            self.n_events = int(grp.attrs["n_events"])
            self.n_chunks = int(grp.attrs["n_chunks"])
            self.chunk_size = int(grp.attrs["chunk_size"])
        elif npz_path.exists():
            data = np.load(npz_path)
            self.proton_numbers = data["proton_numbers"]
            self.mass_numbers = data["mass_numbers"]
            self.n_events = int(data["n_events"])
            self.n_chunks = int(data["n_chunks"])
            self.chunk_size = int(data["chunk_size"])
            self._vertex = data["vertex"]
            self._p4 = data["p4"]
        else:
            raise FileNotFoundError(f"no kinematics file at {self.path} (or {npz_path})")

    def read(self, start: int, stop: int):
        """-> (vertex [n,3], p4 [n,N,4]) for events start..stop-1."""
        if self.h5 is None:
            return self._vertex[start:stop].copy(), self._p4[start:stop].copy()
        n = stop - start
        vertex = np.empty((n, 3))
        p4 = np.empty((n, len(self.proton_numbers), 4))
        for i in range(n):
            event = start + i
            dset = self.group[f"chunk_{event // self.chunk_size}"][f"event_{event}"]
            p4[i] = dset[:]
            vertex[i] = [dset.attrs["vertex_x"], dset.attrs["vertex_y"], dset.attrs["vertex_z"]]
        return vertex, p4
