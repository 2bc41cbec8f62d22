"""Kinematics file -> one flat table (parquet), the reference's ``convert-kinematics`` tool
(reference src/attpc_engine/kinematics/convert_kinematics.py:11-63).

One row per (event, nucleus), event-major, with the columns of the reference's data frame in its order:
``event, Z, A, isotope, energy, px, py, pz, vertex_x, vertex_y, vertex_z`` (:29-41, :49-60; ``energy`` is column 3 of
the event's 4-vectors, px / py / pz columns 0 / 1 / 2).  The reference walks the HDF5 file event by event and lets
polars write the parquet file; here the table is built from whole blocks of events and written with polars when it
is installed, with pyarrow otherwise (same column names, order and values; polars' own parquet metadata is
"parity unpinned": neither polars nor h5py is installable in the build container).  The input is what
``run_kinematics_pipeline`` wrote: HDF5 with h5py, or this package's ``.npz`` stand-in for it.
"""
from __future__ import annotations

import argparse
from pathlib import Path

import numpy as np

from .. import nuclear_map
from ..io import KinematicsFileReader

COLUMNS = ("event", "Z", "A", "isotope", "energy", "px", "py", "pz", "vertex_x", "vertex_y", "vertex_z")


def kinematics_table(input_path: Path, block_events: int = 65536) -> dict:
    """The data frame of the reference as a dict of numpy arrays / a list of str, keyed by ``COLUMNS``."""
    input_path = Path(input_path)
    if not input_path.exists() and not input_path.with_suffix(".npz").exists():
        raise Exception(f"Input path {input_path} does not exist!")  # the reference's own error (:12-13)
    reader = KinematicsFileReader(input_path)
    z = np.asarray(reader.proton_numbers).astype(np.int64)
    a = np.asarray(reader.mass_numbers).astype(np.int64)
    n_nuclei, n_events = len(z), reader.n_events
    symbols = [nuclear_map.get_data(int(z[i]), int(a[i])).isotopic_symbol for i in range(n_nuclei)]
    vertex = np.empty((n_events, 3))
    p4 = np.empty((n_events, n_nuclei, 4))
    for start in range(0, n_events, block_events):
        stop = min(n_events, start + block_events)
        vertex[start:stop], p4[start:stop] = reader.read(start, stop)
    flat = p4.reshape(n_events * n_nuclei, 4)
    return {
        "event": np.repeat(np.arange(n_events, dtype=np.int64), n_nuclei),
        "Z": np.tile(z, n_events),
        "A": np.tile(a, n_events),
        "isotope": symbols * n_events,
        "energy": flat[:, 3].copy(),
        "px": flat[:, 0].copy(),
        "py": flat[:, 1].copy(),
        "pz": flat[:, 2].copy(),
        "vertex_x": np.repeat(vertex[:, 0], n_nuclei),
        "vertex_y": np.repeat(vertex[:, 1], n_nuclei),
        "vertex_z": np.repeat(vertex[:, 2], n_nuclei),
    }


def convert_kinematics_hdf5_to_polars(input_path: Path, output_path: Path) -> None:
    """Same name and arguments as the reference's converter (:11)."""
    table = kinematics_table(Path(input_path))
    try:
        import polars as pl  # type: ignore

        pl.DataFrame(data=table).write_parquet(output_path)
        return
    except ImportError:
        pass
    import pyarrow as pa
    import pyarrow.parquet as pq

    pq.write_table(pa.table({name: table[name] for name in COLUMNS}), str(output_path))


def main() -> None:
    parser = argparse.ArgumentParser(description="Convert the simulation kinematics HDF5 data to a dataframe")
    parser.add_argument("input", type=Path, help="The simulation HDF5 data")
    parser.add_argument("output", type=Path, help="The output dataframe file path (parquet)")
    args = parser.parse_args()
    convert_kinematics_hdf5_to_polars(args.input, args.output)


if __name__ == "__main__":
    main()
