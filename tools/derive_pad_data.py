"""Derive the packaged pad-plane DATA from the reference's data files.

Run once in the build container (the reference tree is absent on the GPU box):

    python3 -B tools/derive_pad_data.py

Reads  /root/reference/src/attpc_engine/detector/data/{pad_grid.npz,padxy.csv,pad_scale.csv}
Writes attpc_engine_amd/detector/data/{pad_lut_1mm.npz,pad_geometry.npz}

Only *data* is derived, no reference source is copied.

Why a 1 mm LUT is lossless: the reference floors every position to a whole
millimetre before it indexes the 0.1 mm grid (detector/transporter.py:110-118),
so only rows/cols 0,10,20,... of the 5600x5600 grid are ever read.  The derived
559x559 grid with edges [-280, 279, 1.0] gives the identical pad id under the
reference's own `position_to_index` for every input position.
"""
from pathlib import Path
import numpy as np

REF = Path("/root/reference/src/attpc_engine/detector/data")
OUT = Path(__file__).resolve().parent.parent / "attpc_engine_amd" / "detector" / "data"


def main() -> None:
    d = np.load(REF / "pad_grid.npz")  # allow_pickle=False (default)
    grid = d["grid"]
    edges = d["edges"]
    lo, hi, step = (float(v) for v in edges)
    ks = np.arange(int(np.ceil(lo)), int(np.ceil(hi)))  # integer mm with lo <= k < hi
    idx = ((np.floor(ks.astype(np.float64)) - lo) / step).astype(np.int64)
    lut = np.ascontiguousarray(grid[np.ix_(idx, idx)]).astype(np.int16)
    new_edges = np.array([float(ks[0]), float(ks[-1] + 1), 1.0])
    OUT.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT / "pad_lut_1mm.npz", grid=lut, edges=new_edges)

    centers = np.loadtxt(REF / "padxy.csv", delimiter=",", skiprows=1)
    sizes = np.loadtxt(REF / "pad_scale.csv", delimiter=",", skiprows=1)
    assert centers.shape == (10240, 2) and sizes.shape == (10240,)
    np.savez_compressed(OUT / "pad_geometry.npz", centers=centers, sizes=sizes)
    print("lut", lut.shape, new_edges, "beamless -1 frac", float((lut == -1).mean()))


if __name__ == "__main__":
    main()
