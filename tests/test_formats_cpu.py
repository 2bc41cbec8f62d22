"""On-disk layouts driven through a stand-in ``h5py`` module.

h5py is not importable in the build container, so the real HDF5 file format stays "parity
unpinned"; what IS pinned here is everything the product code decides: group / dataset / attribute
names, dtypes, event numbering and file roll-over of ``io.KinematicsFileWriter`` and
``detector.SpyralWriter`` against the reference's writers (kinematics/pipeline.py:449-493,
detector/writer.py:179-192, 240-263), plus the reader on the same layout.
"""
import sys
import types
import warnings
from pathlib import Path

import numpy as np
import pytest


class _Attrs(dict):
    pass


class _Dataset:
    def __init__(self, data):
        self.data = np.array(data)
        self.attrs = _Attrs()

    def __getitem__(self, key):
        return self.data[key]


class _Group(dict):
    def __init__(self):
        super().__init__()
        self.attrs = _Attrs()

    def create_group(self, name):
        assert name not in self, f"group {name} created twice"
        self[name] = _Group()
        return self[name]

    def create_dataset(self, name, data=None):
        assert name not in self, f"dataset {name} created twice"
        self[name] = _Dataset(data)
        return self[name]


class _File(_Group):
    registry: dict = {}

    def __init__(self, path, mode="r"):
        super().__init__()
        path = str(path)
        if mode == "w":
            _File.registry[path] = self
            self.closed = False
        else:
            stored = _File.registry[path]
            self.update(stored)
            self.attrs = stored.attrs
            self.closed = False
        self.path, self.mode = path, mode

    def close(self):
        self.closed = True


@pytest.fixture
def fake_h5py(monkeypatch):
    mod = types.ModuleType("h5py")
    mod.File, mod.Group, mod.Dataset = _File, _Group, _Dataset
    _File.registry = {}
    monkeypatch.setitem(sys.modules, "h5py", mod)
    return mod


def test_kinematics_file_layout(fake_h5py, tmp_path):
    """reference run_kinematics_pipeline, pipeline.py:449-493"""
    from attpc_engine_amd.io import KinematicsFileReader, KinematicsFileWriter

    rng = np.random.default_rng(1)
    n, chunk = 11, 4
    vertex, p4 = rng.normal(size=(n, 3)), rng.normal(size=(n, 6, 4))
    z, a = np.array([2, 8, 2, 8, 2, 6]), np.array([4, 16, 4, 16, 4, 12])
    path = tmp_path / "kin.h5"
    w = KinematicsFileWriter(path, n, z, a, chunk)
    w.write_batch(0, vertex[:7], p4[:7])
    w.write_batch(7, vertex[7:], p4[7:])
    w.close()
    f = _File.registry[str(path)]
    assert f.closed and list(f) == ["data"]
    data = f["data"]
    assert set(data.attrs) == {"n_events", "proton_numbers", "mass_numbers", "chunk_size", "n_chunks"}
    assert data.attrs["n_events"] == n and data.attrs["chunk_size"] == chunk and data.attrs["n_chunks"] == 3
    np.testing.assert_array_equal(data.attrs["proton_numbers"], z)
    np.testing.assert_array_equal(data.attrs["mass_numbers"], a)
    assert sorted(data) == ["chunk_0", "chunk_1", "chunk_2"]
    for c, (lo, hi) in enumerate([(0, 3), (4, 7), (8, 10)]):
        grp = data[f"chunk_{c}"]
        assert grp.attrs["min_event"] == lo and grp.attrs["max_event"] == hi
        assert sorted(grp, key=lambda k: int(k.split("_")[1])) == [f"event_{e}" for e in range(lo, hi + 1)]
        for e in range(lo, hi + 1):
            d = grp[f"event_{e}"]
            assert d.data.dtype == np.float64 and d.data.shape == (6, 4)
            np.testing.assert_array_equal(d.data, p4[e])
            assert set(d.attrs) == {"vertex_x", "vertex_y", "vertex_z"}
            assert [d.attrs["vertex_x"], d.attrs["vertex_y"], d.attrs["vertex_z"]] == list(vertex[e])
    path.touch()  # the reader checks that the .h5 exists (the stand-in keeps files in memory)
    r = KinematicsFileReader(path)
    assert r.n_events == n and r.n_chunks == 3 and r.chunk_size == chunk
    v, q = r.read(2, 9)
    np.testing.assert_array_equal(v, vertex[2:9])
    np.testing.assert_array_equal(q, p4[2:9])


def _spyral_config():
    from attpc_engine_amd import workloads

    _, config, _ = workloads.be10dp()
    return config


def test_spyral_writer_layout(fake_h5py, tmp_path):
    """reference SpyralWriter, writer.py:164-192 (files, group), :214-218 (roll-over), :240-263
    (datasets, attributes, min/max event); rows handed over already converted (write_rows)."""
    from attpc_engine_amd.detector import SpyralWriter

    config = _spyral_config()
    w = SpyralWriter(tmp_path, config, max_events_per_file=2, first_run_number=7)
    rng = np.random.default_rng(2)
    written = {}
    for ev in (3, 4, 9):
        rows = rng.normal(size=(5, 8))
        labels = rng.integers(0, 6, size=5)
        w.write_rows(rows, labels, ev)
        order = np.argsort(rows[:, 2])
        written[ev] = (rows[order], labels[order])
    w.write_rows(np.empty((0, 8)), np.empty(0, dtype=np.int64), 12)  # all rows below threshold: still an event
    w.close()
    assert sorted(Path(p).name for p in _File.registry) == ["run_0007.h5", "run_0008.h5"]
    f0, f1 = _File.registry[str(tmp_path / "run_0007.h5")], _File.registry[str(tmp_path / "run_0008.h5")]
    assert f0.closed and f1.closed and list(f0) == ["cloud"] and list(f1) == ["cloud"]
    g0, g1 = f0["cloud"], f1["cloud"]
    assert sorted(g0) == ["cloud_3", "cloud_4", "labels_3", "labels_4"]
    assert sorted(g1) == ["cloud_12", "cloud_9", "labels_12", "labels_9"]
    assert (g0.attrs["min_event"], g0.attrs["max_event"]) == (0, 4)  # starting_event starts at 0, writer.py:173
    assert (g1.attrs["min_event"], g1.attrs["max_event"]) == (9, 12)
    for grp, run, events in ((g0, 7, (3, 4)), (g1, 8, (9,))):
        for ev in events:
            d = grp[f"cloud_{ev}"]
            assert d.data.dtype == np.float64 and d.data.shape == (5, 8)
            np.testing.assert_array_equal(d.data, written[ev][0])
            np.testing.assert_array_equal(grp[f"labels_{ev}"].data, written[ev][1])
            assert dict(d.attrs) == {"orig_run": run, "orig_event": ev, "ic_amplitude": -1.0,
                                     "ic_multiplicity": -1.0, "ic_integral": -1.0, "ic_centroid": -1.0}
            assert not grp[f"labels_{ev}"].attrs
    assert g1["cloud_12"].data.shape == (0, 8)


def test_presorted_rows_are_not_resorted(fake_h5py, tmp_path):
    from attpc_engine_amd.detector import SpyralWriter

    w = SpyralWriter(tmp_path, _spyral_config())
    rows = np.arange(24, dtype=np.float64).reshape(3, 8)[::-1].copy()  # descending z
    w.write_rows(rows, np.arange(3), 0, presorted=True)
    w.close()
    np.testing.assert_array_equal(_File.registry[str(tmp_path / "run_0000.h5")]["cloud"]["cloud_0"].data, rows)


def test_missing_h5py_warns_or_raises(tmp_path, monkeypatch):
    """a requested .h5 must not silently become .npz."""
    from attpc_engine_amd.detector import SpyralWriter
    from attpc_engine_amd.io import KinematicsFileWriter

    monkeypatch.setitem(sys.modules, "h5py", None)  # import h5py -> ImportError
    with pytest.warns(RuntimeWarning, match="h5py is not installed"):
        w = KinematicsFileWriter(tmp_path / "kin.h5", 1, [1], [1], 4)
    w.write_batch(0, np.zeros((1, 3)), np.zeros((1, 1, 4)))
    w.close()
    assert (tmp_path / "kin.npz").exists() and not (tmp_path / "kin.h5").exists()
    with pytest.raises(ImportError):
        KinematicsFileWriter(tmp_path / "kin2.h5", 1, [1], [1], 4, npz_fallback=False)
    with pytest.warns(RuntimeWarning, match="h5py is not installed"):
        sw = SpyralWriter(tmp_path, _spyral_config())
    sw.close()
    with pytest.raises(ImportError):
        SpyralWriter(tmp_path, _spyral_config(), npz_fallback=False)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        KinematicsFileWriter(tmp_path / "kin3.npz", 1, [1], [1], 4)  # an explicit .npz needs no warning


def test_convert_kinematics_table(tmp_path):
    """The reference's convert-kinematics tool (convert_kinematics.py:11-63): one row per (event, nucleus),
    event-major, columns event, Z, A, isotope, energy (= column 3 of the 4-vector), px, py, pz, vertex_x/y/z.
    Written here from whole blocks of events with polars or pyarrow; the values and their order are checked against
    the file's contents (the parquet bytes of polars themselves are parity unpinned: polars is not installable here)."""
    import warnings

    import pyarrow.parquet as pq

    from attpc_engine_amd import nuclear_map
    from attpc_engine_amd.io import KinematicsFileWriter
    from attpc_engine_amd.kinematics.convert_kinematics import COLUMNS, convert_kinematics_hdf5_to_polars

    rng = np.random.default_rng(7)
    z, a = np.array([2, 8, 2, 8, 2, 6]), np.array([4, 16, 4, 16, 4, 12])
    n = 37
    vertex = rng.normal(size=(n, 3))
    p4 = rng.normal(size=(n, len(z), 4))
    path = tmp_path / "kine.npz"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        writer = KinematicsFileWriter(path, n, z, a, chunk_size=10)
        writer.write_batch(0, vertex[:20], p4[:20])
        writer.write_batch(20, vertex[20:], p4[20:])
        writer.close()
    out = tmp_path / "kine.parquet"
    convert_kinematics_hdf5_to_polars(path, out)
    table = pq.read_table(str(out))
    assert tuple(table.column_names) == COLUMNS
    assert table.num_rows == n * len(z)
    cols = {name: table.column(name).to_numpy(zero_copy_only=False) for name in COLUMNS}
    assert np.array_equal(cols["event"], np.repeat(np.arange(n), len(z)))
    assert np.array_equal(cols["Z"], np.tile(z, n)) and np.array_equal(cols["A"], np.tile(a, n))
    assert list(cols["isotope"][: len(z)]) == [nuclear_map.get_data(int(zz), int(aa)).isotopic_symbol for zz, aa in zip(z, a)]
    flat = p4.reshape(-1, 4)
    for name, col in (("px", 0), ("py", 1), ("pz", 2), ("energy", 3)):
        assert np.array_equal(cols[name], flat[:, col]), name
    for k, name in enumerate(("vertex_x", "vertex_y", "vertex_z")):
        assert np.array_equal(cols[name], np.repeat(vertex[:, k], len(z))), name
    with pytest.raises(Exception, match="does not exist"):
        convert_kinematics_hdf5_to_polars(tmp_path / "missing.h5", out)
