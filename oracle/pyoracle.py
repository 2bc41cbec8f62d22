"""ctypes access to the CPU oracle (oracle/attpc_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
the attpc_engine_amd package.

The oracle's descriptor structs have the same layout as the product's C ABI, so the
ctypes Structure classes of attpc_engine_amd._abi are reused to marshal inputs.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

from attpc_engine_amd import _abi
from attpc_engine_amd.detector.beam_pads import BEAM_PADS_ARRAY

HERE = Path(__file__).resolve().parent
# ATTPC_ORACLE_LIBRARY: another build of the same source (the sanitizer build of `make -C oracle asan`)
LIB_PATH = Path(os.environ["ATTPC_ORACLE_LIBRARY"]) if os.environ.get("ATTPC_ORACLE_LIBRARY") else HERE / "libattpc_oracle.so"
_dp = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)

_lib = None


def build() -> Path:
    subprocess.run(["make", "-C", str(HERE)], check=True, capture_output=True)
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        build()
    L = C.CDLL(str(LIB_PATH))
    L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.orc_philox4x32.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_int32, C.POINTER(C.c_uint32)]
    L.orc_rng_pair.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, _dp, _dp]
    L.orc_philox2x32.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_int32, C.POINTER(C.c_uint32)]
    L.orc_philox2x32.restype = None
    L.orc_jitter_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
    L.orc_jitter_uniform.restype = C.c_double
    L.orc_rng_normal.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
    L.orc_rng_normal.restype = C.c_double
    L.orc_reaction_allowed.argtypes = [_dp, C.c_double, C.c_double]
    L.orc_reaction_calculate.argtypes = [_dp, C.c_double, C.c_double, C.c_double, C.c_double, _dp]
    L.orc_decay_allowed.argtypes = [_dp, C.c_double, C.c_double, C.c_double]
    L.orc_decay_calculate.argtypes = [_dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _dp]
    L.orc_sample_excitation.argtypes = [C.POINTER(_abi.ExcitationDesc), C.c_double, C.c_double]
    L.orc_sample_excitation.restype = C.c_double
    L.orc_sample_polar.argtypes = [C.POINTER(_abi.PolarDesc), C.c_double, C.c_double]
    L.orc_sample_polar.restype = C.c_double
    L.orc_kin_event.argtypes = [C.POINTER(_abi.KinDesc), C.c_uint64, C.c_uint64, _dp, _dp, C.POINTER(C.c_uint32)]
    L.orc_kin_batch.argtypes = [C.POINTER(_abi.KinDesc), C.c_uint64, C.c_uint64, C.c_uint64, _dp, _dp,
                                _i32p, C.POINTER(C.c_uint32), C.c_int32]
    L.orc_kin_batch.restype = None
    L.orc_set_beam_pads.argtypes = [_i32p, C.c_int32]
    L.orc_set_beam_pads.restype = None
    L.orc_pair.argtypes = [C.c_int64, C.c_int64]
    L.orc_pair.restype = C.c_int64
    L.orc_unpair.argtypes = [C.c_int64, _i64p, _i64p]
    L.orc_unpair.restype = None
    L.orc_dedx_lookup.argtypes = [_dp, C.c_double]
    L.orc_dedx_lookup.restype = C.c_double
    L.orc_generate_trajectory.argtypes = [C.POINTER(_abi.DetDesc), C.POINTER(_abi.SpeciesDesc), _dp, _dp, _dp]
    L.orc_generate_electrons.argtypes = [C.POINTER(_abi.DetDesc), C.POINTER(_abi.SpeciesDesc), _dp, C.c_int32,
                                         C.c_uint64, C.c_uint64, C.c_uint32, _i64p]
    L.orc_generate_electrons.restype = None
    L.orc_dict_new.restype = C.c_void_p
    L.orc_dict_free.argtypes = [C.c_void_p]
    L.orc_dict_free.restype = None
    L.orc_dict_len.argtypes = [C.c_void_p]
    L.orc_dict_len.restype = C.c_int64
    L.orc_dict_item.argtypes = [C.c_void_p, C.c_int64, _i64p, _i64p, _i64p]
    L.orc_dict_item.restype = None
    L.orc_transport_track.argtypes = [C.POINTER(_abi.DetDesc), _dp, _i64p, C.c_int32, C.c_void_p, C.c_int64]
    L.orc_transport_track.restype = None
    L.orc_generate_point_cloud.argtypes = [C.POINTER(_abi.DetDesc), C.POINTER(_abi.SpeciesDesc), _dp, _dp,
                                           C.c_uint64, C.c_uint64, C.c_int64, C.c_void_p, _dp, _i32p]
    L.orc_simulate.argtypes = [C.POINTER(_abi.DetDesc), C.POINTER(_abi.EventLayout), C.c_uint64, C.c_uint64,
                               _dp, _dp, C.c_int64, _dp, _i64p, C.POINTER(C.c_uint64)]
    L.orc_simulate.restype = C.c_int64
    L.orc_sim_batch.argtypes = [C.POINTER(_abi.KinDesc), C.POINTER(_abi.DetDesc), C.POINTER(_abi.EventLayout),
                                C.c_uint64, C.c_uint64, C.c_uint64, _dp, _dp, _i32p, C.c_int64, _i64p, _dp,
                                _i64p, C.POINTER(C.c_uint64), C.c_int32]
    L.orc_sim_batch.restype = C.c_int64
    L.orc_get_response.argtypes = [C.c_double, C.c_double, C.c_double, _dp]
    L.orc_get_response.restype = None
    L.orc_convert_to_spyral.argtypes = [_dp, C.c_int64, C.c_int32, C.c_int32, C.c_double, _dp, _dp, _dp, _dp]
    L.orc_convert_to_spyral.restype = None
    pads = np.ascontiguousarray(BEAM_PADS_ARRAY, dtype=np.int32)
    L.orc_set_beam_pads(pads.ctypes.data_as(_i32p), len(pads))
    _lib = L
    return L


def d(arr):
    return None if arr is None else arr.ctypes.data_as(_dp)


# ------------------------------------------------------------------ convenience -------
def philox(ctr, key, rounds: int = 10) -> np.ndarray:
    c = (C.c_uint32 * 4)(*[int(v) for v in ctr])
    k = (C.c_uint32 * 2)(*[int(v) for v in key])
    out = (C.c_uint32 * 4)()
    lib().orc_philox4x32(c, k, int(rounds), out)
    return np.array(list(out), dtype=np.uint32)


def philox2x32(ctr, key: int, rounds: int = 10) -> np.ndarray:
    c = (C.c_uint32 * 2)(*[int(v) for v in ctr])
    out = (C.c_uint32 * 2)()
    lib().orc_philox2x32(c, C.c_uint32(int(key)), int(rounds), out)
    return np.array(list(out), dtype=np.uint32)


def jitter_uniform(seed: int, event: int, key24: int) -> float:
    return float(lib().orc_jitter_uniform(int(seed), int(event), int(key24)))


def kin_calculate(desc: _abi.KinDesc, beam, ex, th, ph):
    """Deterministic map parameters -> (p4, status) with the golden fixtures' status
    convention: 0 ok, 1 reaction not allowed, -1 below NR threshold, k+1 decay k not allowed."""
    L = lib()
    n_steps = desc.n_steps
    n_rows = 4 + 2 * (n_steps - 1)
    beam = np.atleast_1d(np.asarray(beam, dtype=np.float64))
    n = beam.size
    ex = np.asarray(ex, dtype=np.float64).reshape(n, n_steps)
    th = np.asarray(th, dtype=np.float64).reshape(n, n_steps)
    ph = np.asarray(ph, dtype=np.float64).reshape(n, n_steps)
    p4 = np.full((n, n_rows, 4), np.nan)
    status = np.zeros(n, dtype=np.int32)
    masses = np.array(list(desc.masses), dtype=np.float64)
    for i in range(n):
        if not L.orc_reaction_allowed(d(masses), beam[i], ex[i, 0]):
            status[i] = 1
            continue
        rows = np.empty((4, 4))
        if L.orc_reaction_calculate(d(masses), beam[i], th[i, 0], ph[i, 0], ex[i, 0], d(rows)):
            status[i] = -1
            continue
        p4[i, :4] = rows
        prev = rows[3].copy()
        for s in range(1, n_steps):
            m1, m2 = masses[4 + 2 * (s - 1)], masses[5 + 2 * (s - 1)]
            if not L.orc_decay_allowed(d(prev), m1, m2, ex[i, s]):
                status[i] = s + 1
                break
            out = np.empty((2, 4))
            L.orc_decay_calculate(d(prev), m1, m2, th[i, s], ph[i, s], ex[i, s], d(out))
            p4[i, 4 + 2 * (s - 1): 6 + 2 * (s - 1)] = out
            prev = out[1].copy()
    return p4, status


def kin_batch(desc: _abi.KinDesc, seed: int, first: int, n: int, threads: int = 1):
    n_rows = 4 + 2 * (desc.n_steps - 1)
    p4 = np.empty((n, n_rows, 4))
    vertex = np.empty((n, 3))
    status = np.empty(n, dtype=np.int32)
    attempts = np.empty(n, dtype=np.uint32)
    lib().orc_kin_batch(desc, seed, first, n, d(p4), d(vertex), status.ctypes.data_as(_i32p),
                        attempts.ctypes.data_as(C.POINTER(C.c_uint32)), threads)
    return vertex, p4, status, attempts


def trajectory(det: _abi.DetDesc, species_index: int, vertex, momentum) -> np.ndarray:
    track = np.empty((_abi.TIME_SAMPLES, 6))
    vertex = np.ascontiguousarray(vertex, dtype=np.float64)
    momentum = np.ascontiguousarray(momentum, dtype=np.float64)
    n = lib().orc_generate_trajectory(det, det.species[species_index], d(vertex), d(momentum), d(track))
    return track[:n].copy()


def electrons(det: _abi.DetDesc, species_index: int, track: np.ndarray, seed: int, event: int, domain: int):
    track = np.ascontiguousarray(track, dtype=np.float64)
    out = np.empty(len(track), dtype=np.int64)
    lib().orc_generate_electrons(det, det.species[species_index], d(track), len(track), seed, event, domain,
                                 out.ctypes.data_as(_i64p))
    return out


def transport(det: _abi.DetDesc, cases):
    """cases: list of (xyt [n,3], electrons [n] int64, label) -> (keys, charge, labels) in
    dictionary insertion order."""
    L = lib()
    handle = L.orc_dict_new()
    for xyt, elec, label in cases:
        xyt = np.ascontiguousarray(xyt, dtype=np.float64)
        elec = np.ascontiguousarray(elec, dtype=np.int64)
        L.orc_transport_track(det, d(xyt), elec.ctypes.data_as(_i64p), len(xyt), handle, int(label))
    n = L.orc_dict_len(handle)
    keys = np.empty(n, dtype=np.int64)
    charge = np.empty(n, dtype=np.int64)
    labels = np.empty(n, dtype=np.int64)
    k, c, l = C.c_int64(), C.c_int64(), C.c_int64()
    for i in range(n):
        L.orc_dict_item(handle, i, C.byref(k), C.byref(c), C.byref(l))
        keys[i], charge[i], labels[i] = k.value, c.value, l.value
    L.orc_dict_free(handle)
    return keys, charge, labels


def unpair(key: int) -> tuple[int, int]:
    """(time bucket, pad) of a dictionary key (reference pairing.py:30-55)."""
    tb, pad = C.c_int64(), C.c_int64()
    lib().orc_unpair(int(key), C.byref(tb), C.byref(pad))
    return tb.value, pad.value


def point_cloud_samples(det: _abi.DetDesc, species_index: int, momentum, vertex, seed: int, event: int,
                        label: int):
    """kept samples [m,4] (x, y, time bucket, electrons*gain) and the number of ODE rows."""
    L = lib()
    handle = L.orc_dict_new()
    samples = np.empty((_abi.TIME_SAMPLES, 4))
    n_rows = C.c_int32()
    momentum = np.ascontiguousarray(momentum, dtype=np.float64)
    vertex = np.ascontiguousarray(vertex, dtype=np.float64)
    m = L.orc_generate_point_cloud(det, det.species[species_index], d(momentum), d(vertex), seed, event,
                                   label, handle, d(samples), C.byref(n_rows))
    L.orc_dict_free(handle)
    return samples[:m].copy(), n_rows.value


def simulate(det: _abi.DetDesc, layout: _abi.EventLayout, seed: int, event: int, p4, vertex,
             capacity: int = 1 << 16):
    p4 = np.ascontiguousarray(p4, dtype=np.float64)
    vertex = np.ascontiguousarray(vertex, dtype=np.float64)
    pts = np.empty((capacity, 3))
    lab = np.empty(capacity, dtype=np.int64)
    ns = C.c_uint64()
    n = lib().orc_simulate(det, layout, seed, event, d(p4), d(vertex), capacity, d(pts),
                           lab.ctypes.data_as(_i64p), C.byref(ns))
    if n < 0:
        raise BufferError("oracle simulate capacity")
    return pts[:n].copy(), lab[:n].copy(), ns.value


def sim_batch(kin, det, layout, seed: int, first: int, n: int, capacity: int | None = None, threads: int = 1,
              p4_in=None, vertex_in=None):
    """Fused kinematics+detector (kin may be None with p4_in/vertex_in given).
    Returns dict(vertex, p4, status, offsets, points, labels, stats)."""
    n_rows = layout.n_rows
    p4 = np.ascontiguousarray(p4_in, dtype=np.float64) if p4_in is not None else np.empty((n, n_rows, 4))
    vertex = np.ascontiguousarray(vertex_in, dtype=np.float64) if vertex_in is not None else np.empty((n, 3))
    status = np.zeros(n, dtype=np.int32)
    stats = (C.c_uint64 * 4)()
    offsets = np.zeros(n + 1, dtype=np.int64)
    if capacity is None:
        total = lib().orc_sim_batch(kin, det, layout, seed, first, n, d(p4), d(vertex),
                                    status.ctypes.data_as(_i32p), 0, offsets.ctypes.data_as(_i64p), None, None,
                                    stats, threads)
        return {"vertex": vertex, "p4": p4, "status": status, "offsets": offsets, "points": None,
                "labels": None, "stats": list(stats), "total": total}
    pts = np.empty((capacity, 3))
    lab = np.empty(capacity, dtype=np.int64)
    total = lib().orc_sim_batch(kin, det, layout, seed, first, n, d(p4), d(vertex), status.ctypes.data_as(_i32p),
                                capacity, offsets.ctypes.data_as(_i64p), d(pts), lab.ctypes.data_as(_i64p), stats,
                                threads)
    if total < 0:
        raise BufferError(f"oracle sim_batch needs capacity {-total}")
    return {"vertex": vertex, "p4": p4, "status": status, "offsets": offsets, "points": pts[:total],
            "labels": lab[:total], "stats": list(stats), "total": total}
