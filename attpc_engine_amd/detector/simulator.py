"""Detector simulation entry points (reference ``detector/simulator.py``), device backed.

``simulate`` keeps the reference signature for one event; ``simulate_batch`` is the same
operator over many events (one HIP launch sequence), and ``run_simulation`` drives a
kinematics file through it in batches, calling the writer once per non-empty event in
event order, exactly like the reference loop (simulator.py:183-208).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
from numpy.random import Generator, default_rng

from .. import _abi
from .luts import build_det_desc, build_layout, species_for
from .parameters import Config
from .writer import SimulationWriter


def default_indices(n_rows: int) -> list[int]:
    """All final products: rows 2, 4, 6, ... plus the last row (simulator.py:157-158)."""
    indices = [idx for idx in range(2, n_rows, 2)]
    indices.append(n_rows - 1)
    return indices


def _nuclear_map():
    from .. import nuclear_map

    return nuclear_map


def _digest(struct, arrays) -> bytes:
    """Content of a descriptor: its scalar fields and the arrays its pointers refer to (the pointer values
    themselves are left out -- they differ from build to build of the same content)."""
    import ctypes
    import hashlib

    copy = type(struct).from_buffer_copy(bytes(struct))
    for name, ctype in copy._fields_:
        if isinstance(getattr(copy, name), ctypes._Pointer):
            setattr(copy, name, ctypes.cast(None, ctype))
    if hasattr(copy, "species"):
        for sp in copy.species:
            sp.dedx = ctypes.cast(None, type(sp.dedx))
    h = hashlib.blake2b(bytes(copy), digest_size=16)
    for arr in arrays:
        h.update(np.ascontiguousarray(arr).tobytes())
    return h.digest()


def configure_detector(config: Config, species_keys: list[tuple[int, int]], ctx: _abi.Context,
                       ode_substeps: int = 1) -> None:
    """Upload Config + species tables unless this ctx already holds the same ones.  "The same" is decided on the
    CONTENT of the descriptor (every parameter, the pad look-up table, the stopping-power tables), not on object
    identity: a parameter changed in place on the same Config object is seen."""
    nuclei = [_nuclear_map().get_data(z, a) for (z, a) in species_keys]
    keys: list = []
    desc, keep = build_det_desc(config, nuclei, ode_substeps=ode_substeps, content_keys=keys)
    # (the tables are memoised on their inputs' content, luts.py: their keys stand for them; a table without a key --
    #  a target object that cannot be compared by value -- is hashed itself)
    token = (_digest(desc, [arr for arr, key in zip(keep, keys) if key is None]), tuple(keys))
    if getattr(ctx, "_det_token", None) == token:
        return
    ctx.check(ctx.lib.attpc_det_configure(ctx.handle, desc), "attpc_det_configure")
    ctx._det_token = token
    del keep


def configure_spyral(config: Config, ctx: _abi.Context, response: np.ndarray | None = None) -> None:
    """Upload what SpyralWriter.write needs per event (reference writer.py:164-181, 220-234) unless this ctx already
    holds the same: the GET response of the electronics (``response``: the writer's own, writer.py:176; default
    get_response(config)), pad centres / sizes, ADC threshold and time-bucket edges."""
    from .response import get_response

    if config.pad_centers is None:
        raise ValueError("Pad centers are not assigned at write!")  # writer.py:220-221
    response = np.ascontiguousarray(get_response(config) if response is None else response, dtype=np.float64)
    centers = np.ascontiguousarray(config.pad_centers, dtype=np.float64)
    sizes = np.ascontiguousarray(config.pad_sizes, dtype=np.float64)
    desc = _abi.SpyralDesc(_abi.dptr(response), _abi.dptr(centers), _abi.dptr(sizes), len(sizes),
                           int(config.elec_params.windows_edge), int(config.elec_params.micromegas_edge), 0,
                           float(config.det_params.length), float(config.elec_params.adc_threshold))
    token = _digest(desc, [response, centers, sizes])
    if getattr(ctx, "_spyral_token", None) == token:
        return
    ctx.check(ctx.lib.attpc_spyral_configure(ctx.handle, desc), "attpc_spyral_configure")
    ctx._spyral_token = token


def simulate_batch(momenta: np.ndarray, vertices: np.ndarray, proton_numbers, mass_numbers,
                   config: Config, seed: int, indices: list[int], first_event: int = 0,
                   ctx: _abi.Context | None = None, capacity_per_event: int = 16384):
    """simulate() for n events: momenta [n,N,4], vertices [n,3] ->
    (offsets [n+1], points [P,3], labels [P], stats dict)."""
    ctx = ctx or _abi.default_context()
    momenta = np.ascontiguousarray(momenta, dtype=np.float64)
    vertices = np.ascontiguousarray(vertices, dtype=np.float64)
    n = momenta.shape[0]
    keys = species_for(proton_numbers, mass_numbers, indices)
    configure_detector(config, keys, ctx)
    layout = build_layout(proton_numbers, mass_numbers, indices, keys)
    capacity = max(1024, int(capacity_per_event) * n)
    while True:
        offsets = np.zeros(n + 1, dtype=np.int64)
        points = np.empty((capacity, 3), dtype=np.float64)
        labels = np.empty(capacity, dtype=np.int64)
        out = _abi.CloudOut(capacity, _abi.iptr(offsets, _abi.C.c_int64), _abi.dptr(points),
                            _abi.iptr(labels, _abi.C.c_int64))
        stats = _abi.RunStats()
        status = ctx.lib.attpc_det_run(
            ctx.handle, int(seed), int(first_event), n, layout, _abi.dptr(momenta),
            _abi.dptr(vertices), out, stats,
        )
        if status == _abi.E_CAPACITY:
            capacity = int(stats.n_points) + 1024
            continue
        ctx.check(status, "attpc_det_run")
        break
    total = int(offsets[n])
    return offsets, points[:total], labels[:total], stats.as_dict()


def simulate_batch_spyral(momenta: np.ndarray, vertices: np.ndarray, proton_numbers, mass_numbers,
                          config: Config, seed: int, indices: list[int], first_event: int = 0,
                          ctx: _abi.Context | None = None, response: np.ndarray | None = None,
                          capacity_per_event: int = 8192):
    """simulate() + what SpyralWriter.write does per event (convert_to_spyral, ADC threshold, z-sort; reference
    writer.py:194-238) for n events in one launch sequence, all on the device (``attpc_det_run_spyral``) ->
    (offsets [n+1], rows [P',8], labels [P'], event_points [n] = cloud rows of every event BEFORE the threshold,
    stats dict)."""
    ctx = ctx or _abi.default_context()
    momenta = np.ascontiguousarray(momenta, dtype=np.float64)
    vertices = np.ascontiguousarray(vertices, dtype=np.float64)
    n = momenta.shape[0]
    keys = species_for(proton_numbers, mass_numbers, indices)
    configure_detector(config, keys, ctx)
    configure_spyral(config, ctx, response)
    layout = build_layout(proton_numbers, mass_numbers, indices, keys)
    capacity = max(1024, int(capacity_per_event) * n)
    while True:
        offsets = np.zeros(n + 1, dtype=np.int64)
        rows = np.empty((capacity, 8), dtype=np.float64)
        labels = np.empty(capacity, dtype=np.int64)
        event_points = np.zeros(n, dtype=np.int64)
        out = _abi.CloudOut(capacity, _abi.iptr(offsets, _abi.C.c_int64), _abi.dptr(rows),
                            _abi.iptr(labels, _abi.C.c_int64), _abi.iptr(event_points, _abi.C.c_int64))
        stats = _abi.RunStats()
        status = ctx.lib.attpc_det_run_spyral(
            ctx.handle, int(seed), int(first_event), n, layout, _abi.dptr(momenta),
            _abi.dptr(vertices), out, stats,
        )
        if status == _abi.E_CAPACITY:
            capacity = int(stats.n_points) + 1024
            continue
        ctx.check(status, "attpc_det_run_spyral")
        break
    total = int(offsets[n])
    return offsets, rows[:total], labels[:total], event_points, stats.as_dict()


def simulate(momenta: np.ndarray, vertex: np.ndarray, proton_numbers: np.ndarray,
             mass_numbers: np.ndarray, config: Config, rng: Generator, indices: list[int]):
    """One kinematics event -> (points [P,3] = pad, time bucket, electrons; labels [P])
    (reference simulator.py:52-115).  ``rng`` seeds the device Philox streams (one draw).
    Row order is unspecified (the reference's is dict-insertion order); use
    ``np.lexsort((points[:,1], points[:,0]))`` for a canonical order."""
    seed = int(rng.integers(0, 1 << 63))
    momenta = np.ascontiguousarray(momenta, dtype=np.float64)[None]
    vertex = np.ascontiguousarray(vertex, dtype=np.float64)[None]
    _, points, labels, _ = simulate_batch(momenta, vertex, proton_numbers, mass_numbers, config,
                                          seed, list(indices))
    return points, labels


def run_simulation(config: Config, input_path: Path, writer: SimulationWriter,
                   indices: list[int] | None = None, batch_size: int = 16384,
                   seed: int | None = None):
    """Apply the detector simulation to every event of a kinematics file (reference
    simulator.py:118-210): the writer is called once per event with a non-empty cloud, in event order, then
    closed.  A writer that offers ``write_rows`` (SpyralWriter) receives its rows ready to store: the response
    scaling, row conversion, ADC threshold and z-sort it would do per event in ``write`` (writer.py:194-238) run on
    the device, fused behind the scatter, before anything crosses PCIe (``attpc_det_run_spyral``) -- the same
    datasets as ``write`` produces, without one GPU round trip per event.  Any other SimulationWriter gets
    ``write(points, labels, config, event)`` exactly as in the reference."""
    from ..io import KinematicsFileReader

    print("------- AT-TPC Simulation Engine (MI355X) -------")
    print(f"Applying detector effects to kinematics from file: {input_path}")
    reader = KinematicsFileReader(Path(input_path))
    proton_numbers, mass_numbers = reader.proton_numbers, reader.mass_numbers
    nuclei_to_sim = list(indices) if indices is not None else default_indices(len(proton_numbers))
    n_events = reader.n_events
    print(f"Found {n_events} kinematics events in {reader.n_chunks} {reader.chunk_size} event chunks.")
    print(f"Output will be written to {writer.get_directory_name()}.")
    rng = default_rng(seed)
    run_seed = int(rng.integers(0, 1 << 63))
    fused = callable(getattr(writer, "write_rows", None))
    for start in range(0, n_events, batch_size):
        stop = min(n_events, start + batch_size)
        vertices, momenta = reader.read(start, stop)
        if fused:
            offsets, rows, labels, raw_points, _ = simulate_batch_spyral(
                momenta, vertices, proton_numbers, mass_numbers, config, run_seed, nuclei_to_sim,
                first_event=start, response=getattr(writer, "response", None),
            )
            for i in range(stop - start):
                if raw_points[i] == 0:
                    continue  # simulator.py:204-205: decided on the cloud BEFORE the threshold
                writer.write_rows(rows[offsets[i]:offsets[i + 1]], labels[offsets[i]:offsets[i + 1]], start + i,
                                  presorted=True)
            continue
        offsets, points, labels, _ = simulate_batch(
            momenta, vertices, proton_numbers, mass_numbers, config, run_seed, nuclei_to_sim,
            first_event=start,
        )
        for i in range(stop - start):
            lo, hi = offsets[i], offsets[i + 1]
            if hi == lo:
                continue  # simulator.py:204-205
            writer.write(points[lo:hi], labels[lo:hi], config, start + i)
    writer.close()
    print("Done.")
    print("----------------------------------------")
