"""Fused driver: kinematics + detector simulation without the file in between.

The reference couples its two stages through an HDF5 file (``run_kinematics_pipeline``
then ``run_simulation``, reference kinematics/pipeline.py:429-495 and
detector/simulator.py:118-210).  ``Engine`` runs the same two operators back to back on
one GPU with the kinematics staying in HBM (``attpc_sim_run``); it is what the benchmark
and the multi-GPU sharding drive.  Events are identified by a *global* event id, so any
split of an id range over calls, chunks, processes or GPUs produces the same events.
"""
from __future__ import annotations

import numpy as np

from . import _abi, nuclear_map
from .detector.luts import build_det_desc, build_layout, species_for
from .detector.simulator import default_indices


class Engine:
    def __init__(self, pipeline, config, indices: list[int] | None = None,
                 context: _abi.Context | None = None, chunk_events: int | None = None,
                 ode_substeps: int = 1):
        self.pipeline = pipeline
        self.config = config
        self.ctx = context or _abi.default_context()
        self.z = pipeline.get_proton_numbers()
        self.a = pipeline.get_mass_numbers()
        self.n_rows = len(self.z)
        self.indices = list(indices) if indices is not None else default_indices(self.n_rows)
        ctx = self.ctx
        kin, keep_k = pipeline.device_desc()
        ctx.check(ctx.lib.attpc_kin_configure(ctx.handle, kin), "attpc_kin_configure")
        ctx._kin_owner = id(pipeline)
        self.species = species_for(self.z, self.a, self.indices)
        nuclei = [nuclear_map.get_data(z, a) for z, a in self.species]
        det, keep_d = build_det_desc(config, nuclei, ode_substeps=ode_substeps)
        ctx.check(ctx.lib.attpc_det_configure(ctx.handle, det), "attpc_det_configure")
        ctx._det_token = None
        self.layout = build_layout(self.z, self.a, self.indices, self.species)
        if chunk_events:
            ctx.check(ctx.lib.attpc_set_chunk_events(ctx.handle, int(chunk_events)), "attpc_set_chunk_events")
        del keep_k, keep_d

    def _out_arrays(self, n_events: int, capacity: int, width: int, pinned: bool, reuse: bool):
        """Output arrays of a delivered run: (offsets, rows [capacity, width], labels, event_points).
        ``pinned``: page-locked memory (PCIe-rate copies).  ``reuse``: keep them for the next call of
        the same shape -- the previous call's arrays are then overwritten."""
        key = (n_events, capacity, width, pinned)
        cached = getattr(self, "_out_cache", None)
        if reuse and cached is not None and cached[0] == key:
            return cached[1]
        make = self.ctx.pinned_empty if pinned else (lambda shape, dtype: np.empty(shape, dtype=dtype))
        arrays = (np.zeros(n_events + 1, dtype=np.int64), make((capacity, width), np.float64),
                  make((capacity,), np.int64), np.zeros(n_events, dtype=np.int64))
        self._out_cache = (key, arrays) if reuse else None
        return arrays

    def hint_next(self, n_events: int, seed: int = 0, first_event: int = 0) -> None:
        """Announce the ``run`` / ``run_spyral`` call after the next one (``attpc_sim_hint_next``): the next call then
        queues that call's first kinematics + track batch behind its own last scatter launches.  A scheduling hint
        only -- results never depend on it; ``n_events = 0`` withdraws it."""
        ctx = self.ctx
        ctx.check(ctx.lib.attpc_sim_hint_next(ctx.handle, int(seed), int(first_event), int(n_events), self.layout),
                  "attpc_sim_hint_next")

    def run(self, n_events: int, seed: int = 0, first_event: int = 0, fetch: bool = False,
            capacity_per_event: int = 12288, pinned: bool = False, reuse_buffers: bool = False) -> dict:
        """Simulate events ``first_event .. first_event + n_events - 1``.

        ``fetch=False``: everything stays device resident (chunk buffers are overwritten);
        only the statistics / checksums come back.  ``fetch=True``: also returns vertex, p4,
        status and the point clouds in CSR form (offsets, points, labels); ``pinned`` puts the cloud
        arrays in page-locked host memory (the copy is PCIe bound), ``reuse_buffers`` reuses them
        from call to call (a consumer that is done with one batch before it asks for the next).
        Raises ``DataLossError`` if an event lost charge (``n_failed`` / ``n_inconsistent``)."""
        ctx = self.ctx
        stats = _abi.RunStats()
        if not fetch:
            ctx.check(
                ctx.lib.attpc_sim_run(ctx.handle, int(seed), int(first_event), int(n_events), self.layout,
                                      None, None, None, None, stats),
                "attpc_sim_run",
            )
            return {"stats": stats.as_dict()}
        p4 = np.empty((n_events, self.n_rows, 4), dtype=np.float64)
        vertex = np.empty((n_events, 3), dtype=np.float64)
        status = np.empty(n_events, dtype=np.int32)
        capacity = max(4096, int(capacity_per_event) * int(n_events))
        while True:
            offsets, points, labels, event_points = self._out_arrays(n_events, capacity, 3, pinned, reuse_buffers)
            out = _abi.CloudOut(capacity, _abi.iptr(offsets, _abi.C.c_int64), _abi.dptr(points),
                                _abi.iptr(labels, _abi.C.c_int64), _abi.iptr(event_points, _abi.C.c_int64))
            rc = ctx.lib.attpc_sim_run(ctx.handle, int(seed), int(first_event), int(n_events), self.layout,
                                       _abi.dptr(p4), _abi.dptr(vertex), _abi.iptr(status, _abi.C.c_int32),
                                       out, stats)
            if rc == _abi.E_CAPACITY:
                capacity = int(stats.n_points) + 4096
                continue
            ctx.check(rc, "attpc_sim_run")
            break
        total = int(offsets[-1])
        return {"vertex": vertex, "p4": p4, "status": status, "offsets": offsets, "points": points[:total],
                "labels": labels[:total], "event_points": event_points, "stats": stats.as_dict()}

    # ---------------------------------------------------------------- Spyral rows on the device
    def configure_spyral(self, config=None) -> None:
        """Upload what SpyralWriter needs (reference writer.py:164-181, 220-234): the GET response of
        the electronics, pad centres / sizes, ADC threshold and time-bucket edges."""
        from .detector.simulator import configure_spyral

        configure_spyral(config or self.config, self.ctx)
        self._spyral_configured = True

    def run_spyral(self, n_events: int, seed: int = 0, first_event: int = 0, capacity_per_event: int = 6144,
                   pinned: bool = False, reuse_buffers: bool = False) -> dict:
        """Fused kinematics + detector + (on the device) GET response, ADC threshold, Spyral row
        conversion and z-sort.  Returns rows [P', 8] (x mm, y mm, z mm, amplitude, integral, pad, tb,
        pad scale) in CSR form, the rows of every event in ascending z (reference writer.py:232-238), and
        ``event_points`` [n] = cloud rows of every event before the threshold (an event is "empty" for
        the writer only if that is 0, simulator.py:204-205)."""
        if not getattr(self, "_spyral_configured", False):
            self.configure_spyral()
        ctx = self.ctx
        stats = _abi.RunStats()
        p4 = np.empty((n_events, self.n_rows, 4), dtype=np.float64)
        vertex = np.empty((n_events, 3), dtype=np.float64)
        status = np.empty(n_events, dtype=np.int32)
        capacity = max(4096, int(capacity_per_event) * int(n_events))
        while True:
            offsets, rows, labels, event_points = self._out_arrays(n_events, capacity, 8, pinned, reuse_buffers)
            out = _abi.CloudOut(capacity, _abi.iptr(offsets, _abi.C.c_int64), _abi.dptr(rows),
                                _abi.iptr(labels, _abi.C.c_int64), _abi.iptr(event_points, _abi.C.c_int64))
            rc = ctx.lib.attpc_sim_run_spyral(ctx.handle, int(seed), int(first_event), int(n_events), self.layout,
                                              _abi.dptr(p4), _abi.dptr(vertex), _abi.iptr(status, _abi.C.c_int32),
                                              out, stats)
            if rc == _abi.E_CAPACITY:
                capacity = int(stats.n_points) + 4096
                continue
            ctx.check(rc, "attpc_sim_run_spyral")
            break
        total = int(offsets[-1])
        return {"vertex": vertex, "p4": p4, "status": status, "offsets": offsets, "rows": rows[:total],
                "labels": labels[:total], "event_points": event_points, "stats": stats.as_dict()}


def run_fused(pipeline, config, writer, n_events: int, indices: list[int] | None = None, seed: int | None = None,
              batch_size: int = 65536, context: _abi.Context | None = None) -> None:
    """run_kinematics_pipeline + run_simulation + SpyralWriter without the kinematics file and with
    the response / threshold / row conversion / z-sort done on the GPU: per event with a non-empty
    cloud (before the threshold, as simulator.py:204-205 decides it -- an event whose rows all fall
    below the ADC threshold is still written, with 0 rows, and counts towards the file roll-over exactly
    as in run_simulation + SpyralWriter.write), in event order,
    ``writer.write_rows(rows, labels, event_number, presorted=True)``, then ``close()``."""
    engine = Engine(pipeline, config, indices, context=context)
    engine.configure_spyral(config)
    seed = pipeline.seed if seed is None else int(seed)
    for start in range(0, n_events, batch_size):
        n = min(batch_size, n_events - start)
        res = engine.run_spyral(n, seed=seed, first_event=start)
        off, raw = res["offsets"], res["event_points"]
        for i in range(n):
            if raw[i] > 0:
                writer.write_rows(res["rows"][off[i]:off[i + 1]], res["labels"][off[i]:off[i + 1]], start + i,
                                  presorted=True)
    writer.close()
