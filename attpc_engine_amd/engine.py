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

    def run(self, n_events: int, seed: int = 0, first_event: int = 0, fetch: bool = False,
            capacity_per_event: int = 12288) -> dict:
        """Simulate events ``first_event .. first_event + n_events - 1``.

        ``fetch=False``: everything stays device resident (chunk buffers are overwritten);
        only the statistics / checksums come back.  ``fetch=True``: also returns vertex, p4,
        status and the point clouds in CSR form (offsets, points, labels)."""
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
            offsets = np.zeros(n_events + 1, dtype=np.int64)
            points = np.empty((capacity, 3), dtype=np.float64)
            labels = np.empty(capacity, dtype=np.int64)
            out = _abi.CloudOut(capacity, _abi.iptr(offsets, _abi.C.c_int64), _abi.dptr(points),
                                _abi.iptr(labels, _abi.C.c_int64))
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
                "labels": labels[:total], "stats": stats.as_dict()}
