"""Measure the non-headline BASELINE configs and the delivered-to-host rate (run on the GPU box):
    python tools/measure_configs.py > gpurun_out/other_configs.json
"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np

from attpc_engine_amd import _abi, workloads
from attpc_engine_amd.engine import Engine

ctx = _abi.Context(0)
out = {}

# configs[0]: 1k-event 12C(p,p) kinematics only (GPU vs the per-call API)
pipe, _, _ = workloads.c12pp(seed=1)
pipe._ctx = ctx
pipe.run_many(1000)
t0 = time.perf_counter()
v, p4 = pipe.run_many(1000)
dt = time.perf_counter() - t0
t0 = time.perf_counter()
big = pipe.run_many(4_000_000, return_status=True)
dt_big = time.perf_counter() - t0
out["c12pp_1k"] = {"events": 1000, "seconds": dt, "events_per_s_incl_d2h": 1000 / dt,
                   "bulk_4e6_events_per_s_incl_d2h": 4e6 / dt_big,
                   "conservation_max_abs_MeV": float(np.abs(p4[:, 0] + p4[:, 1] - p4[:, 2] - p4[:, 3]).max())}

for name, n in (("be10dp", 100_000), ("o16aa", 200_000), ("b10chain", 50_000)):
    pipe, cfg, idx = workloads.WORKLOADS[name]()
    eng = Engine(pipe, cfg, idx, context=ctx)
    eng.run(min(n, 20000), seed=1)
    ctx.lib.attpc_sync(ctx.handle)
    t0 = time.perf_counter()
    st = eng.run(n, seed=1)["stats"]
    ctx.lib.attpc_sync(ctx.handle)
    dt = time.perf_counter() - t0
    out[name] = {"events": n, "events_per_s_device_resident": n / dt, "points_per_event": st["n_points"] / n,
                 "samples_per_event": st["n_track_samples"] / n, "failed": st["n_failed"],
                 "window_retries_per_event": st["n_lds_overflow"] / n,
                 "ms": {k: st[k] for k in ("ms_kinematics", "ms_tracks", "ms_scatter")}}
    if name == "o16aa":
        m = 50_000
        t0 = time.perf_counter()
        res = eng.run(m, seed=1, fetch=True)
        dt = time.perf_counter() - t0
        out["o16aa_delivered_to_host"] = {"events": m, "events_per_s_incl_d2h_and_csr_assembly": m / dt,
                                          "GB_copied": res["points"].nbytes / 1e9 + res["labels"].nbytes / 1e9}
        del res
        eng.run_spyral(2000, seed=1)
        t0 = time.perf_counter()
        res = eng.run_spyral(m, seed=1)
        dt = time.perf_counter() - t0
        out["o16aa_spyral_rows_delivered_to_host"] = {
            "events": m, "events_per_s_incl_response_threshold_d2h": m / dt,
            "rows_per_event_after_threshold": len(res["rows"]) / m,
            "GB_copied": res["rows"].nbytes / 1e9 + res["labels"].nbytes / 1e9}
        del res
print(json.dumps(out, indent=1))
