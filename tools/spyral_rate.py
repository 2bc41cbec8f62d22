"""Delivered rates of the fused paths after a device-resident run (the sequence of bench.py's delivered leg)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from attpc_engine_amd import _abi, workloads  # noqa: E402
from attpc_engine_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
big = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
pipe, cfg, idx = workloads.o16aa()
eng = Engine(pipe, cfg, idx, context=_abi.Context(0))
t0 = time.perf_counter()
eng.run(big, seed=3)
print("resident", big, time.perf_counter() - t0, flush=True)
for name in ("cloud", "spyral"):
    for rep in range(3):
        t0 = time.perf_counter()
        if name == "cloud":
            res = eng.run(n, seed=3, first_event=1000 * rep, fetch=True, pinned=False, reuse_buffers=True, capacity_per_event=9200)
        else:
            res = eng.run_spyral(n, seed=3, first_event=1000 * rep, pinned=True, reuse_buffers=True, capacity_per_event=9200)
        dt = time.perf_counter() - t0
        print(name, rep, n / dt, "events/s", dt, "s", res["offsets"][-1] / n, "rows/event", flush=True)
    eng._out_cache = None
    del res
