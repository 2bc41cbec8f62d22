"""Delivered rate against the delivery chunk size (run on the GPU box)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from attpc_engine_amd import _abi, workloads  # noqa: E402
from attpc_engine_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
pipe, cfg, idx = workloads.o16aa()
ctx = _abi.Context(0)
eng = Engine(pipe, cfg, idx, context=ctx)
eng.run(1000000, seed=3)
SWEEP = ((8192, 0, 2), (8192, 8, 2), (8192, 16, 2), (8192, 32, 2), (16384, 16, 2), (4096, 16, 2), (8192, 16, 1))  # 0 = the library's default
if len(sys.argv) > 2:  # chunk sizes only: python tools/deliver_sweep.py N 2048,4096,...
    SWEEP = tuple((int(c), 0, 2) for c in sys.argv[2].split(","))
for chunk, threads, compact in SWEEP:
    ctx.set_option("deliver_chunk_events", chunk)
    ctx.set_option("unpack_threads", threads)
    ctx.set_option("compact_transfer", compact)
    for name in ("cloud", "spyral"):
        best = 0.0
        for rep in range(3):
            t0 = time.perf_counter()
            if name == "cloud":
                res = eng.run(n, seed=3, first_event=1000 * rep, fetch=True, reuse_buffers=True, capacity_per_event=9200)
            else:
                res = eng.run_spyral(n, seed=3, first_event=1000 * rep, reuse_buffers=True, capacity_per_event=5500)
            dt = time.perf_counter() - t0
            if rep:
                best = max(best, n / dt)
        eng._out_cache = None
        del res
        print("chunk", chunk, "unpack_threads", threads, "compact_transfer", compact, name, round(best), "events/s", flush=True)
