import sys, time, json
sys.path.insert(0, '/root/repo')
from attpc_engine_amd import _abi, workloads
from attpc_engine_amd.engine import Engine
ctx = _abi.Context(0)
for name, n in (("o16aa", 1000000), ("be10dp", 1000000), ("b10chain", 24000)):
    pipe, cfg, idx = workloads.WORKLOADS[name]()
    eng = Engine(pipe, cfg, idx, context=ctx)
    eng.run(min(n, 100000), seed=1)
    for rep in range(2):
        for mode in (0, 1):
            ctx.set_option("track_species_major", mode)
            ctx.lib.attpc_sync(ctx.handle)
            t0 = time.perf_counter()
            st = eng.run(n, seed=1)["stats"]
            ctx.lib.attpc_sync(ctx.handle)
            dt = time.perf_counter() - t0
            print(name, "species_major", mode, "ev/s", round(n / dt), "ms_tracks", round(st["ms_tracks"], 2), "ms_scatter", round(st["ms_scatter"], 2), "charge", st["charge_checksum"], flush=True)
