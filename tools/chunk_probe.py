import sys, time, json
sys.path.insert(0, '/root/repo')
from attpc_engine_amd import _abi, workloads
from attpc_engine_amd.engine import Engine
pipe, cfg, idx = workloads.o16aa()
ctx = _abi.Context(0)
for chunk in (int(sys.argv[1]),):
    eng = Engine(pipe, cfg, idx, context=ctx, chunk_events=chunk)
    for rep in range(5):
        t0 = time.perf_counter()
        st = eng.run(1000000, seed=3, first_event=rep * 1000000)["stats"]
        ctx.lib.attpc_sync(ctx.handle)
        dt = time.perf_counter() - t0
        print(chunk, rep, round(dt * 1e3, 1), {k: st[k] for k in ("ms_scatter", "ms_tracks", "launches_scatter", "launches_tracks", "n_buffer_growths", "n_lds_overflow")}, flush=True)
