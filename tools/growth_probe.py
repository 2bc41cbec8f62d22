import sys
sys.path.insert(0, '/root/repo')
from attpc_engine_amd import _abi, workloads
from attpc_engine_amd.engine import Engine
ctx = _abi.Context(0)
pipe, cfg, idx = workloads.o16aa()
eng = Engine(pipe, cfg, idx, context=ctx)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
for step in range(6):
    st = eng.run(n, seed=3, first_event=step * n)["stats"]
    print(step, "growths", st.get("n_buffer_growths"), "device_bytes", st.get("device_bytes"), "launches", st.get("launches_tracks"), st.get("launches_scatter"), "points", st["n_points"], flush=True)
