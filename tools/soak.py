import sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))
from attpc_engine_amd import _abi, workloads
from attpc_engine_amd.engine import Engine
ctx = _abi.Context(0)
for name, n, reps in (("o16aa", 500000, 6), ("be10dp", 500000, 4), ("b10chain", 8000, 3)):
    pipe, cfg, idx = workloads.WORKLOADS[name]()
    eng = Engine(pipe, cfg, idx, context=ctx)
    ref = None
    for r in range(reps):
        st = eng.run(n, seed=9)["stats"]
        key = (st["n_points"], st["charge_checksum"], st["key_checksum"], st["n_failed"], st["n_inconsistent"])
        ref = ref or key
        assert key == ref, (name, r, key, ref)
    print(name, n, "x", reps, "identical:", ref, flush=True)
