"""Latency of the reference's per-event entry point simulate() (detector/simulator.py:52-115) through the engine:
    python tools/simulate_latency.py [calls]   (run on the GPU box)
One kinematics event per call (host arrays in, the cloud in the reference's dtypes out)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from attpc_engine_amd import workloads  # noqa: E402
from attpc_engine_amd.detector.simulator import simulate  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
pipe, cfg, idx = workloads.o16aa()
rng = np.random.default_rng(5)
from attpc_engine_amd import _abi  # noqa: E402
from attpc_engine_amd.engine import Engine  # noqa: E402

eng = Engine(pipe, cfg, idx, context=_abi.Context(0))  # the kinematics of n events (4-vectors and vertices) from the engine
res = eng.run(n + 5, seed=1, fetch=True)
events = [(res["vertex"][i], res["p4"][i]) for i in range(n + 5)]
z, a = pipe.get_proton_numbers(), pipe.get_mass_numbers()
for vertex, momenta in events[:5]:
    simulate(momenta, vertex, z, a, cfg, rng, idx)
t = []
rows = 0
for vertex, momenta in events[5:]:
    t0 = time.perf_counter()
    pts, lab = simulate(momenta, vertex, z, a, cfg, rng, idx)
    t.append(time.perf_counter() - t0)
    rows += len(pts)
t = np.array(t) * 1e3
print(f"simulate(): {len(t)} calls, median {np.median(t):.3f} ms, mean {t.mean():.3f} ms, 95th {np.percentile(t, 95):.3f} ms, "
      f"{rows / len(t):.0f} cloud rows per call")
