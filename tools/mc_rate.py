import sys, time
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))
from attpc_engine_amd import _abi, workloads
from attpc_engine_amd.engine import Engine
ctx = _abi.Context(0)
pipe, cfg, idx = workloads.o16aa()
cfg.det_params.mc_diffusion = True
eng = Engine(pipe, cfg, idx, context=ctx)
eng.run(2000, seed=1)
st = eng.run(20000, seed=1)["stats"]
print("MC o16aa 20000 events: ms_scatter", st["ms_scatter"], "points/event", st["n_points"] / 20000, "retries", st["n_lds_overflow"], "failed", st["n_failed"], "inconsistent", st["n_inconsistent"])
