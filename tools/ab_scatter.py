"""Same-session A/B of library builds (run on the GPU box):
    python tools/ab_scatter.py libA.so libB.so ... [--events N] [--workloads o16aa,be10dp]
Each library runs in its own child process (ATTPC_HIP_LIBRARY), twice in alternation, and prints
events/s, kernel milliseconds and the charge / key checksums (which must not change)."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def child(workloads_csv: str, n: int) -> None:
    sys.path.insert(0, str(ROOT))
    import time

    from attpc_engine_amd import _abi, workloads
    from attpc_engine_amd.engine import Engine

    ctx = _abi.Context(0)
    if os.environ.get("ATTPC_AB_VARIANT"):  # 1 = small, 2 = big scatter build
        ctx.set_option("scatter_variant", int(os.environ["ATTPC_AB_VARIANT"]))
    out = {}
    for name in workloads_csv.split(","):
        pipe, cfg, idx = workloads.WORKLOADS[name]()
        eng = Engine(pipe, cfg, idx, context=ctx, chunk_events=int(os.environ.get("ATTPC_AB_CHUNK", "0")) or None)
        eng.run(min(n, 20000), seed=1)
        ctx.lib.attpc_sync(ctx.handle)
        t0 = time.perf_counter()
        st = eng.run(n, seed=1)["stats"]
        ctx.lib.attpc_sync(ctx.handle)
        dt = time.perf_counter() - t0
        out[name] = {"ev_s": round(n / dt), "ms_tracks": round(st["ms_tracks"], 2), "ms_scatter": round(st["ms_scatter"], 2),
                     "points": st["n_points"], "samples": st["n_track_samples"], "charge": st["charge_checksum"], "keys": st["key_checksum"],
                     "failed": st["n_failed"], "retries": st["n_lds_overflow"], "inconsistent": st["n_inconsistent"]}
        if "phase_cycles" in st:
            out[name]["phase"] = st["phase_cycles"]
    print(json.dumps(out))


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--child":
        child(args[1], int(args[2]))
        sys.exit(0)
    n, wl, libs = 200_000, "o16aa", []
    i = 0
    while i < len(args):
        if args[i] == "--events":
            n = int(args[i + 1]); i += 2
        elif args[i] == "--workloads":
            wl = args[i + 1]; i += 2
        else:
            libs.append(args[i]); i += 1
    for rep in range(2):
        for lib in libs:
            env = dict(os.environ, ATTPC_HIP_LIBRARY=str((ROOT / lib).resolve()))
            proc = subprocess.run([sys.executable, __file__, "--child", wl, str(n)], env=env, capture_output=True, text=True)
            print(rep, lib, proc.stdout.strip() or proc.stderr[-2000:], flush=True)
            diag = [line for line in proc.stderr.splitlines() if line.startswith("[attpc")]
            if diag:  # phase-timer build: the last chunk's counters
                print("\n".join(diag[-5:]), flush=True)
