"""Find events whose clouds differ between two library builds at a launch size where the
differences occur (per-event counts and charge sums only: cheap)."""
import os, subprocess, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]

def child(out, n, first):
    sys.path.insert(0, str(ROOT))
    from attpc_engine_amd import _abi, workloads
    from attpc_engine_amd.engine import Engine
    pipe, cfg, idx = workloads.o16aa()
    eng = Engine(pipe, cfg, idx, context=_abi.Context(0))
    res = eng.run(n, seed=1, first_event=first, fetch=True, capacity_per_event=8000)
    off, pts = res["offsets"], res["points"]
    cnt = np.diff(off)
    idx = np.minimum(off[:-1], max(len(pts) - 1, 0))
    qsum = np.where(cnt > 0, np.add.reduceat(pts[:, 2], idx), 0.0) if len(pts) else np.zeros(n)
    key = (np.floor(pts[:, 1]).astype(np.int64) << 14) | pts[:, 0].astype(np.int64)
    ksum = np.where(cnt > 0, np.add.reduceat(key, idx), 0)
    np.savez(out, cnt=cnt, qsum=qsum, ksum=ksum, npts=res["stats"]["n_points"])
    print("child done", out, res["stats"]["n_points"], flush=True)

if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4])); sys.exit(0)
    libs = sys.argv[1:3]; n = int(sys.argv[3])
    outs = []
    for i, lib in enumerate(libs):
        out = f"/tmp/d2_{i}.npz"
        env = dict(os.environ, ATTPC_HIP_LIBRARY=str((ROOT / lib).resolve()))
        subprocess.run([sys.executable, __file__, "--child", out, str(n), "0"], env=env, check=True)
        outs.append(np.load(out))
    a, b = outs
    bad = np.nonzero((a["cnt"] != b["cnt"]) | (a["qsum"] != b["qsum"]) | (a["ksum"] != b["ksum"]))[0]
    print("events differing:", len(bad), "of", n, flush=True)
    for e in bad[:25]:
        print("event", e, "count", a["cnt"][e], b["cnt"][e], "charge", a["qsum"][e], b["qsum"][e], "dq", b["qsum"][e] - a["qsum"][e], "dkeysum", b["ksum"][e] - a["ksum"][e])
