"""Host expansion rate of the compact transfer records (run anywhere; no GPU needed):
    python tools/unpack_rate.py [rows] -> rows/s of attpc_unpack_rows (16-byte) and attpc_unpack_rows8 (8-byte, jitter
    regenerated) for several thread counts."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from attpc_engine_amd import _abi  # noqa: E402

lib = _abi.load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000_000
rng = np.random.default_rng(1)
packed16 = np.zeros(n, dtype=[("tb", np.float64), ("bits", np.uint64)])
packed16["bits"] = rng.integers(0, 1 << 62, size=n, dtype=np.uint64)
packed8 = rng.integers(0, 1 << 62, size=n, dtype=np.uint64)
n_events = n // 7300
offsets = np.linspace(0, n, n_events + 1).astype(np.int64)
points = np.empty((n, 3))
labels = np.empty(n, dtype=np.int64)
points[:] = 0.0
labels[:] = 0
for threads in (8, 16, 32, 64):
    for name in ("rows16", "rows8"):
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            if name == "rows16":
                lib.attpc_unpack_rows(packed16.ctypes.data_as(C.c_void_p), n, _abi.dptr(points), _abi.iptr(labels, C.c_int64), threads)
            else:
                lib.attpc_unpack_rows8(packed8.ctypes.data_as(C.c_void_p), n, _abi.iptr(offsets, C.c_int64), n_events, 3, 0,
                                       _abi.dptr(points), _abi.iptr(labels, C.c_int64), threads)
            best = max(best, n / (time.perf_counter() - t0))
        print(f"{name} threads {threads}: {best / 1e9:.2f}e9 rows/s = {best / 7300:.3g} events/s of 7300 rows", flush=True)
