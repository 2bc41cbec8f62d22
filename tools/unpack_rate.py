"""Host-side expansion rate of the 16-byte transfer records (attpc_unpack_rows), by thread count."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from attpc_engine_amd import _abi  # noqa: E402

lib = _abi.load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120_000_000
packed = np.zeros(n, dtype=[("tb", np.float64), ("bits", np.uint64)])
packed["bits"] = np.arange(n, dtype=np.uint64)
points = np.empty((n, 3))
labels = np.empty(n, dtype=np.int64)
for threads in (1, 4, 8, 16, 32, 64):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        lib.attpc_unpack_rows(packed.ctypes.data_as(C.c_void_p), n, _abi.dptr(points), _abi.iptr(labels, C.c_int64), threads)
        best = min(best, time.perf_counter() - t0)
    print(f"{threads:3d} threads: {n / best / 1e9:.2f} G rows/s, {n * 48 / best / 1e9:.0f} GB/s of host traffic, {best * 1e3:.0f} ms per {n} rows", flush=True)
