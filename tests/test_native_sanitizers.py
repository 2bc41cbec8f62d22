"""CPU sanitizer builds of the native code that runs on the host: the oracle under
AddressSanitizer + UBSan driving the golden-vector tests, and the host-only exports of the product library
(the expansion of the compact transfer records, attpc_unpack_rows / attpc_unpack_spyral_rows and their thread
pools) under ASan + UBSan and under ThreadSanitizer.  Never run on the GPU box (GPU sanitizers are not available
there, and these need none)."""
import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "attpc_engine_amd" / "csrc"


def _gxx():
    exe = shutil.which("g++")
    if exe is None:
        pytest.skip("g++ not available")
    return exe


@pytest.mark.timeout(600)
@pytest.mark.parametrize("name,flags", [
    ("asan_ubsan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]),
    ("tsan", ["-fsanitize=thread"]),
])
def test_unpack_threads_under_sanitizers(tmp_path, name, flags):
    """1 ... 16 expansion threads over ragged slices into exactly-sized heap arrays: no out-of-bounds write, no
    undefined behaviour, no data race; every expanded value checked."""
    exe = tmp_path / f"unpack_{name}"
    cmd = [_gxx(), "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-pthread", f"-I{CSRC}", *flags,
           str(ROOT / "tests" / "native" / "unpack_san.cpp"), str(CSRC / "unpack_host.cpp"), "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    run = subprocess.run([str(exe), "300000"], capture_output=True, text=True, env=env)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    assert "mismatches 0" in run.stdout
    assert "ERROR" not in run.stderr and "WARNING: ThreadSanitizer" not in run.stderr, run.stderr[-4000:]


@pytest.mark.timeout(900)
def test_oracle_golden_under_asan():
    """`make -C oracle asan`, then tests/test_oracle_golden.py in a child interpreter that loads that build (libasan
    preloaded; leak checking off: the interpreter itself is not instrumented)."""
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    made = subprocess.run(["make", "-C", str(ROOT / "oracle"), "asan"], capture_output=True, text=True)
    assert made.returncode == 0, made.stderr[-3000:]
    lib = ROOT / "oracle" / "_san" / "libattpc_oracle_asan.so"
    assert lib.exists()
    libasan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not libasan or not Path(libasan).exists():
        pytest.skip("libasan.so not found")
    env = dict(os.environ, ATTPC_ORACLE_LIBRARY=str(lib), LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               OMP_NUM_THREADS="4")
    run = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_oracle_golden.py"), "-x", "-q",
                          "-p", "no:cacheprovider"], capture_output=True, text=True, env=env, cwd=str(ROOT))
    tail = run.stdout[-3000:] + run.stderr[-3000:]
    assert run.returncode == 0, tail
    assert "passed" in run.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
