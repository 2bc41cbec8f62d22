"""Pins the round-1 barrier fix in the generated code ("Pin the barrier fix").

hipcc puts ``s_waitcnt lgkmcnt(0)`` in front of most ``s_barrier``s by itself, but not in front of the
one at the head of scatter_kernel's event-batch loop: thread 0's ``ds_write`` of the next batch index
could still be queued when the barrier released the other waves.  Every barrier
of the library goes through ``block_sync()`` (explicit wait) since; this test disassembles the shipped
``libattpc_hip.so`` and checks, per ``s_barrier`` and along every path into it, that the wave has waited
for its own LDS operations -- so a bare ``__syncthreads()`` in a new kernel fails here, on the CPU.
``profiles/r02_barrier_isa.md`` keeps the before / after excerpt of the loop-head barrier."""
import subprocess
import sys
from pathlib import Path

import pytest

from tests.isa_tools import barriers_without_lds_wait, disassemble, disassemble_objects, llvm_tool

ROOT = Path(__file__).resolve().parents[1]
LIB = ROOT / "attpc_engine_amd" / "_lib" / "libattpc_hip.so"

pytestmark = pytest.mark.skipif(llvm_tool("llvm-objdump") is None or llvm_tool("llvm-objcopy") is None,
                                reason="ROCm LLVM tools not installed")


def test_every_barrier_waits_for_the_waves_lds_operations():
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as entry

    entry.build()
    functions = disassemble(LIB)
    kernels = {name: insns for name, insns in functions.items() if any(t.startswith("s_barrier") for _, t in insns)}
    # the kernels that synchronise: both scatter builds and instantiations, tracks, lone buckets, Spyral rows
    for needle in ("sc_big14scatter_kernelILb0", "sc_small14scatter_kernelILb0", "sc_big14scatter_kernelILb1",
                   "sc_small14scatter_kernelILb1", "track_kernelILb0", "track_kernelILb1", "lone_bucket_kernel",
                   "spyral_count_kernel", "spyral_write_kernel", "spyral_rows_kernel", "exclusive_scan_kernel"):
        assert any(needle in name for name in kernels), f"{needle} not found among {sorted(kernels)}"
    n_barriers = 0
    for name, insns in kernels.items():
        n_barriers += sum(1 for _, t in insns if t.startswith("s_barrier"))
        bad = barriers_without_lds_wait(insns)
        assert not bad, f"{name}: s_barrier at {[hex(a) for a in bad]} without a preceding s_waitcnt lgkmcnt(0)"
    assert n_barriers >= 60


def test_checker_catches_the_round1_defect(tmp_path):
    """The same source with bare ``__syncthreads()`` (-DATTPC_BARE_BARRIER): the checker must flag the
    scatter kernel -- this is the code that went wrong about once per 20 000 windows in round 1."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("hipcc not installed")
    co = tmp_path / "scatter_bare.co"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "--no-gpu-bundle-output", "-DATTPC_BARE_BARRIER",
                    f"-I{ROOT / 'include'}", f"-I{ROOT / 'attpc_engine_amd' / 'csrc'}", "-c", "-o", str(co),
                    str(ROOT / "attpc_engine_amd" / "csrc" / "scatter_small.hip")], check=True, capture_output=True)
    functions = disassemble_objects([co])
    flagged = {name: barriers_without_lds_wait(insns) for name, insns in functions.items() if "scatter_kernelILb0" in name}
    assert flagged and all(flagged.values()), flagged
