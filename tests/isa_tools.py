"""Disassembly helpers for the tests that pin properties of the generated gfx950 code
(test infrastructure only).  The library's ``.hip_fatbin`` section is a sequence of clang offload
bundles (one per translation unit); each holds one gfx950 code object."""
from __future__ import annotations

import shutil
import struct
import subprocess
import tempfile
from pathlib import Path

LLVM_BIN = Path("/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def llvm_tool(name: str) -> Path | None:
    for cand in (LLVM_BIN / name, Path("/opt/rocm/llvm/bin") / name):
        if cand.exists():
            return cand
    found = shutil.which(name)
    return Path(found) if found else None


def device_code_objects(library: Path, workdir: Path) -> list[Path]:
    """Extract every gfx950 code object of ``library`` into ``workdir``."""
    objcopy = llvm_tool("llvm-objcopy")
    assert objcopy is not None, "llvm-objcopy not found"
    fat = workdir / "fat.bin"
    subprocess.run([str(objcopy), f"--dump-section=.hip_fatbin={fat}", str(library)], check=True, capture_output=True)
    blob = fat.read_bytes()
    out: list[Path] = []
    pos = blob.find(MAGIC)
    while pos >= 0:
        (n_entries,) = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        cursor = pos + len(MAGIC) + 8
        for _ in range(n_entries):
            offset, size, triple_len = struct.unpack_from("<QQQ", blob, cursor)
            triple = blob[cursor + 24: cursor + 24 + triple_len].decode()
            cursor += 24 + triple_len
            if "gfx950" in triple and size:
                path = workdir / f"code_{len(out)}.co"
                path.write_bytes(blob[pos + offset: pos + offset + size])
                out.append(path)
        pos = blob.find(MAGIC, pos + len(MAGIC))
    return out


def disassemble_objects(objects: list[Path]) -> dict[str, list[tuple[int, str]]]:
    """{symbol: [(address, instruction text)]} for every function of the given code objects."""
    objdump = llvm_tool("llvm-objdump")
    assert objdump is not None, "llvm-objdump not found"
    functions: dict[str, list[tuple[int, str]]] = {}
    for co in objects:
        text = subprocess.run([str(objdump), "-d", "--no-show-raw-insn", str(co)], check=True,
                              capture_output=True, text=True).stdout
        current = None
        for line in text.splitlines():
            stripped = line.strip()
            if stripped.endswith(">:") and "<" in stripped:
                current = stripped[stripped.index("<") + 1: -2]
                functions.setdefault(current, [])
            elif current is not None and "//" in stripped:
                # "s_waitcnt lgkmcnt(0)    // 000000001234: ..."
                insn, _, comment = stripped.partition("//")
                address = int(comment.strip().split(":")[0], 16)
                functions[current].append((address, insn.strip()))
    return functions


def disassemble(library: Path) -> dict[str, list[tuple[int, str]]]:
    """Every function of every gfx950 code object inside a shared library."""
    with tempfile.TemporaryDirectory() as tmp:
        return disassemble_objects(device_code_objects(library, Path(tmp)))


_LGKM_OPS = ("ds_", "s_load_", "s_buffer_load_", "s_memtime", "s_memrealtime", "flat_", "s_sendmsg", "s_dcache",
             "s_atomic_", "s_scratch_", "s_store_", "s_buffer_store_")


def barriers_without_lds_wait(insns: list[tuple[int, str]]) -> list[int]:
    """Addresses of the ``s_barrier`` instructions that some path can reach while one of the wave's own
    LDS / scalar-memory operations (``lgkmcnt``) may still be outstanding: walking back from the barrier
    there must be an ``s_waitcnt ... lgkmcnt(0)`` before any lgkm-counted instruction and before any
    point where another path joins (branch target) or the block starts."""
    targets = set()
    for i, (addr, text) in enumerate(insns):
        op, _, arg = text.partition(" ")
        if op.startswith(("s_cbranch", "s_branch")) and arg.strip().lstrip("-").isdigit():
            offset = int(arg)
            if offset >= 0x8000:
                offset -= 0x10000
            targets.add(addr + 4 + 4 * offset)
    bad = []
    for i, (addr, text) in enumerate(insns):
        if not text.startswith("s_barrier"):
            continue
        ok = False
        j = i - 1
        while j >= 0:
            a, t = insns[j]
            if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                ok = True
                break
            if t.startswith(_LGKM_OPS) or t.startswith(("s_branch", "s_endpgm", "s_setpc", "s_swappc", "s_barrier")):
                break
            if insns[j + 1][0] in targets:  # another path joins between instruction j and the barrier
                break
            j -= 1
        if not ok:
            bad.append(addr)
    return bad
