"""Build-time check of the shipped device code: no packed-fp32 arithmetic.

``v_pk_fma_f32`` / ``v_pk_mul_f32`` / ``v_pk_add_f32`` returned wrong values intermittently on gfx950 / ROCm 7.2 when a
SIMD co-executed another queue's MFMA stream (the two-lane refiner steps).  The library is compiled with
``-fno-slp-vectorize`` so that the compiler does not form them; this module disassembles every gfx950 code object of the
built ``libhappypose_amd.so`` and fails if one is there anyway -- a compiler upgrade, a ``float2`` in new code or a lost
flag cannot bring them back silently.  One exception, by name: the exact-fp32 Winograd kernels (``ALLOWED_KERNELS``).  ``python -m happypose_amd.build`` runs it after linking; ``tests/test_abi.py`` asserts it.
"""

from __future__ import annotations

import re
import shutil
import subprocess
import tempfile
from pathlib import Path
from typing import Dict

FORBIDDEN = re.compile(r"\bv_pk_(?:fma|mul|add)_f32\b")
# the exact-fp32 Winograd kernels place packed adds / FMAs by hand (inline asm in csrc/conv_wino.hip; build.PACKED_OK_SOURCES)
ALLOWED_KERNELS = re.compile(r"conv3x3_wino8?_f32")
_SYMBOL = re.compile(r"^[0-9a-f]+ <(.+)>:$")


def _objdump() -> str:
    for cand in ("/opt/rocm/lib/llvm/bin/llvm-objdump", shutil.which("llvm-objdump")):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("llvm-objdump not found (ROCm's LLVM tools)")


def packed_f32_instructions(lib: Path) -> Dict[str, int]:
    """``{kernel symbol: count}`` of the forbidden instructions in every gfx950 code object bundled in ``lib``."""
    objdump = _objdump()
    hits: Dict[str, int] = {}
    with tempfile.TemporaryDirectory() as tmp:
        work = Path(tmp) / lib.name
        shutil.copy2(lib, work)  # --offloading extracts the bundles NEXT to its input
        subprocess.run([objdump, "--offloading", str(work)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = sorted(Path(tmp).glob(lib.name + ".*gfx950*"))
        if not objs:
            raise RuntimeError(f"no gfx950 code object found in {lib}")
        for co in objs:
            text = subprocess.run([objdump, "-d", str(co)], check=True, capture_output=True, text=True).stdout
            sym = "?"
            for line in text.splitlines():
                m = _SYMBOL.match(line)
                if m:
                    sym = m.group(1)
                elif FORBIDDEN.search(line) and not ALLOWED_KERNELS.search(sym):
                    hits[sym] = hits.get(sym, 0) + 1
    return hits


def assert_no_packed_f32(lib: Path) -> None:
    hits = packed_f32_instructions(lib)
    if hits:
        worst = sorted(hits.items(), key=lambda kv: -kv[1])[:8]
        raise RuntimeError(f"{lib.name}: {sum(hits.values())} packed-fp32 instructions (v_pk_fma/mul/add_f32) in {len(hits)} kernels, "
                           f"e.g. {worst} -- see happypose_amd/isa_check.py")


if __name__ == "__main__":
    import sys

    from .build import LIB

    target = Path(sys.argv[1]) if len(sys.argv) > 1 else LIB
    assert_no_packed_f32(target)
    print(f"{target}: no packed-fp32 arithmetic")
