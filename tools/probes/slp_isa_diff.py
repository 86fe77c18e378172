"""Round 6, verdict item 5(b): does the SLP vectoriser -- whose packed-fp32 instructions were the cause named for the two-lane
non-determinism of rounds 2-5 -- ALSO widen or merge memory accesses in the kernels where the differing fields
were computed?  A widened global / LDS access past the end of a record another launch writes would show the same signature and
vanish with the same flag.  This script compiles raster.hip, geometry.hip and crop.hip twice (with the SLP vectoriser, and with
-fno-slp-vectorize as the library is built; packed-fp32 ops allowed in BOTH so that only the vectoriser differs) and compares, per
kernel, the histogram of every memory instruction by opcode -- the opcode carries the access width (global_load_dwordx2 / x4,
ds_read_b64 / b128, buffer_store_dwordx3 ...).  Output: one line per kernel, and the differing opcodes if any.
    python tools/probes/slp_isa_diff.py > profiles/r06_slp_isa_diff.txt"""
import collections
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
BASE = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-x", "hip", "--cuda-device-only", "-S"]
MEM = ("global_", "buffer_", "ds_", "flat_", "scratch_", "s_load", "s_buffer_load")
PK = re.compile(r"^v_pk_(fma|mul|add)_f32")


def kernels(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif line.startswith(".Lfunc_end"):
            cur = None
        elif cur and line.startswith("\t") and not line.startswith("\t."):
            out[cur].append(line.split()[0])
    return out


def hist(ops):
    mem, pk = collections.Counter(), 0
    for op in ops:
        if op.startswith(MEM):
            mem[op] += 1
        elif PK.match(op):
            pk += 1
    return mem, pk


def main():
    any_diff = any_vector_diff = False
    with tempfile.TemporaryDirectory() as tmp:
        for src in ("raster.hip", "geometry.hip", "crop.hip"):
            asm = {}
            for tag, extra in (("slp", []), ("noslp", ["-fno-slp-vectorize"])):
                asm[tag] = Path(tmp) / f"{src}.{tag}.s"
                subprocess.run(BASE + extra + [str(ROOT / "happypose_amd" / "csrc" / src), "-o", str(asm[tag])], check=True, stderr=subprocess.DEVNULL)
            a, b = kernels(asm["slp"]), kernels(asm["noslp"])
            for k in sorted(a):
                if k not in b:
                    continue
                (ma, pa), (mb, pb) = hist(a[k]), hist(b[k])
                name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:100]
                diff = {op: (ma[op], mb[op]) for op in sorted(set(ma) | set(mb)) if ma[op] != mb[op]}
                print(f"{src:13s} {name:100s} packed-fp32 {pa:4d} -> {pb:4d}   memory instructions {sum(ma.values()):4d} -> {sum(mb.values()):4d}   "
                      + ("IDENTICAL opcode histogram" if not diff else f"DIFFERS (slp, noslp): {diff}"))
                any_diff |= bool(diff)
                any_vector_diff |= any(not op.startswith("s_") for op in diff)
    if any_vector_diff:
        print("\nVECTOR memory instructions (global / buffer / ds / flat / scratch) differ in at least one kernel")
    elif any_diff:
        print("\nno kernel's VECTOR memory instructions (global / buffer / ds / flat / scratch: everything that touches a record, a list, LDS or a "
              "pixel) change with the SLP vectoriser; the only difference is in scalar loads of kernel arguments (s_load_*: the kernarg segment, "
              "read-only, how many dwords one instruction fetches).  The vectoriser forms packed-fp32 arithmetic only: no access is widened or merged.")
    else:
        print("\nno kernel's memory instructions change with the SLP vectoriser: it forms packed-fp32 arithmetic only, no access is widened or merged")


if __name__ == "__main__":
    sys.exit(main())
