"""In-tree build of the HIP library: ``python -m happypose_amd.build``.

Compiles every source under ``happypose_amd/csrc`` for gfx950 with hipcc into
``happypose_amd/lib/libhappypose_amd.so`` (git-ignored; it travels to the GPU box
with the source snapshot).  hipcc cross-compiles without a GPU present.
"""

from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB_DIR = PKG / "lib"
LIB = LIB_DIR / "libhappypose_amd.so"
OBJ_DIR = PKG / "build_obj"
SOURCES = ["api.cpp", "net.cpp", "raster.hip", "geometry.hip", "crop.hip", "conv.hip", "conv_patch.hip", "conv_wino.hip", "conv_wino2.hip", "conv_split.hip", "conv_pp.hip", "conv_igemm_split.hip", "conv_stem_split.hip", "conv_stem7.hip", "conv_f16.hip", "pool_head.hip", "icp.hip", "mbconv.hip", "mbconv_front.hip", "probe.hip", "detect.hip"]
# -fno-slp-vectorize: the SLP vectoriser turns adjacent scalar fp32 operations into packed-fp32 instructions (v_pk_fma_f32 /
# v_pk_mul_f32 / v_pk_add_f32 with op_sel operand swizzles).  On gfx950 / ROCm 7.2 those intermittently returned WRONG
# values when the SIMD co-executed another queue's MFMA stream (two-lane steps: the rasteriser's set-up planes beside the
# other lane's conv kernels; 100 - 160 of 1200 steps differed, 0 of 2400 with this flag -- DESIGN.md "the co-scheduling
# non-determinism", tools/probes/two_lane_repro.py).  Explicit vector types (float4 epilogues) are not affected by the flag.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "-fno-slp-vectorize",
         "-fgpu-rdc" if False else "-fno-gpu-rdc"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (Path(cand).exists() or cand == "hipcc"):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    hipcc = _hipcc()
    OBJ_DIR.mkdir(exist_ok=True)
    LIB_DIR.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.h")) + [PKG.parent / "include" / "happypose_amd.h"]
    jobs = []
    for src in SOURCES:
        obj = OBJ_DIR / (src.replace(".", "_") + ".o")
        if force or _stale(obj, [CSRC / src, *headers]):
            cmd = [hipcc, *FLAGS, "-x", "hip", "-c", str(CSRC / src), "-o", str(obj)]
            jobs.append(cmd)
    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [str(OBJ_DIR / (s.replace(".", "_") + ".o")) for s in SOURCES]
    if force or jobs or not LIB.exists():
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
