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
TORCH_LIB = LIB_DIR / "libhappypose_amd_torch.so"
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
    build_torch_library(force=force, run=run)
    return LIB


def build_torch_library(force: bool = False, run=subprocess.check_call) -> Path:
    """``csrc/torch_library.cpp`` -> ``lib/libhappypose_amd_torch.so``: the compiled TORCH_LIBRARY over the C ABI (host code
    only; g++ against the installed torch's headers, linked to libhappypose_amd.so through ``$ORIGIN``)."""
    src = CSRC / "torch_library.cpp"
    if not (force or _stale(TORCH_LIB, [src, PKG.parent / "include" / "happypose_amd.h", LIB])):
        return TORCH_LIB
    import torch
    from torch.utils import cpp_extension as ce

    tlib = Path(torch.__file__).resolve().parent / "lib"
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           *[f"-isystem{p}" for p in ce.include_paths("cuda")], str(src), "-o", str(TORCH_LIB),
           f"-L{LIB_DIR}", "-lhappypose_amd", f"-L{tlib}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch",
           "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    run(cmd)
    return TORCH_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
