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
SOURCES = ["api.cpp", "debug.cpp", "net.cpp", "raster.hip", "geometry.hip", "crop.hip", "conv.hip", "conv_patch.hip", "conv_wino.hip", "conv_split.hip", "conv_pp.hip", "conv_igemm_split.hip", "conv_stem_split.hip", "conv_stem7.hip", "conv_f16.hip", "pool_head.hip", "icp.hip", "mbconv.hip", "mbconv_front.hip", "probe.hip", "detect.hip"]
# -fno-slp-vectorize: the SLP vectoriser turns adjacent scalar fp32 operations into packed-fp32 instructions (v_pk_fma_f32 /
# v_pk_mul_f32 / v_pk_add_f32 with op_sel operand swizzles).  On gfx950 / ROCm 7.2 those intermittently returned WRONG
# values when the SIMD co-executed another queue's MFMA stream (two-lane steps: the rasteriser's set-up planes beside the
# other lane's conv kernels; 100 - 160 of 1200 steps differed, 0 of 2400 with this flag -- DESIGN.md "the co-scheduling
# non-determinism", tools/probes/two_lane_repro.py).  Explicit vector types (float4 epilogues) are not affected by the flag.
# -target-feature -packed-fp32-ops (round 6): the explicit vector types kept ~27 k v_pk_*_f32 in the conv kernels' epilogues and
# staging code.  They never misbehaved in the stress tests, but nothing separated them in principle from the faulty ones, so
# the device compiler is told the target has no packed-fp32 arithmetic at all: float4 expressions legalise to scalar FMAs
# (measured neutral: CHANGELOG round 6), and `isa_check.assert_no_packed_f32` -- run below after every link and by
# tests/test_abi.py -- fails the build if a single one is left.  The host pass prints "not a recognized feature" for it: filtered.
# -fvisibility=hidden: the dynamic symbols are the declarations of include/happypose_amd.h (visibility pragma there), nothing else.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "-fno-slp-vectorize",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-fvisibility=hidden", "-fno-gpu-rdc"]
_HOST_NOISE = "is not a recognized feature for this target"
_NO_PK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# conv_wino.hip (the exact-fp32 Winograd kernels: hp_net_set_conv_algo(WINOGRAD), the overflow guard's fallback, bench.py's
# exact_fp32_kernels pass) places v_pk_add_f32 / v_pk_fma_f32 BY HAND in its input transform (inline asm: beside fp32 MFMAs every
# VALU instruction costs its issue time, and the scalar spelling measured slower); it is compiled with the target's packed ops
# and its kernels are the one exception isa_check allows (ALLOWED_KERNELS there).
PACKED_OK_SOURCES = {"conv_wino.hip"}

# A/B builds: HP_BUILD_VARIANT=<name> compiles into build_obj_<name>/ and lib_<name>/ (load with HAPPYPOSE_AMD_LIB=...), with
# HP_BUILD_DROP_FLAGS / HP_BUILD_ADD_FLAGS (space separated) applied to FLAGS; the ISA check is skipped for variants.
VARIANT = os.environ.get("HP_BUILD_VARIANT", "")
if VARIANT:
    OBJ_DIR = PKG / f"build_obj_{VARIANT}"
    LIB_DIR = PKG / f"lib_{VARIANT}"
    LIB = LIB_DIR / "libhappypose_amd.so"
    TORCH_LIB = LIB_DIR / "libhappypose_amd_torch.so"
    _drop = os.environ.get("HP_BUILD_DROP_FLAGS", "").split()
    FLAGS = [f for f in FLAGS if f not in _drop] + os.environ.get("HP_BUILD_ADD_FLAGS", "").split()
    if "-packed-fp32-ops" in _drop:  # its -Xclang -target-feature -Xclang prefix goes with it
        FLAGS = [f for f in FLAGS if f not in ("-Xclang", "-target-feature")]


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
    if LIB.exists() and (CSRC / "exports.map").stat().st_mtime > LIB.stat().st_mtime:
        force_link = True
    else:
        force_link = False
    jobs = []
    for src in SOURCES:
        obj = OBJ_DIR / (src.replace(".", "_") + ".o")
        if force or _stale(obj, [CSRC / src, *headers]):
            flags = FLAGS
            if src in PACKED_OK_SOURCES and "-packed-fp32-ops" in flags:
                i = flags.index("-packed-fp32-ops")
                flags = flags[:i - 3] + flags[i + 1:]
            cmd = [hipcc, *flags, "-x", "hip", "-c", str(CSRC / src), "-o", str(obj)]
            jobs.append(cmd)
    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        p = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        err = "\n".join(l for l in p.stderr.splitlines() if _HOST_NOISE not in l)
        if err.strip():
            print(err, file=sys.stderr, flush=True)
        if p.returncode:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [str(OBJ_DIR / (s.replace(".", "_") + ".o")) for s in SOURCES]
    if force or jobs or force_link or not LIB.exists():
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,--version-script={CSRC / 'exports.map'}", "-o", str(LIB), *objs])
        if not VARIANT:
            from .isa_check import assert_no_packed_f32

            assert_no_packed_f32(LIB)
    if not VARIANT:
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
