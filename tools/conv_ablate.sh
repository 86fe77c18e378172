#!/bin/bash
# Ablation of the conv K loop (run on the GPU box): builds variants of the library with parts
# of the loop compiled out and times the WideResNet-34 layer shapes with each.
#   MFMA_ONLY            only the MFMAs (+ block prologue/epilogue)  -> structural ceiling
#   NO_STAGE             + LDS fragment reads + barrier
#   NO_BARRIER / full    + global loads, address math, LDS stores
# Results of round 1 are in DESIGN.md ("conv kernel: where the time goes").
set -e
cd "$(dirname "$0")/.."
SRC=$(python3 -c "from happypose_amd.build import SOURCES; print(' '.join(SOURCES))")
mkdir -p gpurun_out/abl
for v in "FULL:" "NO_BARRIER:-DHP_ABL_NO_BARRIER" "NO_STAGE:-DHP_ABL_NO_STAGE" \
         "MFMA_ONLY:-DHP_ABL_NO_STAGE -DHP_ABL_NO_BARRIER -DHP_ABL_NO_DSREAD"; do
  name=${v%%:*}; flags=${v#*:}
  (cd happypose_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags \
      -x hip $SRC -o ../../gpurun_out/abl/$name.so)
  echo "== $name"
  HAPPYPOSE_AMD_LIB=$PWD/gpurun_out/abl/$name.so python3 tools/conv_bench.py 2>&1 | grep -E "TFLOP"
done
