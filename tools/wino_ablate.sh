#!/bin/bash
# Ablation of the Winograd conv loop (run on the GPU box); see tools/conv_ablate.sh.
set -e
cd "$(dirname "$0")/.."
SRC=$(python3 -c "from happypose_amd.build import SOURCES; print(' '.join(SOURCES))")
mkdir -p gpurun_out/abl
for v in ${VARIANTS:-"FULL:" "NO_GLOAD:-DHP_WABL_NO_GLOAD" "NO_GLOAD_RAW:-DHP_WABL_NO_GLOAD_RAW" "NO_GLOAD_U:-DHP_WABL_NO_GLOAD_U" "NO_LSTORE:-DHP_WABL_NO_LSTORE" "NO_STAGE:-DHP_WABL_NO_STAGE"}; do
  name=${v%%:*}; flags=${v#*:}
  (cd happypose_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags \
      -x hip $SRC -o ../../gpurun_out/abl/w$name.so)
  echo "== $name"
  HAPPYPOSE_AMD_LIB=$PWD/gpurun_out/abl/w$name.so python3 tools/conv_bench.py 2>&1 | grep -E "k3 s1 pre=0"
done
