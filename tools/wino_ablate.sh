#!/bin/bash
# Ablation of the Winograd conv loop (run on the GPU box); see tools/conv_ablate.sh.
set -e
cd "$(dirname "$0")/.."
SRC="api.cpp net.cpp raster.hip geometry.hip crop.hip conv.hip conv_patch.hip conv_wino.hip pool_head.hip"
mkdir -p gpurun_out/abl
for v in "FULL:" "NO_XFORM:-DHP_WABL_NO_XFORM" "NO_GLOBAL:-DHP_WABL_NO_GLOBAL" "NO_LDS_STORE:-DHP_WABL_NO_LDS_STORE" \
         "NO_BARRIER:-DHP_WABL_NO_BARRIER" "NO_READ_D:-DHP_WABL_NO_READ_D" "NO_DSREAD:-DHP_WABL_NO_DSREAD" \
         "NO_GLOBAL_STORE_BARRIER:-DHP_WABL_NO_GLOBAL -DHP_WABL_NO_LDS_STORE -DHP_WABL_NO_BARRIER" \
         "MFMA_XFORM:-DHP_WABL_NO_GLOBAL -DHP_WABL_NO_LDS_STORE -DHP_WABL_NO_BARRIER -DHP_WABL_NO_READ_D -DHP_WABL_NO_DSREAD" \
         "MFMA_ONLY:-DHP_WABL_NO_GLOBAL -DHP_WABL_NO_LDS_STORE -DHP_WABL_NO_BARRIER -DHP_WABL_NO_READ_D -DHP_WABL_NO_DSREAD -DHP_WABL_NO_XFORM"; do
  name=${v%%:*}; flags=${v#*:}
  (cd happypose_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags \
      -x hip $SRC -o ../../gpurun_out/abl/w$name.so)
  echo "== $name"
  HAPPYPOSE_AMD_LIB=$PWD/gpurun_out/abl/w$name.so python3 tools/conv_bench.py 2>&1 | grep -E "k3 s1 pre=0"
done
