#!/bin/bash
# Ablation of the rasteriser phases (run on the GPU box): coverage loop / resolve shading compiled out.
set -e
cd "$(dirname "$0")/.."
SRC=$(python3 -c "from happypose_amd.build import SOURCES; print(' '.join(SOURCES))")
mkdir -p gpurun_out/abl
for v in "FULL:" "NO_COVER:-DHP_RABL_NO_COVER" "NO_SHADE:-DHP_RABL_NO_SHADE" "NEITHER:-DHP_RABL_NO_COVER -DHP_RABL_NO_SHADE"; do
  name=${v%%:*}; flags=${v#*:}
  (cd happypose_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags \
      -x hip $SRC -o ../../gpurun_out/abl/r$name.so)
  echo "== $name"
  HAPPYPOSE_AMD_LIB=$PWD/gpurun_out/abl/r$name.so python3 tools/raster_crop_bench.py 2>&1 | grep -E "raster"
done
