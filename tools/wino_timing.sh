#!/bin/bash
# Build the library with per-block time stamps in the Winograd kernel and print them (GPU box).
set -e
cd "$(dirname "$0")/.."
SRC=$(python3 -c "from happypose_amd.build import SOURCES; print(' '.join(SOURCES))")
mkdir -p gpurun_out/abl
(cd happypose_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DHP_WABL_TIMING ${EXTRA_FLAGS} \
    -x hip $SRC -o ../../gpurun_out/abl/wTIMING.so)
HAPPYPOSE_AMD_LIB=$PWD/gpurun_out/abl/wTIMING.so python3 tools/wino_timing.py
