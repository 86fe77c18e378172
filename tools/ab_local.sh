#!/bin/bash
# A/B builds made HERE (hipcc cross-compiles): one source recompiled with extra flags and linked with the
# objects of the regular build.  The variants travel to the GPU box with the snapshot (happypose_amd/lib/abl/).
#   tools/ab_local.sh conv_split.hip "NOMASK:-DHP_SABL_NOMASK" "NOBAR:-DHP_SABL_NOBARRIER"
# then on the box:  HAPPYPOSE_AMD_LIB=happypose_amd/lib/abl/NOMASK.so python3 tools/conv_bench.py
set -e
cd "$(dirname "$0")/.."
src=$1; shift
mkdir -p happypose_amd/lib/abl
objs=$(python3 -c "from happypose_amd.build import SOURCES; print(' '.join('happypose_amd/build_obj/' + s.replace('.', '_') + '.o' for s in SOURCES if s != '$src'))")
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -fno-gpu-rdc $flags -x hip -c happypose_amd/csrc/$src -o /tmp/ab_$name.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o happypose_amd/lib/abl/$name.so $objs /tmp/ab_$name.o
  echo built $name
done
