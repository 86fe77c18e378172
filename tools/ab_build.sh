#!/bin/bash
# A/B builds of the library on one GPU box: VARIANTS="name:flags;name:flags", BENCH="command", FILTER="grep -E pattern"
set -e
cd "$(dirname "$0")/.."
SRC=$(python3 -c "from happypose_amd.build import SOURCES; print(' '.join(SOURCES))")
mkdir -p gpurun_out/abl
IFS=';' read -ra VS <<< "${VARIANTS:-FULL:}"
for v in "${VS[@]}"; do
  name=${v%%:*}; flags=${v#*:}
  (cd happypose_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags \
      -x hip $SRC -o ../../gpurun_out/abl/ab_$name.so 2>/dev/null)
done
for rep in 1 2; do
for v in "${VS[@]}"; do
  name=${v%%:*}
  echo "== $name (pass $rep)"
  HAPPYPOSE_AMD_LIB=$PWD/gpurun_out/abl/ab_$name.so ${BENCH:-python3 tools/conv_bench.py} 2>&1 | grep -E "${FILTER:-k3 s1}"
done
done
