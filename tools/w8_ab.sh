#!/bin/bash
# A/B builds of the Winograd kernels on one GPU box: VARIANTS="name:flags ..." (see tools/wino_ablate.sh)
set -e
cd "$(dirname "$0")/.."
SRC=$(python3 -c "from happypose_amd.build import SOURCES; print(' '.join(SOURCES))")
mkdir -p gpurun_out/abl
IFS=';' read -ra VS <<< "${VARIANTS:-FULL:}"
for v in "${VS[@]}"; do
  name=${v%%:*}; flags=${v#*:}
  (cd happypose_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags \
      -x hip $SRC -o ../../gpurun_out/abl/ab_$name.so)
done
for rep in 1 2; do
for v in "${VS[@]}"; do
  name=${v%%:*}
  echo "== $name (pass $rep)"
  HAPPYPOSE_AMD_LIB=$PWD/gpurun_out/abl/ab_$name.so python3 tools/conv_bench.py 2>&1 | grep -E "k3 s1" | awk '{printf "%s %s pre%s %s us | ", $1$2, $5, substr($7,5), $8} END {print ""}'
done
done
