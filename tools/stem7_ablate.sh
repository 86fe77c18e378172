#!/bin/bash
# Phase ablation of conv_stem7x7s2_pool_f16_pp (C5, one lane): builds of conv_stem7.hip with a phase compiled out (VARIANTS =
# ';'-separated macro sets, FULL = none), linked against the other objects of the last build:  bash tools/stem7_ablate.sh
# here, then on the GPU box:  bash tools/stem7_ablate.sh run
cd "$(dirname "$0")/.."
if [ "$1" != "run" ]; then
  mkdir -p happypose_amd/lib/abl; rm -f happypose_amd/lib/abl/s7_*.so
  IFS=';' read -ra VS <<< "${VARIANTS:-FULL;HP_S7_ABL_NOMFMA;HP_S7_ABL_NOEPI;HP_S7_ABL_NOLOAD;HP_S7_ABL_NOPRIO;HP_S7_ABL_NOEPI -DHP_S7_ABL_NOLOAD;HP_S7_ABL_NOMFMA -DHP_S7_ABL_NOEPI;HP_S7_ABL_NOMFMA -DHP_S7_ABL_NOEPI -DHP_S7_ABL_NOLOAD}"
  for v in "${VS[@]}"; do
    name=$(echo $v | tr -d ' ' | sed "s/-D/_/g")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -D$v -x hip -c happypose_amd/csrc/conv_stem7.hip -o /tmp/s7_$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o happypose_amd/lib/abl/s7_$name.so $(ls happypose_amd/build_obj/*.o | grep -v conv_stem7) /tmp/s7_$name.o
  done
  ls happypose_amd/lib/abl; exit 0
fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S="--workload C5 --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-extra-workloads --lanes 1"
for so in happypose_amd/lib/abl/s7_*.so; do
  export HAPPYPOSE_AMD_LIB=$PWD/$so
  rocprofv3 --kernel-trace --stats -d gpurun_out/ks_s7 -o p --output-format csv -- python3 bench.py $S > gpurun_out/ks_s7.log 2>&1
  echo "$so $(grep -E "stem7" $(find gpurun_out/ks_s7 -name p_kernel_stats.csv) | cut -d, -f2-4)"; rm -rf gpurun_out/ks_s7
done
