#!/bin/bash
# Phase ablation of the persistent stem kernels: builds of one source file with phases compiled out (VARIANTS = ';'-separated
# macro sets, FULL = none), linked against the other objects of the last build.
#   KERNEL=s7 (default): conv_stem7x7s2_pool_f16_pp, macros HP_S7_ABL_{NOMFMA,NOEPI,NOLOAD,NOPRIO}, bench C5
#   KERNEL=s5:           conv_stem5x5s2_pool_split_pp, macros HP_S5_ABL_{NOMFMA,NOEPI,NOLOAD}, bench C2
# bash tools/stem_ablate.sh   here, then on the GPU box:   bash tools/stem_ablate.sh run
cd "$(dirname "$0")/.."
K=${KERNEL:-s7}
if [ $K = s7 ]; then SRC=conv_stem7; PAT=stem7; WL=C5; P=HP_S7_ABL; else SRC=conv_stem_split; PAT=stem5x5; WL=C2; P=HP_S5_ABL; fi
if [ "$1" != "run" ]; then
  mkdir -p happypose_amd/lib/abl; rm -f happypose_amd/lib/abl/${K}_*.so
  IFS=';' read -ra VS <<< "${VARIANTS:-FULL;${P}_NOMFMA;${P}_NOEPI;${P}_NOLOAD;${P}_NOEPI -D${P}_NOLOAD;${P}_NOMFMA -D${P}_NOEPI;${P}_NOMFMA -D${P}_NOEPI -D${P}_NOLOAD}"
  for v in "${VS[@]}"; do
    name=$(echo $v | tr -d ' ' | sed "s/-D/_/g")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -D$v -x hip -c happypose_amd/csrc/$SRC.hip -o /tmp/${K}_$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o happypose_amd/lib/abl/${K}_$name.so $(ls happypose_amd/build_obj/*.o | grep -v ${SRC}_hip) /tmp/${K}_$name.o
  done
  ls happypose_amd/lib/abl; exit 0
fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S="--workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-extra-workloads --lanes 1"
for so in happypose_amd/lib/abl/${K}_*.so; do
  export HAPPYPOSE_AMD_LIB=$PWD/$so
  rocprofv3 --kernel-trace --stats -d gpurun_out/ks_abl -o p --output-format csv -- python3 bench.py $S > gpurun_out/ks_abl.log 2>&1
  echo "$so $(grep -E "$PAT" $(find gpurun_out/ks_abl -name p_kernel_stats.csv) | cut -d, -f2-4)"; rm -rf gpurun_out/ks_abl
done
