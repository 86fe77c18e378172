#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06f; mkdir -p $O
export PYTHONUNBUFFERED=1
export HAPPYPOSE_AMD_LIB=$PWD/happypose_amd/lib_stamps/libhappypose_amd.so
for b in 64 128; do
  for v in 0 1; do
    HP_CONV_PP1=$v B=$b timeout 300 python tools/conv_bench.py > $O/stamps_b${b}_pp1_$v.txt 2>&1
    echo "== B=$b PP1=$v"; grep -E "x" $O/stamps_b${b}_pp1_$v.txt | grep -E "30x| 15x|  8x" 
  done
done
