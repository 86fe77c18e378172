#!/bin/bash
# Phase ablation of the rasteriser stage (reference state and single-sample) on the C2 / C3 inputs + per-kernel durations.
# Variants are built HERE by tools/ab_local.sh (happypose_amd/lib/abl/*.so), then run on the GPU box:
#   tools/ab_local.sh raster.hip "RFULL:-DHP_X" "RNOCOVER:-DHP_RABL_NO_COVER" "RNOSHADE:-DHP_RABL_NO_SHADE" \
#       "RNEITHER:-DHP_RABL_NO_COVER -DHP_RABL_NO_SHADE" "RNOSTORE:-DHP_RABL_NO_STORE"
mkdir -p gpurun_out/rz
for st in "1 1 ref" "0 0 single"; do
  set -- $st
  for v in ${VARIANTS:-RFULL RNOCOVER RNOSHADE RNEITHER RNOSTORE}; do
    HAPPYPOSE_AMD_LIB=happypose_amd/lib/abl/$v.so HP_STAGE_ONLY=raster HP_STAGE_MSAA=$1 HP_STAGE_ANISO=$2 timeout 300 python3 tools/stage_workload.py > gpurun_out/rz/abl_$3_$v.json 2>/dev/null
    echo $3 $v $(python3 -c "import json;d=json.load(open('gpurun_out/rz/abl_$3_$v.json'));print(round(d['C2']['raster']['us']), round(d['C3']['raster']['us']))")
  done
done
