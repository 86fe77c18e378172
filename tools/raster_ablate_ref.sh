#!/bin/bash
# reference-state rasteriser, phases compiled out (variants built by tools/ab_local.sh raster.hip ...): run on the GPU box
mkdir -p gpurun_out/rz
for v in ${VARIANTS:-RFULL RNOCOVER RNOSHADE RNEITHER}; do
  HAPPYPOSE_AMD_LIB=happypose_amd/lib/abl/$v.so HP_STAGE_MSAA=1 HP_STAGE_ANISO=1 timeout 300 python3 tools/stage_workload.py > gpurun_out/rz/abl_$v.json 2>/dev/null
  echo $v $(python3 -c "import json;d=json.load(open('gpurun_out/rz/abl_$v.json'));print(round(d['C2']['raster']['us']), round(d['C3']['raster']['us']))")
done
