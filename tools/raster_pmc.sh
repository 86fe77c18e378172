#!/bin/bash
# SQ counters of the rasteriser stage (reference state, C2 inputs) for the full build and the ablation variants.
mkdir -p gpurun_out/rpmc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-RFULL RNOSHADE}; do
  HAPPYPOSE_AMD_LIB=happypose_amd/lib/abl/$v.so HP_STAGE_ONLY=raster HP_STAGE_WORKLOADS=C2 HP_STAGE_MSAA=1 HP_STAGE_ANISO=1 timeout 300 rocprofv3 --kernel-trace \
    --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE \
    -d gpurun_out/rpmc/$v -o p --output-format csv -- python3 tools/stage_workload.py > gpurun_out/rpmc/$v.log 2>&1
  HAPPYPOSE_AMD_LIB=happypose_amd/lib/abl/$v.so HP_STAGE_ONLY=raster HP_STAGE_WORKLOADS=C2 HP_STAGE_MSAA=1 HP_STAGE_ANISO=1 timeout 300 rocprofv3 --kernel-trace \
    --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS \
    -d gpurun_out/rpmc/${v}_b -o p --output-format csv -- python3 tools/stage_workload.py >> gpurun_out/rpmc/$v.log 2>&1
  echo == $v
  python3 tools/pmc_kernels.py $(find gpurun_out/rpmc/$v -name "*counter_collection.csv" | head -1) | grep -E "kernel|raster"
  python3 - <<PY
import csv, collections, glob
for d in ("gpurun_out/rpmc/$v", "gpurun_out/rpmc/${v}_b"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "raster" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        print(k, {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
done
