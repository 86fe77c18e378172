cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in RFULL SNOREC SNOBIN SNEITHER; do
  HAPPYPOSE_AMD_LIB=happypose_amd/lib/abl/$v.so HP_STAGE_MSAA=1 HP_STAGE_ANISO=1 HP_STAGE_ONLY=raster HP_STAGE_WORKLOADS=C2 rocprofv3 --kernel-trace -d gpurun_out/prof_sabl_$v -o p -- python3 tools/stage_workload.py > /dev/null 2>&1
  echo == $v; python3 tools/rocprof_kernels.py gpurun_out/prof_sabl_$v 20 | grep -E "raster_setup|raster_xform|raster_kernel"
done
