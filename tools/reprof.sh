set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01h -o p --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-fp32 > gpurun_out/prof_r01h.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/trf_fetch -o p --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exact-fp32 > gpurun_out/trf_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/trf_write -o p --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exact-fp32 > gpurun_out/trf_write.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/trf_fetch gpurun_out/trf_write gpurun_out/r01h_conv_hbm_traffic.json 1800
rm -rf gpurun_out/trf_fetch/p_kernel_trace.csv gpurun_out/trf_write/p_kernel_trace.csv gpurun_out/prof_r01h/p_kernel_trace.csv
