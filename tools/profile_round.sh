set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/bench_r01h.json 2> gpurun_out/bench_r01h.err
timeout 300 python bench.py --workload C3 --no-cpu-baseline > gpurun_out/bench_r01h_C3.json 2>/dev/null
timeout 300 python bench.py --workload C5 --no-cpu-baseline > gpurun_out/bench_r01h_C5.json 2>/dev/null
timeout 300 python bench.py --workload E2E --steps 5 --warmup 2 > gpurun_out/bench_r01h_E2E.json 2>/dev/null
timeout 300 python bench.py --workload E2E --steps 5 --warmup 2 --precision f16 > gpurun_out/bench_r01h_E2E_f16.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01h -o p --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-fp32 > gpurun_out/prof_r01h.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/trf_fetch -o p --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exact-fp32 > gpurun_out/trf_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/trf_write -o p --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exact-fp32 > gpurun_out/trf_write.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/trf_fetch gpurun_out/trf_write gpurun_out/r01h_conv_hbm_traffic.json 1800
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d gpurun_out/spmc1 -o p --output-format csv -- python3 tools/conv_bench.py > gpurun_out/spmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/spmc2 -o p --output-format csv -- python3 tools/conv_bench.py > gpurun_out/spmc2.log 2>&1
python3 tools/pmc_summary.py gpurun_out/spmc1/p_counter_collection.csv split > gpurun_out/r01h_split_pmc.txt
python3 tools/pmc_summary.py gpurun_out/spmc2/p_counter_collection.csv split >> gpurun_out/r01h_split_pmc.txt
rm -rf gpurun_out/trf_fetch/p_kernel_trace.csv gpurun_out/trf_write/p_kernel_trace.csv
ls gpurun_out/prof_r01h
