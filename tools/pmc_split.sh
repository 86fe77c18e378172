# SQ counters of the split-fp16 conv kernel on the WideResNet layer shapes (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HP_CONV_NO_SPLITK=1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d gpurun_out/spmc1 -o p --output-format csv -- python3 tools/conv_bench.py > gpurun_out/spmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/spmc2 -o p --output-format csv -- python3 tools/conv_bench.py > gpurun_out/spmc2.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/sstat -o p --output-format csv -- python3 tools/conv_bench.py > gpurun_out/sstat.log 2>&1
python3 tools/pmc_summary.py gpurun_out/spmc1/p_counter_collection.csv split > gpurun_out/split_pmc.txt
python3 tools/pmc_summary.py gpurun_out/spmc2/p_counter_collection.csv split >> gpurun_out/split_pmc.txt
grep split gpurun_out/sstat/p_kernel_stats.csv | cut -c1-300 >> gpurun_out/split_pmc.txt
rm -f gpurun_out/spmc1/*trace* gpurun_out/spmc2/*trace* gpurun_out/sstat/p_kernel_trace.csv
cat gpurun_out/split_pmc.txt
