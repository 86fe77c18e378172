#!/bin/bash
# Round 6, GPU call 2: the full GPU suite with the measured C5 tolerances, the entry point's overhead probe, and a kernel TRACE of
# the C2 step (start / end of every kernel on both lanes' streams: where the step's time goes between the lanes).
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06b; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1
echo "tests rc=$?"; tail -8 $O/gpu_tests.log
timeout 600 python tools/probes/estimator_overhead.py > $O/estimator_overhead.txt 2> $O/estimator_overhead.err
head -c 1500 $O/estimator_overhead.txt; echo
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace -d $O/trace -o p --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-extra-workloads > $O/trace.log 2>&1
T=$(find $O/trace -name p_kernel_trace.csv | head -1)
python3 - "$T" $O/trace_c2.csv.gz <<'P'
import csv, gzip, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "kernel rows;", list(rows[0].keys()))
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id", "Stream_Id", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Thread_Id"]
keep = [k for k in keep if k in rows[0]]
with gzip.open(sys.argv[2], "wt") as f:
    w = csv.writer(f); w.writerow(keep)
    for r in rows:
        w.writerow([r[k][:60] if k == "Kernel_Name" else r[k] for k in keep])
P
find $O -name "p_kernel_trace.csv" -delete; find $O -name "*.db" -delete
du -sh $O
