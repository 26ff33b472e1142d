for t in 8 12 16 24; do echo "### READ_THREADS=$t"; VMAF_HIP_READ_THREADS=$t timeout -k 10 100 bash tools/trace_run.sh 2>&1 | grep -E "^==|all batches" ; done
nproc; python3 -c "import os; print(len(os.sched_getaffinity(0)))"
