cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_data_parallel.py -q -x 2>&1 | tail -2
mkdir -p gpurun_out/dp_trace; rm -f gpurun_out/dp_trace/*; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dp_trace -o t -- python $GRAFT_REPO_ROOT/scripts/dp_overlap_trace.py > $GRAFT_REPO_ROOT/gpurun_out/dp_trace/run.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/dp_overlap_trace.py --analyse $(find gpurun_out/dp_trace -name "*kernel_trace.csv" | head -1) | tee gpurun_out/dp_trace/overlap_summary.txt
