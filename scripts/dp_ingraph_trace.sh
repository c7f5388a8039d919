cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/dpi_trace; rm -rf gpurun_out/dpi_trace/*; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dpi_trace -o t -- python $GRAFT_REPO_ROOT/scripts/dp_ingraph_trace.py > $GRAFT_REPO_ROOT/gpurun_out/dpi_trace/run.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/dpi_trace/run.log
python scripts/dp_ingraph_trace.py --analyse $(find gpurun_out/dpi_trace -name "*kernel_trace.csv" | head -1) | tee gpurun_out/dpi_trace/overlap_summary.txt
find gpurun_out/dpi_trace -name "*kernel_trace.csv" -delete
