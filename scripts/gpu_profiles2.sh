#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
NAME=sac_depth bash scripts/profile_round.sh > /dev/null 2>&1
NAME=sac_depth_b128 BENCH_ARGS="--global-batch 128" PMC=0 bash scripts/profile_round.sh > /dev/null 2>&1
NAME=sac_rgbd BENCH_ARGS="--workload sac_rgbd" bash scripts/profile_round.sh > /dev/null 2>&1
NAME=bdq_per BENCH_ARGS="--workload bdq_per" PMC=0 bash scripts/profile_round.sh > /dev/null 2>&1
NAME=ae_train BENCH_ARGS="--workload ae_train" PMC=0 bash scripts/profile_round.sh > /dev/null 2>&1
mkdir -p gpurun_out/dp_trace; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/dp_trace -o t -- python $R/scripts/dp_overlap_trace.py > $R/gpurun_out/dp_trace/run.log 2>&1
cd $R
python scripts/dp_overlap_trace.py --analyse $(find gpurun_out/dp_trace -name "*kernel_trace.csv" | head -1) > gpurun_out/dp_trace/overlap_summary.txt 2>&1
find gpurun_out/dp_trace -name "*kernel_trace.csv" -size +4M -delete
for n in sac_depth sac_depth_b128 sac_rgbd bdq_per ae_train; do echo "=== $n"; cat gpurun_out/prof_$n/kernel_summary.txt | cut -c1-160; done
cat gpurun_out/dp_trace/overlap_summary.txt; tail -3 gpurun_out/dp_trace/run.log
