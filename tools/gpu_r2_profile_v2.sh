#!/bin/bash
# End-of-round kernel trace of the headline bench command (rocprofv3 --kernel-trace --stats), summarised per (kernel, grid, bucket).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_r2
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_r2 -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_bench_r2b.log 2>&1
for db in $(find $O/prof_r2 -name '*.db'); do python $R/tools/rocpd_summary.py $db --top 70 --split > $O/rocprof_kernel_stats_bench_r2b.txt 2>&1; done
rm -rf $O/prof_r2
tail -1 $O/prof_bench_r2b.log | cut -c1-300
head -60 $O/rocprof_kernel_stats_bench_r2b.txt | cut -c1-150
