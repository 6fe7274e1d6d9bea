#!/bin/bash
# End-of-round extras: merged-CFG step against the two-call step on one box; kernel trace of the geometry heads with implicit-GEMM convolutions.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_two_calls_final.log 2>&1; tail -1 $O/bench_two_calls_final.log | cut -c1-260
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --merge-cfg > $O/bench_merge_cfg_final.log 2>&1; tail -1 $O/bench_merge_cfg_final.log | cut -c1-260
cd /tmp
rm -rf $O/prof_heads
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_heads -o heads -- python $R/tools/heads_fullsize.py --reps 2 > $O/prof_heads.log 2>&1
for db in $(find $O/prof_heads -name '*.db'); do python $R/tools/rocpd_summary.py $db --top 30 > $O/rocprof_kernel_stats_heads_r2.txt 2>&1; done
rm -rf $O/prof_heads
head -24 $O/rocprof_kernel_stats_heads_r2.txt | cut -c1-150
