#!/bin/bash
# Whole-step bench + full GPU parity suite + rocprofv3 kernel trace of the bench command.  TAG names the outputs.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${TAG:-x}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_$TAG.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu_$TAG.log
tail -3 $O/pytest_gpu_$TAG.log
timeout 600 python bench.py --steps 2 --warmup 1 > $O/bench_$TAG.log 2>&1; tail -1 $O/bench_$TAG.log
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_bench_$TAG.log 2>&1
cd $R
for db in $(find $O/prof_$TAG -name '*.db'); do python tools/rocpd_summary.py $db --top 60 > $O/prof_summary_$TAG.txt 2>&1; done
head -30 $O/prof_summary_$TAG.txt | cut -c1-150
rm -rf $O/prof_$TAG
