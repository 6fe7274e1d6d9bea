#!/bin/bash
# GPU call: parity tests, whole-step bench, rocprofv3 kernel trace of the same bench command, per-kernel microbench.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --steps 2 --warmup 1 > $O/bench.log 2>&1; tail -1 $O/bench.log
timeout 600 python tools/microbench.py --iters 5 > $O/microbench.log 2>&1; tail -30 $O/microbench.log
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1
cd $R
ls -la $O/prof $O/prof/* | head -30
for db in $(find $O/prof -name '*.db'); do python tools/rocpd_summary.py $db --top 60 > $O/prof_summary.txt 2>&1; done
find $O/prof -name '*stats*' | head
cat $O/prof_summary.txt | head -70
# keep the merged output small: drop the raw db if it is big
find $O/prof -size +40M -delete
