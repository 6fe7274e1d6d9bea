#!/bin/bash
# Full GPU pass: parity suite (writes gpurun_out/parity_gpu.json), microbench, headline bench, optional extra bench lines.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${TAG:-x}
mkdir -p $O
cd $R
rm -f $O/parity_gpu.json
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_$TAG.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu_$TAG.log
tail -4 $O/pytest_gpu_$TAG.log
timeout 600 python tools/microbench.py --iters 5 > $O/microbench_$TAG.log 2>&1; grep -E "gemm|attn|fp8 " $O/microbench_$TAG.log
timeout 900 python bench.py --steps 2 --warmup 1 ${BENCH_EXTRA:-} > $O/bench_$TAG.log 2>&1; tail -1 $O/bench_$TAG.log | cut -c1-700
