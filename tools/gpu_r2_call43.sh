#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_joint_forward_gpu.py -m gpu -x -q -k "sharded_engine or rccl" > $O/pytest_shard.log 2>&1; echo "pytest exit $?" >> $O/pytest_shard.log
grep -E "passed|failed|exit|Error|assert" $O/pytest_shard.log | tail -8
timeout 900 python bench.py --model wan22 --frames 121 --height 720 --width 1280 --steps 1 --warmup 1 --no-cpu-baseline --precision fp8 > $O/bench_cfg5_fp8.log 2>&1; tail -1 $O/bench_cfg5_fp8.log | cut -c1-500
timeout 600 python tools/microbench.py --iters 5 > $O/microbench_c43.log 2>&1; grep -E "^gemm" $O/microbench_c43.log
