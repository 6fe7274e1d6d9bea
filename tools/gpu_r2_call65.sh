#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_joint_forward_gpu.py -m gpu -x -q -k "fp8" > $O/pytest_c65.log 2>&1; echo "pytest exit $?" >> $O/pytest_c65.log
grep -E "passed|failed|exit|Error|assert" $O/pytest_c65.log | tail -8
timeout 900 python bench.py --steps 2 --warmup 1 --precision fp8 --fp8-attention --no-cpu-baseline > $O/bench_fp8_attn.log 2>&1; tail -1 $O/bench_fp8_attn.log | cut -c1-600
