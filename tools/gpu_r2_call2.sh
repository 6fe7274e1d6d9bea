#!/bin/bash
# Round-2 GPU call 2: fp8 path (op tests, engine-vs-fp8-oracle, microbench, bench line).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_joint_forward_gpu.py -m gpu -x -q -k "fp8" -s > $O/pytest_fp8_c2.log 2>&1; echo "pytest exit $?" >> $O/pytest_fp8_c2.log
grep -E "passed|failed|Error|error|rel-L2|assert" $O/pytest_fp8_c2.log | tail -15
timeout 600 python tools/microbench.py --iters 5 --only fp8 > $O/microbench_fp8_c2.log 2>&1; tail -8 $O/microbench_fp8_c2.log
timeout 900 python bench.py --steps 2 --warmup 1 --precision fp8 --no-cpu-baseline > $O/bench_fp8_c2.log 2>&1; tail -1 $O/bench_fp8_c2.log
