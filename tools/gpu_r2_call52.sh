#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_fp8_gpu.py tests/test_heads_gpu.py -m gpu -x -q > $O/pytest_c52.log 2>&1; echo "pytest exit $?" >> $O/pytest_c52.log
grep -E "passed|failed|exit|Error|assert" $O/pytest_c52.log | tail -6
python tools/probes/gemm_intercept.py 2>&1 | grep -v amdgpu | tail -4
timeout 600 python tools/microbench.py --iters 5 > $O/microbench_c52.log 2>&1; grep -E "^gemm|^fp8" $O/microbench_c52.log
