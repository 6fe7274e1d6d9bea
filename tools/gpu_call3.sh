#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "gemm" > $O/pytest_c3.log 2>&1; echo "pytest exit $?" >> $O/pytest_c3.log
tail -15 $O/pytest_c3.log
timeout 600 python tools/microbench.py --iters 5 --only gemm --gemm-variants 2:0,2:1,2:4,1:0 > $O/mb_gemm3.log 2>&1; cat $O/mb_gemm3.log
