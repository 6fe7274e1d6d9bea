#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
for v in ${VARS:-32 96}; do
  FW_GEMM_KERNEL=4 FW_GEMM_VAR=$v timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "gemm" > $O/pytest_gemm_k4v$v.log 2>&1; echo "k4 v$v pytest exit $?"; tail -1 $O/pytest_gemm_k4v$v.log
done
timeout 600 python tools/microbench.py --iters 8 --only gemmonly --gemm-variants ${GV:-4:0,4:32,4:96} > $O/microbench_c6.log 2>&1; grep -E "==|dit|ffn|vggt|bicross" $O/microbench_c6.log
