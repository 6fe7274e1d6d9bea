#!/bin/bash
# Round-2 GPU call 4: second-generation bf16 ping-pong kernel (FW_GEMM_KERNEL=4; var bit 0 = LDS-DMA issued from the LOAD phases).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for v in 0 1; do
  FW_GEMM_KERNEL=4 FW_GEMM_VAR=$v timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "gemm" > $O/pytest_gemm_k4v$v.log 2>&1; echo "k4 v$v pytest exit $?"; tail -2 $O/pytest_gemm_k4v$v.log
done
timeout 600 python tools/microbench.py --iters 8 --only gemmonly --gemm-variants 3:1,4:0,4:1 > $O/microbench_c4.log 2>&1; grep -E "==|dit|ffn|vggt|bicross" $O/microbench_c4.log
