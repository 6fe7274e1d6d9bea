#!/bin/bash
# Implicit-GEMM convolution: parity of the new op, the suites that now run on it, and the A/B against gather + GEMM at full size.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_heads_gpu.py tests/test_vae_gpu.py tests/test_pose_gpu.py -m gpu -x -q > $O/pytest_conv.log 2>&1; echo "pytest exit $?" >> $O/pytest_conv.log
grep -E "passed|failed|exit|Error" $O/pytest_conv.log | tail -5
timeout 600 python tools/heads_fullsize.py > $O/heads_implicit.log 2>&1; tail -5 $O/heads_implicit.log
timeout 600 python tools/heads_fullsize.py --gather > $O/heads_gather.log 2>&1; tail -5 $O/heads_gather.log
timeout 900 python tools/oneshot_fullsize.py > $O/oneshot_implicit.log 2>&1; tail -8 $O/oneshot_implicit.log
timeout 900 python tools/oneshot_fullsize.py --gather > $O/oneshot_gather.log 2>&1; tail -8 $O/oneshot_gather.log
