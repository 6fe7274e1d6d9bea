#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "gemm_ring" > $O/pytest_c4.log 2>&1; echo "pytest exit $?" >> $O/pytest_c4.log
tail -5 $O/pytest_c4.log
timeout 600 python tools/microbench.py --iters 5 --only gemm --gemm-variants 1:8,1:9,1:0 > $O/mb_gemm4.log 2>&1; cat $O/mb_gemm4.log
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d $O/pmc_tcc -o g -- python $R/tools/microbench.py --iters 1 --only gemm --gemm-variants 1:0 > $O/pmc_tcc.log 2>&1
cd $R
for db in $(find $O/pmc_tcc -name '*.db'); do python tools/rocpd_summary.py $db --top 12 --filter gemm > $O/pmc_tcc_summary.txt 2>&1; done
grep -E "ring_kernel" $O/pmc_tcc_summary.txt | head -60
tail -3 $O/pmc_tcc.log
find $O -name '*.db' -size +30M -delete
