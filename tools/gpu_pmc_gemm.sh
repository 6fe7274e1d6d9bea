#!/bin/bash
# LDS / issue counters of the default GEMM kernel (counters in their own pass, kernel trace only).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf $O/pmc_gemm
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM -d $O/pmc_gemm -o g -- python $R/tools/microbench.py --iters 1 --only gemm > $O/pmc_gemm.log 2>&1
cd $R
for db in $(find $O/pmc_gemm -name '*.db'); do python tools/rocpd_summary.py $db --top 20 > $O/pmc_gemm_summary.txt 2>&1; done
grep -E "gemm_bf16_pp" $O/pmc_gemm_summary.txt | head -60 | cut -c1-160
tail -3 $O/pmc_gemm.log
rm -rf $O/pmc_gemm
