#!/bin/bash
# Matrix-pipe / LDS counters of the GEMM and attention kernels at the end of the round (own pass: --pmc with the kernel trace only).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf $O/pmc_sq
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM -d $O/pmc_sq -o g -- python $R/tools/microbench.py --iters 1 > $O/pmc_sq.log 2>&1
for db in $(find $O/pmc_sq -name '*.db'); do python $R/tools/rocpd_summary.py $db --top 40 > $O/pmc_sq_microbench_r2_v2.txt 2>&1; done
rm -rf $O/pmc_sq
grep -E "attention_sp|attention_pp3|gemm_bf16_pp2|gemm_fp8_pp" $O/pmc_sq_microbench_r2_v2.txt | grep -E "SQ_BUSY_CYCLES|SQ_VALU_MFMA_BUSY|BANK_CONFLICT" | cut -c1-200 | head -60
