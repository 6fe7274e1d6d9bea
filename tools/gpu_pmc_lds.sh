#!/bin/bash
# LDS and issue counters of the default attention kernels (counters in their own pass, kernel trace only).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf $O/pmc_lds
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -d $O/pmc_lds -o g -- python $R/tools/microbench.py --iters 1 --only attn --attn-variants 192,64 > $O/pmc_lds.log 2>&1
cd $R
for db in $(find $O/pmc_lds -name '*.db'); do python tools/rocpd_summary.py $db --top 20 > $O/pmc_lds_summary.txt 2>&1; done
grep -E "attention" $O/pmc_lds_summary.txt | head -80 | cut -c1-160
tail -3 $O/pmc_lds.log
rm -rf $O/pmc_lds
