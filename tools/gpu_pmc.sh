#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf $O/pmc_clk
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -d $O/pmc_clk -o g -- python $R/tools/microbench.py --iters 2 --only attn --attn-variants 18,0 > $O/pmc_clk.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -d $O/pmc_clk2 -o g -- python $R/tools/microbench.py --iters 2 --only gemm --gemm-variants 3:1 > $O/pmc_clk2.log 2>&1
cd $R
for d in pmc_clk pmc_clk2; do for db in $(find $O/$d -name '*.db'); do python tools/rocpd_summary.py $db --top 14 > $O/${d}_summary.txt 2>&1; done; done
grep -E "attention|gemm_bf16" $O/pmc_clk_summary.txt $O/pmc_clk2_summary.txt | head -150
rm -rf $O/pmc_clk $O/pmc_clk2
