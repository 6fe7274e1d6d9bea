#!/bin/bash
# GPU call 2: new glue kernels + ring GEMM: parity, A/B microbench, PMC counters of the GEMM kernels.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "gemm or layernorm or qk_prep" > $O/pytest_c2.log 2>&1; echo "pytest exit $?" >> $O/pytest_c2.log
tail -5 $O/pytest_c2.log
timeout 600 python tools/microbench.py --iters 5 --only gemm --gemm-variants 1:0,1:1,1:2,1:3,1:4,0:0 > $O/mb_gemm.log 2>&1; cat $O/mb_gemm.log
timeout 300 python tools/microbench.py --iters 5 --only glue > $O/mb_glue.log 2>&1; cat $O/mb_glue.log
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/pmc_gemm -o g -- python $R/tools/microbench.py --iters 1 --only gemm --gemm-variants 1:0,1:4,0:0 > $O/pmc_gemm.log 2>&1
cd $R
for db in $(find $O/pmc_gemm -name '*.db'); do python tools/rocpd_summary.py $db --top 12 --filter gemm > $O/pmc_gemm_summary.txt 2>&1; done
cat $O/pmc_gemm_summary.txt | grep -v "^at::" | head -80
tail -5 $O/pmc_gemm.log
find $O -name '*.db' -size +30M -delete
