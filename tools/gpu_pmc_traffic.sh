#!/bin/bash
# HBM traffic of the dominant kernels (MI355X_MICROARCH.md section HBM): FETCH_SIZE and WRITE_SIZE in SEPARATE passes.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o g -- python $R/tools/microbench.py --iters 1 > $O/pmc_$c.log 2>&1
  for db in $(find $O/pmc_$c -name '*.db'); do python $R/tools/rocpd_summary.py $db --top 30 > $O/pmc_${c}_summary.txt 2>&1; done
  rm -rf $O/pmc_$c
done
grep -E "attention_sp|attention_pp3|gemm_bf16_pp|layernorm|qk_prep" $O/pmc_FETCH_SIZE_summary.txt | grep -E "FETCH|WRITE" | head -40
grep -E "attention_sp|attention_pp3|gemm_bf16_pp|layernorm|qk_prep" $O/pmc_WRITE_SIZE_summary.txt | grep -E "FETCH|WRITE" | head -40
