#!/bin/bash
# Round-2 profile pass: (1) rocprofv3 --kernel-trace of the headline bench command, summarised per (kernel, grid, duration bucket);
# (2) HBM-side traffic of the dominant kernels, FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (MI355X_MICROARCH.md, HBM);
# (3) LDS / matrix-pipe counters of the GEMM and attention kernels in their own pass.  Counters never share a pass with a trace domain
# other than the kernel trace.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_r2
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_r2 -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_bench_r2.log 2>&1
for db in $(find $O/prof_r2 -name '*.db'); do python $R/tools/rocpd_summary.py $db --top 70 --split > $O/rocprof_kernel_stats_bench_r2.txt 2>&1; done
rm -rf $O/prof_r2
tail -1 $O/prof_bench_r2.log | cut -c1-300
head -40 $O/rocprof_kernel_stats_bench_r2.txt | cut -c1-150
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o g -- python $R/tools/microbench.py --iters 1 > $O/pmc_$c.log 2>&1
  for db in $(find $O/pmc_$c -name '*.db'); do python $R/tools/rocpd_summary.py $db --top 40 > $O/pmc_${c}_microbench_r2.txt 2>&1; done
  rm -rf $O/pmc_$c
  grep -E "attention_sp|attention_pp3|gemm_bf16|gemm_fp8|layernorm|qk_prep" $O/pmc_${c}_microbench_r2.txt | grep -E "FETCH|WRITE" | cut -c1-170
done
rm -rf $O/pmc_sq
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM -d $O/pmc_sq -o g -- python $R/tools/microbench.py --iters 1 > $O/pmc_sq.log 2>&1
for db in $(find $O/pmc_sq -name '*.db'); do python $R/tools/rocpd_summary.py $db --top 40 > $O/pmc_sq_microbench_r2.txt 2>&1; done
rm -rf $O/pmc_sq
grep -E "attention_sp|attention_pp3|gemm_bf16_pp2|gemm_bf16_w4b|gemm_fp8_pp" $O/pmc_sq_microbench_r2.txt | grep -E "SQ_" | cut -c1-170 | head -80
