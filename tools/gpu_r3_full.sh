#!/bin/bash
# round 3 full pass: GPU parity suite, headline bench (+ HIP-graph A/B), kernel trace of the bench at HEAD, PMC traffic of the
# self-attention launch ALONE (separate FETCH / WRITE passes), SQ counters on the microbench.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${TAG:-r3}; mkdir -p $O; cd $R
rm -f $O/parity_gpu.json
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu_$TAG.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu_$TAG.log
grep -v "Warning\|warn\|amp\.\|super()" $O/pytest_gpu_$TAG.log | tail -25
timeout 1200 python bench.py --steps 2 --warmup 1 --hip-graph > $O/bench_$TAG.log 2>&1; tail -1 $O/bench_$TAG.log | cut -c1-400; tail -1 $O/bench_$TAG.log | grep -o '"hip_graph".*' | cut -c1-600
if [ -z "$SKIP_PROF" ]; then
export TMPDIR=/tmp; cd /tmp
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_bench_$TAG.log 2>&1
for db in $(find $O/prof -name '*.db'); do python $R/tools/rocpd_summary.py $db --top 70 --split > $O/rocprof_kernel_stats_bench_$TAG.txt 2>&1; done
rm -rf $O/prof
tail -1 $O/prof_bench_$TAG.log | cut -c1-200; head -45 $O/rocprof_kernel_stats_bench_$TAG.txt | cut -c1-150
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o g -- python $R/tools/attn_self_only.py > $O/pmc_$c.log 2>&1
  for db in $(find $O/pmc_$c -name '*.db'); do python $R/tools/rocpd_summary.py $db --top 20 > $O/pmc_${c}_attn_self_only_$TAG.txt 2>&1; done
  rm -rf $O/pmc_$c
  grep -E "attention_sp|layernorm" $O/pmc_${c}_attn_self_only_$TAG.txt | cut -c1-170
done
rm -rf $O/pmc_sq
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM -d $O/pmc_sq -o g -- python $R/tools/microbench.py --iters 1 > $O/pmc_sq.log 2>&1
for db in $(find $O/pmc_sq -name '*.db'); do python $R/tools/rocpd_summary.py $db --top 40 > $O/pmc_sq_microbench_$TAG.txt 2>&1; done
rm -rf $O/pmc_sq
grep -E "attention_sp|attention_pp3|gemm_bf16_pp2" $O/pmc_sq_microbench_$TAG.txt | grep -E "SQ_BUSY_CYCLES|SQ_VALU_MFMA_BUSY|BANK_CONFLICT" | cut -c1-200 | head -40
fi
cd $R
timeout 600 python tools/microbench.py --iters 5 > $O/microbench_$TAG.log 2>&1; grep -E "gemm|attn|fp8 " $O/microbench_$TAG.log | head -60
