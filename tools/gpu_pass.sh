#!/bin/bash
# ONE parameterised GPU pass (replaces the round-stamped gpu_r2_* / gpu_r3_* scripts): STAGES is a "|"-separated list of
#   tests[:<pytest args>]   the -m gpu suite (or a selection)            -> pytest_gpu_$TAG.log, parity_gpu.json
#   bench[:<bench args>]    bench.py                                     -> bench_$TAG.log
#   trace                   rocprofv3 --kernel-trace --stats of bench.py -> rocprof_kernel_stats_bench_$TAG.txt
#   pmc                     FETCH_SIZE / WRITE_SIZE of the self-attention launch alone (separate passes), SQ counters on the microbench
#   micro[:<args>]          tools/microbench.py                          -> microbench_$TAG.log
#   ab:<gemm_ab args>       tools/gemm_ab.py (interleaved A/B)           -> gemm_ab_$TAG.txt (appended)
#   run:<command>           anything else, logged to run_$TAG.log
# usage on the box:  STAGES="tests|bench|trace|ab:--kernels 4,blas --square" TAG=r4a bash tools/gpu_pass.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${TAG:-pass}; mkdir -p $O; cd $R
export TMPDIR=/tmp
IFS='|' read -ra LIST <<< "$STAGES"
for st in "${LIST[@]}"; do
  kind=${st%%:*}; arg=""; [ "$st" != "$kind" ] && arg=${st#*:}
  case $kind in
    tests)
      rm -f $O/parity_gpu.json
      timeout ${T_TESTS:-2400} python -m pytest ${arg:-tests} -m gpu -q -x > $O/pytest_gpu_$TAG.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu_$TAG.log
      grep -v "Warning\|warn\|amp\.\|super()" $O/pytest_gpu_$TAG.log | tail -${TAIL:-25} ;;
    bench)
      timeout ${T_BENCH:-1200} python bench.py ${arg:---steps 2 --warmup 1} > $O/bench_$TAG.log 2>&1; tail -1 $O/bench_$TAG.log | cut -c1-600 ;;
    trace)
      rm -rf $O/prof; ( cd /tmp; export FW_BENCH_DROPIN=0; timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline $arg > $O/prof_bench_$TAG.log 2>&1 )
      for db in $(find $O/prof -name '*.db'); do python tools/rocpd_summary.py $db --top 70 --split > $O/rocprof_kernel_stats_bench_$TAG.txt 2>&1; done
      rm -rf $O/prof; head -40 $O/rocprof_kernel_stats_bench_$TAG.txt | cut -c1-150 ;;
    pmc)
      for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf $O/pmc_$c; ( cd /tmp; timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o g -- python $R/tools/attn_self_only.py > $O/pmc_$c.log 2>&1 )
        for db in $(find $O/pmc_$c -name '*.db'); do python tools/rocpd_summary.py $db --top 20 > $O/pmc_${c}_attn_self_only_$TAG.txt 2>&1; done
        rm -rf $O/pmc_$c; grep -E "attention_sp|layernorm" $O/pmc_${c}_attn_self_only_$TAG.txt | cut -c1-170
      done
      rm -rf $O/pmc_sq; ( cd /tmp; timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM -d $O/pmc_sq -o g -- python $R/tools/microbench.py --iters 1 > $O/pmc_sq.log 2>&1 )
      for db in $(find $O/pmc_sq -name '*.db'); do python tools/rocpd_summary.py $db --top 40 > $O/pmc_sq_microbench_$TAG.txt 2>&1; done
      rm -rf $O/pmc_sq ;;
    micro)
      timeout 600 python tools/microbench.py ${arg:---iters 5} > $O/microbench_$TAG.log 2>&1; grep -E "gemm|attn|fp8 " $O/microbench_$TAG.log | head -60 ;;
    ab)
      echo "# tools/gemm_ab.py $arg" >> $O/gemm_ab_$TAG.txt
      timeout 600 python tools/gemm_ab.py $arg 2>&1 | tee -a $O/gemm_ab_$TAG.txt | tail -12 ;;
    run)
      timeout ${T_RUN:-900} bash -c "$arg" 2>&1 | tee -a $O/run_$TAG.log | tail -${TAIL:-30} ;;
  esac
done
