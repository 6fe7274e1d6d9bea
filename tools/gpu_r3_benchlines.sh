#!/bin/bash
# round 3: the non-headline bench lines with the final code (never `value` of the headline): merged CFG, fp8 linears, fp8 linears + fp8
# attention (parity unpinned), Wan2.2 81f x 720p (BASELINE configs[3] on one GPU), Wan2.2 121f x 720p fp8 (configs[4]'s workload)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { tag=$1; shift; timeout 1200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/bench_r3_$tag.log 2>&1; tail -1 $O/bench_r3_$tag.log | cut -c1-330; }
run merge_cfg --merge-cfg
run fp8 --precision fp8
run fp8_attn --precision fp8 --fp8-attention
run wan22_720p --model wan22 --height 720 --width 1280
run wan22_720p_fp8 --model wan22 --height 720 --width 1280 --precision fp8
timeout 1500 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --model wan22 --frames 121 --height 720 --width 1280 --precision fp8 > $O/bench_r3_cfg5_121f_720p_fp8.log 2>&1; tail -1 $O/bench_r3_cfg5_121f_720p_fp8.log | cut -c1-330
