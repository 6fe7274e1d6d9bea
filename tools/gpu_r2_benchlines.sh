#!/bin/bash
# Non-headline bench lines: fp8 linears (config 5's arithmetic) at 480p, Wan2.2 81f x 720p with the invariant cache off / on.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 900 python bench.py --steps 2 --warmup 1 --precision fp8 --no-cpu-baseline > $O/bench_fp8.log 2>&1; tail -1 $O/bench_fp8.log | cut -c1-400
timeout 900 python bench.py --model wan22 --height 720 --width 1280 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_wan22_720p.log 2>&1; tail -1 $O/bench_wan22_720p.log | cut -c1-400
timeout 900 python bench.py --model wan22 --height 720 --width 1280 --steps 1 --warmup 1 --no-cpu-baseline --cache-invariants > $O/bench_wan22_720p_cached.log 2>&1; tail -1 $O/bench_wan22_720p_cached.log | cut -c1-400
timeout 900 python bench.py --model wan22 --height 720 --width 1280 --steps 1 --warmup 1 --no-cpu-baseline --precision fp8 > $O/bench_wan22_720p_fp8.log 2>&1; tail -1 $O/bench_wan22_720p_fp8.log | cut -c1-400
timeout 900 python bench.py --steps 2 --warmup 1 --precision fp8 --fp8-attention --no-cpu-baseline > $O/bench_fp8_attn.log 2>&1; tail -1 $O/bench_fp8_attn.log | cut -c1-400
