#!/bin/bash
# Round-2 GPU call 1: scaled-MFMA layout probe, the full GPU parity suite (writes gpurun_out/parity_gpu.json), per-kernel
# microbench with the vendor-BLAS yardstick, the headline bench line and the Wan2.2 81f x 720p bench line.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
nproc; free -g | head -2
./tools/probes/mfma_scale_probe > $O/mfma_scale_probe.txt 2>&1; cat $O/mfma_scale_probe.txt
rm -f $O/parity_gpu.json
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_c1.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu_c1.log
tail -5 $O/pytest_gpu_c1.log
timeout 600 python tools/microbench.py --iters 5 --blas-ceiling > $O/microbench_c1.log 2>&1; tail -45 $O/microbench_c1.log
timeout 900 python bench.py --steps 2 --warmup 1 > $O/bench_c1.log 2>&1; tail -1 $O/bench_c1.log
timeout 900 python bench.py --model wan22 --height 720 --width 1280 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_wan22_720p_c1.log 2>&1; tail -1 $O/bench_wan22_720p_c1.log
