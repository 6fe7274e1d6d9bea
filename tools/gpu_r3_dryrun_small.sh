#!/bin/bash
# round 3: multi-PROCESS first-run check of both partitions at a small workload (ranks share GPU 0, gloo): the order of the asynchronous
# collectives across separate processes (deadlock freedom), the launcher, the topology -- in seconds instead of the minutes per step
# that gloo's host staging costs at the BASELINE size (tools/gpu_r3_dryrun_ranks.sh).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export FW_BENCH_DEVICE=0 FW_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for mode in sp tp; do
  for n in 4 8; do
    FW_PARALLEL=$mode timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) \
      bench.py --gpus $n --steps 2 --warmup 1 --no-cpu-baseline --layers 18 --frames 17 --height 128 --width 128 > $O/dryrun_small_${mode}_$n.log 2>&1
    echo "$mode n=$n exit $?"; tail -n 1 $O/dryrun_small_${mode}_$n.log | cut -c1-420
  done
done
