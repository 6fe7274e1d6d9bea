#!/bin/bash
# round 3: the multi-rank bench path end to end on ONE GPU (ranks share device 0, gloo instead of RCCL): launcher, topology, both
# partitions, CFG groups.  Not a measurement -- a first-run check of the code the driver's 8-GPU run would execute.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export FW_BENCH_DEVICE=0 FW_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for mode in sp tp; do
  for n in 2 4; do
    FW_PARALLEL=$mode timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
      bench.py --gpus $n --steps 1 --warmup 0 --no-cpu-baseline > $O/dryrun_${mode}_$n.log 2>&1
    echo "$mode n=$n exit $?"; tail -n 1 $O/dryrun_${mode}_$n.log | cut -c1-700
  done
done
