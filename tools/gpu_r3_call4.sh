#!/bin/bash
# round 3: full GPU parity suite at HEAD + compute-only emulation of both partitions (sequence shard / tensor parallel) on one GPU
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${TAG:-r3c}; mkdir -p $O; cd $R
rm -f $O/parity_gpu.json
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu_$TAG.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu_$TAG.log
grep -E "passed|failed|^FAILED|^ERROR|pytest exit|latents vs" $O/pytest_gpu_$TAG.log | tail -20
timeout 1500 python tools/shard_emulation.py --sp 1,2,4 --tp 2,4,8 --iters 2 2>&1 | grep -v amdgpu.ids > $O/shard_emulation_$TAG.txt; cat $O/shard_emulation_$TAG.txt
