#!/bin/bash
# round 3, call 1: the real reference on the GPU box + headline-size golden + default bench line (cpu_baseline kind=reference)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for m in einops tqdm safetensors transformers huggingface_hub numpy scipy PIL yaml; do python -c "import $m" 2>/dev/null && echo "have $m" || echo "MISSING $m"; done
rm -f $O/parity_gpu.json
timeout 1500 python -m pytest tests/test_reference_on_gpu.py "tests/test_joint_forward_gpu.py::test_full_size_forward_matches_reference_golden" -m gpu -q -s > $O/pytest_r3_call1.log 2>&1; echo "pytest exit $?" >> $O/pytest_r3_call1.log
grep -v Warning $O/pytest_r3_call1.log | tail -60
timeout 900 python bench.py --steps 2 --warmup 1 > $O/bench_r3_call1.log 2>&1; tail -1 $O/bench_r3_call1.log | cut -c1-1500
