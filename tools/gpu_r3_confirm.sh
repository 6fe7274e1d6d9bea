#!/bin/bash
# round 3: driver-style confirmation at HEAD -- smoke(), the full GPU parity suite, the default bench line
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${TAG:-r3d}; mkdir -p $O; cd $R
rm -f $O/parity_gpu.json
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$TAG.log 2>&1; echo "smoke exit $?" >> $O/smoke_$TAG.log; tail -3 $O/smoke_$TAG.log
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu_$TAG.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu_$TAG.log
grep -E "passed|failed|^FAILED|^ERROR|pytest exit" $O/pytest_gpu_$TAG.log | tail -20
timeout 900 python bench.py > $O/bench_$TAG.log 2>&1; echo "bench exit $?" >> $O/bench_$TAG.log
tail -n 2 $O/bench_$TAG.log | cut -c1-900
