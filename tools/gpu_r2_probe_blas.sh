#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_blas
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_blas -o blas -- python $R/tools/probes/blas_kernel_name.py > $O/prof_blas.log 2>&1
cd $R
for db in $(find $O/prof_blas -name '*.db'); do python tools/rocpd_summary.py $db --top 12 > $O/prof_blas_summary.txt 2>&1; done
cat $O/prof_blas_summary.txt | cut -c1-400
find $O/prof_blas -name "*.csv" | head; for f in $(find $O/prof_blas -name "*kernel_stats*.csv"); do head -8 $f | cut -c1-500; done
rm -rf $O/prof_blas
