#!/bin/bash
# rocprofv3 kernel trace of the pose encoder and the VAE decoder at full size (tools/oneshot_fullsize.py).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_oneshot
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_oneshot -o oneshot -- python $R/tools/oneshot_fullsize.py > $O/prof_oneshot.log 2>&1
cd $R
for db in $(find $O/prof_oneshot -name '*.db'); do python tools/rocpd_summary.py $db --top 30 > $O/prof_summary_oneshot.txt 2>&1; done
grep -E "rep [01]" $O/prof_oneshot.log
head -26 $O/prof_summary_oneshot.txt | cut -c1-150
rm -rf $O/prof_oneshot
