#!/bin/bash
# rocprofv3 kernel trace of the full-size geometry heads (tools/heads_fullsize.py): per-kernel time of the A20 tail.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_heads
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_heads -o heads -- python $R/tools/heads_fullsize.py --reps 2 > $O/prof_heads.log 2>&1
cd $R
for db in $(find $O/prof_heads -name '*.db'); do python tools/rocpd_summary.py $db --top 40 > $O/prof_summary_heads.txt 2>&1; done
head -40 $O/prof_summary_heads.txt | cut -c1-160
tail -5 $O/prof_heads.log
rm -rf $O/prof_heads
