#!/bin/bash
# Effective shader clock per kernel under load: GRBM_GUI_ACTIVE (cycles the GPU was busy, per dispatch) / the dispatch's duration, for the
# two ping-pong GEMMs, the vendor BLAS (yardstick) and the attention kernels, on random and on zero operands -- one rocprofv3 --pmc
# pass (counters only).  -> gpurun_out/clock_probe_$TAG.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${TAG:-clk}; export TMPDIR=/tmp
for fill in random zeros; do
  rm -rf $O/pmc_clk
  ( cd /tmp; timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc_clk -o g -- \
      python $R/tools/clock_probe.py --$fill > $O/pmc_clk_$fill.log 2>&1 )
  for db in $(find $O/pmc_clk -name '*.db'); do python $R/tools/clock_probe.py --summarise $db --fill $fill >> $O/clock_probe_$TAG.txt 2>&1; done
  rm -rf $O/pmc_clk
done
cat $O/clock_probe_$TAG.txt
