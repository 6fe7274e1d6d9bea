#!/bin/bash
# tools/attn_counters.py under rocprofv3, one pass per counter group (counters only) -> gpurun_out/attn_counters_$TAG.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${TAG:-cnt}; export TMPDIR=/tmp; mkdir -p $O
GROUPS_=(
  "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES"
  "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_LDS"
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
)
dbs=()
for i in "${!GROUPS_[@]}"; do
  rm -rf $O/pmc_cnt_$i
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc ${GROUPS_[$i]} -d $O/pmc_cnt_$i -o g -- python $R/tools/attn_counters.py > $O/pmc_cnt_$i.log 2>&1 )
  for db in $(find $O/pmc_cnt_$i -name '*.db'); do dbs+=("$db"); done
done
python $R/tools/attn_counters.py --summarise "${dbs[@]}" > $O/attn_counters_$TAG.txt 2>&1
for i in "${!GROUPS_[@]}"; do rm -rf $O/pmc_cnt_$i; done
cat $O/attn_counters_$TAG.txt
