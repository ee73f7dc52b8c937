#!/bin/bash
# tools/profile_bev_pmc.sh [tag]: the SQ / LDS / clock / HBM counter passes of tools/profile_r05_bound.sh for the BEV 3x3 kernels alone
# (tools/bev_micro.py under rocprofv3 --pmc, one counter group per pass) -> gpurun_out/<tag>/r05_bev_mfma.md, r05_bev_pmc.json
R=$GRAFT_REPO_ROOT; TAG=${1:-bev_pmc}; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES"
B="SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"
D="GRBM_GUI_ACTIVE"
python3 $R/tools/bev_micro.py 20 > $OUT/bev_micro.txt 2>&1
i=0
for G in "$A" "$B" "$C" "$D" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/p_b$i
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/p_b$i -o b -- python3 $R/tools/bev_micro.py 10 > $OUT/bev_pass$i.log 2>&1
  cp $(find /tmp/p_b$i -name "*counter_collection.csv" | head -1) $OUT/bev_counters_$i.csv 2>/dev/null
  cp $(find /tmp/p_b$i -name "*kernel_trace.csv" | head -1) $OUT/bev_trace_$i.csv 2>/dev/null
done
python3 $R/tools/summarize_r05_bound.py $OUT > $OUT/summary.log 2>&1
find $OUT -name "*.csv" -size +8M -delete
ls $OUT; tail -3 $OUT/summary.log; cat $OUT/bev_micro.txt | grep us
