#!/bin/bash
# counters of the extractor kernel under a given GLX_POINTNET_FORM: bash tools/experiments/pointnet_pmc.sh <form>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GLX_POINTNET_FORM=$1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pn_pmc_$1_$i -- python3 $R/tools/sampler_time.py 5 > /dev/null 2>&1
  f=$(find $R/gpurun_out/pn_pmc_$1_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_table.py $f k_pointnet_feat
done
