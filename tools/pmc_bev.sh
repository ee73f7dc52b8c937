#!/bin/bash
# MFMA utilisation of the dense BEV head (north_star: "MFMA utilisation on the BEV head"):
# kernel trace + SQ counters of tools/bev_only.py, summarised by tools/summarize_bev_pmc.py
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/bev; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $OUT/pmc -- python3 $R/tools/bev_only.py > $OUT/pmc.log 2>&1
tail -1 $OUT/pmc.log
