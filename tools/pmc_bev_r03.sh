#!/bin/bash
# MFMA utilisation of the dense BEV head on the path the training step runs: SQ counters of the step itself
# (bench.py --mode static: the same launches, eager, so that counters attribute per dispatch), summarised per kernel
# over the last steps by tools/summarize_bev_pmc_r03.py
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/bev_r03; rm -rf $OUT; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $OUT/pmc -- python3 $R/bench.py --mode static --steps 6 --warmup 4 --no-config1 --no-stages --no-cpu-baseline --no-extra > $OUT/pmc.log 2>&1
tail -c 300 $OUT/pmc.log
python3 $R/tools/summarize_bev_pmc_r03.py
