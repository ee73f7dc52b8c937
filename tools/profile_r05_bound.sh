#!/bin/bash
# Round 5: what bounds k_sconv_gemm<64,64> and the BEV 3x3 kernels -- SQ / LDS / clock counters and the in-kernel clock.
#   gpurun_out/<tag>/sconv_*.csv, bev_*.csv   per-dispatch counter values (rocprofv3 --pmc, one pass per counter group)
#   gpurun_out/<tag>/sconv_tiles.txt          TRACE build: per-block timeline, phase shares, in-kernel clock (s_memtime / s_memrealtime)
#   gpurun_out/<tag>/bev_micro.txt            the un-profiled event-timed averages of the same BEV program
# tools/summarize_r05_bound.py turns them into profiles/r05_sconv_bound.md, r05_bev_mfma.md, r05_bev_pmc.json
R=$GRAFT_REPO_ROOT; TAG=${1:-r05_bound}; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES"
B="SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"
D="GRBM_GUI_ACTIVE"
python3 $R/tools/bev_micro.py 20 > $OUT/bev_micro.txt 2>&1
GEMM=1 ONLY6464=1 NW=8 python3 $R/tools/sconv_tiles.py > $OUT/sconv_tiles.txt 2>&1
i=0
for G in "$A" "$B" "$C" "$D" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/p_s$i /tmp/p_b$i
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/p_s$i -o s -- python3 $R/bench.py --roofline-only > $OUT/sconv_pass$i.log 2>&1
  cp $(find /tmp/p_s$i -name "*counter_collection.csv" | head -1) $OUT/sconv_counters_$i.csv 2>/dev/null
  cp $(find /tmp/p_s$i -name "*kernel_trace.csv" | head -1) $OUT/sconv_trace_$i.csv 2>/dev/null
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/p_b$i -o b -- python3 $R/tools/bev_micro.py 10 > $OUT/bev_pass$i.log 2>&1
  cp $(find /tmp/p_b$i -name "*counter_collection.csv" | head -1) $OUT/bev_counters_$i.csv 2>/dev/null
  cp $(find /tmp/p_b$i -name "*kernel_trace.csv" | head -1) $OUT/bev_trace_$i.csv 2>/dev/null
done
python3 $R/tools/summarize_r05_bound.py $OUT > $OUT/summary.log 2>&1
# the csv files are large: keep the summaries, drop the raw per-dispatch tables beyond 8 MB
find $OUT -name "*.csv" -size +8M -delete
ls -la $OUT; tail -5 $OUT/summary.log; cat $OUT/bev_micro.txt; tail -30 $OUT/sconv_tiles.txt
