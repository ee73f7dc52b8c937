#!/bin/bash
# Round profile on the GPU box: headline bench, kernel-trace stats of the same command, and
# separate PMC passes for HBM traffic (FETCH_SIZE / WRITE_SIZE cannot share a pass:
# MI355X_MICROARCH.md "rocprofv3 PMC slots").  The PMC passes use --mode static (eager launches):
# counters are attributed per dispatch either way, and it keeps the profiler off graph replay.
R=$GRAFT_REPO_ROOT; TAG=${1:-r01}; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-train > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --mode static --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-roofline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --mode static --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-roofline > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $R/bench.py --mode static --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-roofline > $OUT/pmc_l2.log 2>&1
cut -c1-400 $OUT/bench.json; echo; tail -2 $OUT/stats.log
