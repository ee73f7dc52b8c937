#!/bin/bash
# Round-2 evidence on the GPU box (writes gpurun_out/<tag>/; tools/summarize_r02.py turns it into profiles/):
#   bench.json                 python3 bench.py (the driver's command, no profiler)
#   roofline_kernel_stats.csv  rocprofv3 --kernel-trace --stats of `bench.py --roofline-only` (the event-bracketed pass)
#   pmc_fetch / pmc_write      separate PMC passes of the same command (FETCH_SIZE and WRITE_SIZE cannot share a pass)
#   train_step_kernels.md      per-step kernel table of the training step (trace window between loss kernels)
R=$GRAFT_REPO_ROOT; TAG=${1:-r02}; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_roof -o roof -- python3 $R/bench.py --roofline-only > $OUT/roofline.json 2> $OUT/roofline.err
cp /tmp/p_roof/roof_kernel_stats.csv $OUT/roofline_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -o f -- python3 $R/bench.py --roofline-only > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -o w -- python3 $R/bench.py --roofline-only > $OUT/pmc_write.log 2>&1
python3 $R/tools/pmc_per_kernel.py /tmp/p_fetch/f_counter_collection.csv FETCH_SIZE > $OUT/pmc_fetch.json
python3 $R/tools/pmc_per_kernel.py /tmp/p_write/w_counter_collection.csv WRITE_SIZE > $OUT/pmc_write.json
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_train -o train -- python3 $R/bench.py --steps 20 --warmup 5 --no-config1 --no-stages --no-cpu-baseline > $OUT/train_trace.log 2>&1
python3 $R/tools/trace_window.py /tmp/p_train/train_kernel_trace.csv --steps 10 --top 200 --out $OUT/train_step_kernels.md > /dev/null
ls -la $OUT; head -c 300 $OUT/bench.json; echo; cat $OUT/pmc_fetch.json $OUT/pmc_write.json
