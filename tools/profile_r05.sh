#!/bin/bash
# Round-5 evidence on the GPU box (writes gpurun_out/<tag>/):
#   bench.json                  python3 bench.py (the driver's command, no profiler)
#   roofline_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --roofline-only`
#   pmc_fetch / pmc_write       separate PMC passes of the same command
#   train_step_kernels.md       per-step kernel table of the training step
#   sconv_launches.md           every k_sconv_gemm<64,64,...> launch of one training step: grid, duration, what ran beside it
R=$GRAFT_REPO_ROOT; TAG=${1:-r05}; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_roof -o roof -- python3 $R/bench.py --roofline-only > $OUT/roofline.json 2> $OUT/roofline.err
cp /tmp/p_roof/roof_kernel_stats.csv $OUT/roofline_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -o f -- python3 $R/bench.py --roofline-only > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -o w -- python3 $R/bench.py --roofline-only > $OUT/pmc_write.log 2>&1
python3 $R/tools/pmc_per_kernel.py /tmp/p_fetch/f_counter_collection.csv FETCH_SIZE > $OUT/pmc_fetch.json
python3 $R/tools/pmc_per_kernel.py /tmp/p_write/w_counter_collection.csv WRITE_SIZE > $OUT/pmc_write.json
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_train -o train -- python3 $R/bench.py --steps 20 --warmup 5 --no-config1 --no-stages --no-cpu-baseline --no-extra > $OUT/train_trace.log 2>&1
python3 $R/tools/trace_window.py /tmp/p_train/train_kernel_trace.csv --steps 10 --top 200 --out $OUT/train_step_kernels.md --launches "k_sconv_gemm<64, 64" --launches-out $OUT/sconv_launches.md > /dev/null
head -3 /tmp/p_train/train_kernel_trace.csv > $OUT/trace_head.txt
ls -la $OUT; head -c 300 $OUT/bench.json; echo; cat $OUT/pmc_fetch.json $OUT/pmc_write.json; head -20 $OUT/sconv_launches.md
