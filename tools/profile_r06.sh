#!/bin/bash
# Round-6 evidence on the GPU box (writes gpurun_out/<tag>/): as profile_r05.sh plus
#   config4_kernel_stats.csv    rocprofv3 --kernel-trace --stats of `bench.py --config4-only`
R=$GRAFT_REPO_ROOT; TAG=${1:-r06}; OUT=$R/gpurun_out/$TAG; WHAT=${2:-all}
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
fi
if [ "$WHAT" = all ] || [ "$WHAT" = config4 ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -o c4 -- python3 $R/bench.py --config4-only > $OUT/config4.json 2> $OUT/config4.err
cp /tmp/p_c4/c4_kernel_stats.csv $OUT/config4_kernel_stats.csv
fi
if [ "$WHAT" = all ] || [ "$WHAT" = roof ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_roof -o roof -- python3 $R/bench.py --roofline-only > $OUT/roofline.json 2> $OUT/roofline.err
cp /tmp/p_roof/roof_kernel_stats.csv $OUT/roofline_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -o f -- python3 $R/bench.py --roofline-only > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -o w -- python3 $R/bench.py --roofline-only > $OUT/pmc_write.log 2>&1
python3 $R/tools/pmc_per_kernel.py /tmp/p_fetch/f_counter_collection.csv FETCH_SIZE > $OUT/pmc_fetch.json
python3 $R/tools/pmc_per_kernel.py /tmp/p_write/w_counter_collection.csv WRITE_SIZE > $OUT/pmc_write.json
fi
if [ "$WHAT" = all ] || [ "$WHAT" = train ]; then
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_train -o train -- python3 $R/bench.py --steps 20 --warmup 5 --no-config1 --no-stages --no-strict --no-cpu-baseline --no-extra > $OUT/train_trace.log 2>&1
python3 $R/tools/trace_window.py /tmp/p_train/train_kernel_trace.csv --steps 10 --top 200 --out $OUT/train_step_kernels.md --launches "k_sconv_gemm<64, 64" --launches-out $OUT/sconv_launches.md > /dev/null
fi
ls -la $OUT; head -c 400 $OUT/bench.json 2>/dev/null; echo
