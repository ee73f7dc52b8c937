#!/bin/bash
# Per-step kernel table of the training step only (rocprofv3 --kernel-trace): gpurun_out/<tag>_train_step_kernels.md
R=$GRAFT_REPO_ROOT; TAG=${1:-t}; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_train_$TAG -o train -- python3 $R/bench.py --steps 20 --warmup 5 --no-config1 --no-stages --no-cpu-baseline --no-extra > $R/gpurun_out/${TAG}_train_trace.log 2>&1
python3 $R/tools/trace_window.py /tmp/p_train_$TAG/train_kernel_trace.csv --steps 10 --top 200 --out $R/gpurun_out/${TAG}_train_step_kernels.md --seq $R/gpurun_out/${TAG}_seq.txt > /dev/null
head -30 $R/gpurun_out/${TAG}_train_step_kernels.md | cut -c1-150
