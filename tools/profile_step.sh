#!/bin/bash
# tools/profile_step.sh [tag]: the per-kernel table of the training step alone (rocprofv3 --kernel-trace of bench.py's headline loop)
# -> gpurun_out/<tag>/train_step_kernels.md
R=$GRAFT_REPO_ROOT; TAG=${1:-step}; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_train -o train -- python3 $R/bench.py --steps 20 --warmup 5 --no-config1 --no-stages --no-cpu-baseline --no-extra > $OUT/train_trace.log 2>&1
python3 $R/tools/trace_window.py /tmp/p_train/train_kernel_trace.csv --steps 10 --top 200 --out $OUT/train_step_kernels.md --seq $OUT/train_step_seq.txt > /dev/null
head -70 $OUT/train_step_kernels.md
