#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/dbg; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_sparse_gpu.py -x -q -m gpu 2>&1 | tail -3
python tools/train_step_time.py 2>&1 | grep -v amdgpu
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_stats -- python3 $R/tools/train_step_time.py > $OUT/train_stats.log 2>&1
