cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_sparse_gpu.py -x -q -m gpu -k "pair_lists or weight_gradient_over" 2>&1 | tail -3
bash tools/trace_step.sh r04b > /dev/null 2>&1
ls gpurun_out | grep r04b
