cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_sparse_gpu.py -q -m gpu -k "pair_lists or weight_gradient_over" 2>&1 | tail -2
PAIRS=1 timeout 300 python tools/wgrad_sweep.py 2>&1 | tail -11 | cut -c60-150
