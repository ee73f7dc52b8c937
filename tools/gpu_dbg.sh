#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_sparse_gpu.py -x -q -m gpu 2>&1 | tail -3
for v in -1 41; do echo "== rb variant $v"; RB_VARIANT=$v timeout 600 python tools/sconv_sweep.py -1 2>&1 | grep "rulebook"; done
