#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
echo "== fused BN"; python tools/train_step_time.py 2>&1 | grep -v amdgpu
echo "== torch BN"; GLX_FUSED_BN=0 python tools/train_step_time.py 2>&1 | grep -v amdgpu
