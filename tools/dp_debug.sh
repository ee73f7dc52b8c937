#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
  PORT=$((29600 + i))
  for r in 0 1; do
    RANK=$r LOCAL_RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT GLX_DP_DEBUG=1 GLX_CONV3X3_TH=6 python3 tests/_dp_step_worker.py > /tmp/dp_$r.log 2>&1 &
  done
  wait
  echo "--- run $i"; grep -h "own recorded" /tmp/dp_0.log /tmp/dp_1.log; grep -o "grad_err_max.: [0-9.e-]*" /tmp/dp_0.log
done
