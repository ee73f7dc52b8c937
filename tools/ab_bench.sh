#!/bin/bash
# A/B of the headline step under environment settings: tools/ab_bench.sh ROUNDS "ENV_A=1" "ENV_B=0 ENV_C=2" ...
# prints ms_per_step of `bench.py --no-extra` per setting, alternating (one box: compare within a call only)
R=${GRAFT_REPO_ROOT:-.}; N=$1; shift
for i in $(seq $N); do
  for e in "$@"; do
    ms=$(env $e python3 $R/bench.py --no-extra --no-config1 --no-stages --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "$e  $ms"
  done
done
