#!/bin/bash
# tools/ab_step.sh "ENV_A" "ENV_B" [pairs] [extra bench args]: the headline step under two environments, alternating runs on one box (ms per step)
R=$GRAFT_REPO_ROOT; A=$1; B=$2; N=${3:-3}; X=$4; cd $R
for i in $(seq 1 $N); do
  for E in "$A" "$B"; do
    ms=$(env $E python bench.py --steps 60 --warmup 10 --no-extra --no-config1 --no-stages --no-cpu-baseline $X 2>/tmp/ab_err.txt | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])" 2>/dev/null || tail -3 /tmp/ab_err.txt)
    echo "[$E] $ms"
  done
done
