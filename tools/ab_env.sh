#!/bin/bash
# A/B of one environment switch on the headline: tools/ab_env.sh VAR [rounds]  (alternating runs, VAR=1 / VAR=0)
cd $GRAFT_REPO_ROOT; VAR=$1; N=${2:-2}
for i in $(seq 1 $N); do for v in 1 0; do
  env $VAR=$v timeout 300 python bench.py --no-config1 --no-cpu-baseline --no-extra --no-stages 2> gpurun_out/ab.err > gpurun_out/ab.json
  python -c "import json; d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('$VAR=$v', d['ms_per_step'], d['value'])"
done; done
