#!/bin/bash
# stage stamps of the recorded step with VAR=1 / VAR=0:  tools/ab_stages.sh VAR
cd $GRAFT_REPO_ROOT; VAR=$1
for v in 1 0 1 0; do
  env $VAR=$v timeout 300 python bench.py --no-config1 --no-cpu-baseline --no-extra 2> gpurun_out/abs.err > gpurun_out/abs.json
  python -c "
import json
d=json.loads(open('gpurun_out/abs.json').read().strip().splitlines()[-1]); s=d['stages_ms']
print('$VAR=$v', d['ms_per_step'], ' '.join('%s=%.3f' % (k[:28], v) for k, v in s.items() if k != 'note'))"
done
