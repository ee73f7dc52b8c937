#!/bin/bash
# HIP runtime switches (read by libamdhip64 at initialisation) against the headline, one at a time, a base run between groups:
#   tools/runtime_knobs.sh "VAR=VALUE VAR2=VALUE ..."   (each token is one run)
cd $GRAFT_REPO_ROOT
run() {
  env $1 timeout 300 python bench.py --no-config1 --no-cpu-baseline --no-extra --no-stages 2> gpurun_out/knob.err > gpurun_out/knob.json
  python -c "
import json,sys
try:
    d=json.loads(open('gpurun_out/knob.json').read().strip().splitlines()[-1]); print('%-44s %.4f ms  %.1f frames/s' % ('$1', d['ms_per_step'], d['value']))
except Exception as e:
    print('%-44s FAILED %s' % ('$1', e))
"
}
run GLX_NOP=0
for kv in $1; do run $kv; done
run GLX_NOP=0
