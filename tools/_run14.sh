cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --no-config1 --no-cpu-baseline --no-extra > gpurun_out/b_def.json 2> gpurun_out/b_def.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/b_def.json").read().strip().splitlines()[-1])
print("default", d["value"], d["ms_per_step"])
PY
DEBUG_HIP_FORCE_GRAPH_QUEUES=4 timeout 600 python bench.py --no-config1 --no-cpu-baseline --no-extra > gpurun_out/b_q4.json 2> gpurun_out/b_q4.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/b_q4.json").read().strip().splitlines()[-1])
print("q4", d["value"], d["ms_per_step"])
PY
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
