cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_conv2d_gpu.py -q -m gpu -k "anchor_head" 2>&1 | tail -2
timeout 900 python bench.py --no-config1 --no-cpu-baseline > gpurun_out/b_acc.json 2> gpurun_out/b_acc.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/b_acc.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print(json.dumps(d["bev"]["conv3x3_error_vs_fp64"]))
PY
