cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_losses.py tests/test_train_step_gpu.py tests/test_reference_step_gpu.py tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do
for v in 1 0; do
GLX_ROI_LOSSES_ONE_LAUNCH=$v timeout 300 python bench.py --no-config1 --no-cpu-baseline --no-extra 2> gpurun_out/rl.err > gpurun_out/rl.json
python - <<PY
import json
d=json.loads(open("gpurun_out/rl.json").read().strip().splitlines()[-1])
print("one_launch=$v", d["ms_per_step"], {k:v for k,v in d["stages_ms"].items() if "RoI" in k or "loss" in k})
PY
done; done
